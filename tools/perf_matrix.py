"""Times the path-loop kernel on the standard workloads (HIP events inside the library), optionally with culling off,
and prints one line per case.  Iteration tool for kernel work: python tools/perf_matrix.py [--cases c5,c4,syn,prim] [--nocull]"""
import argparse, os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import numpy as np, torch
import ray_tracing_amd as rt
from ray_tracing_amd import scenes

ap = argparse.ArgumentParser()
ap.add_argument("--cases", default="c5,c4,c4fill,c3,syn,prim")
ap.add_argument("--nocull", action="store_true", help="also time every case with rt_tuning.no_cull = 1")
ap.add_argument("--cache-dir", default="/tmp")
ap.add_argument("--spp", type=int, default=32)
ap.add_argument("--launches", type=int, default=5)
ap.add_argument("--single", action="store_true", help="rt_tuning.single_kernel = 1 (the one-kernel form of the path loop)")
ap.add_argument("--stats", action="store_true", help="print per-ray traversal means: the reference's walk and the executed walk")
a = ap.parse_args()
cases = a.cases.split(",")
hdr, cache = scenes.hdr_and_cache()
W, H = 1920, 1080
st = torch.cuda.current_stream().cuda_stream
acc = torch.zeros((H, W, 3), dtype=torch.float32, device="cuda:0")


def time_render(sc, d, spp, launches, label):
    for cull in ([True, False] if a.nocull else [True]):
        sc.set_tuning(no_cull=0 if cull else 1, single_kernel=1 if a.single else 0)
        sc.render_device(d, 0, spp, acc.data_ptr(), st)
        torch.cuda.synchronize()
        rt.kernel_time_ms(reset=True)
        t0 = time.perf_counter()
        for i in range(launches):
            sc.render_device(d, (i + 1) * spp, spp, acc.data_ptr(), st)
        torch.cuda.synchronize()
        wall = (time.perf_counter() - t0) / launches * 1e3
        kt = sc.kernel_timing(reset=True)
        ms = kt.avg_ms * kt.launches / launches  # per render call (a call may be several launches)
        first = kt.first_avg_ms * kt.launches / launches
        print(f"{label:34s} cull={'on ' if cull else 'off'} path-loop kernel {ms:9.3f} ms  camera-ray kernel {first:7.3f} ms  wall {wall:9.3f} ms  {W * H * spp / wall / 1e3:9.1f} Mpaths/s", flush=True)
    sc.set_tuning(no_cull=0)
    if a.stats:
        for ex in (False, True):
            s_ = sc.render_stats(d, 0, min(spp, 2), executed=ex)
            r = max(s_["rays"], 1)
            print(f"    {'executed ' if ex else 'reference'} walk per ray: I {s_['internal_nodes'] / r:8.2f}  Lf {s_['leaf_nodes'] / r:7.2f}  T {s_['triangles_tested'] / r:7.2f}  H {s_['hit_updates'] / r:5.2f}  rays/path {s_['rays'] / max(s_['paths'], 1):.3f}", flush=True)


if "c5" in cases:
    t, n = scenes.chap5_scene()
    sc = rt.Scene(t, n, hdr, cache, device=0)
    eye, cam = rt.camera(0.0, 20.0, 4.0)
    time_render(sc, rt.make_desc(W, H, mode=5, eye=eye, camera_rotate=cam), a.spp, a.launches, "config5 1080p mode5")
    if "c3" in cases:
        time_render(sc, rt.make_desc(W, H, mode=3, eye=eye, camera_rotate=cam), a.spp, a.launches, "config5 scene mode3")
    if "one" in cases:
        time_render(sc, rt.make_desc(W, H, mode=5, eye=eye, camera_rotate=cam), 1, 50, "config5 1 frame per call")
if "c4" in cases:
    t, n = scenes.chap4_scene()
    sc = rt.Scene(t, n, hdr, cache, device=0)
    eye, cam = rt.camera(0.0, 0.0, 4.0)
    time_render(sc, rt.make_desc(W, H, mode=4, eye=eye, camera_rotate=cam), a.spp, a.launches, "config4 1080p mode4")
if "c4fill" in cases:
    # config 4 with the sphere filling the frame (98 % of the camera rays hit): the orbit camera looks at the origin and the sphere
    # sits at (-0.8, 0.8, 0) (c4:694), so the eye goes on the line through both — at the start-up camera 97 % of the rays escape and
    # the rate is an environment-lookup rate that says nothing about the anisotropic BRDF
    t, n = scenes.chap4_scene()
    sc = rt.Scene(t, n, hdr, cache, device=0)
    eye, cam = rt.camera(90.0, 45.0, 1.9)
    time_render(sc, rt.make_desc(W, H, mode=4, eye=eye, camera_rotate=cam), a.spp, a.launches, "config4 sphere fills the frame")
if "syn" in cases:
    t0 = time.time()
    t, n = scenes.synthetic_scene_cached(a.cache_dir, 256, 100)
    print(f"synthetic scene: {t.shape[0]} triangles, {time.time() - t0:.1f} s", flush=True)
    sc = rt.Scene(t, n, hdr, cache, device=0)
    print("info:", {k: getattr(sc.info(), k) for k, _ in sc.info()._fields_}, flush=True)
    for (rot, up, r) in ((0.0, 20.0, 4.0), (15.0, 25.0, 6.0)):
        eye, cam = rt.camera(rot, up, r)
        time_render(sc, rt.make_desc(W, H, mode=5, eye=eye, camera_rotate=cam), 4, 2, f"synthetic10M cam({rot:g},{up:g},{r:g})")
if "prim" in cases:
    t, n = scenes.chap2_scene()
    sc = rt.Scene(t, n, device=0)
    tri = torch.empty((H, W), dtype=torch.int32, device="cuda:0")
    dist = torch.empty((H, W), dtype=torch.float32, device="cuda:0")
    for r in (4.0, 1.7):
        d = rt.make_desc(W, H, eye=(0, 0, r))
        for cull in ([True, False] if a.nocull else [True]):
            sc.set_tuning(no_cull=0 if cull else 1)
            for _ in range(3):
                sc.primary_hits_device(d, tri.data_ptr(), dist.data_ptr(), st)
            torch.cuda.synchronize()
            rt.kernel_time_ms(reset=True)
            for _ in range(20):
                sc.primary_hits_device(d, tri.data_ptr(), dist.data_ptr(), st)
            torch.cuda.synchronize()
            ms, k = rt.kernel_time_ms(reset=True)
            print(f"primary hits chap2 r={r:<4g}           cull={'on ' if cull else 'off'} kernel {ms:9.4f} ms  {W * H / ms / 1e3:9.0f} Mrays/s  hits {(tri >= 0).sum().item()}", flush=True)
