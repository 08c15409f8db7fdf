"""Sweeps rt_tuning on config 5 (1080p, mode 5) — or, with --syn, on the 10 M-triangle synthetic scene — and prints Mpaths/s per setting."""
import itertools, os, sys, time
SYN = "--syn" in sys.argv
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import torch
import ray_tracing_amd as rt
from ray_tracing_amd import scenes
t, n = scenes.synthetic_scene_cached("/tmp", 256, 100) if SYN else scenes.chap5_scene()
hdr, cache = scenes.hdr_and_cache()
sc = rt.Scene(t, n, hdr, cache, device=0)
eye, cam = rt.camera(0.0, 20.0, 4.0)
W, H = 1920, 1080
d = rt.make_desc(W, H, mode=5, eye=eye, camera_rotate=cam)
acc = torch.zeros((H, W, 3), dtype=torch.float32, device="cuda:0")
st = torch.cuda.current_stream().cuda_stream
def run(spp, **kw):
    sc.set_tuning()
    sc.set_tuning(**kw)
    sc.render_device(d, 0, spp, acc.data_ptr(), st); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(3):
        sc.render_device(d, (i + 1) * spp, spp, acc.data_ptr(), st)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 3
    print(f"{kw} spp {spp}: {W * H * spp / dt / 1e6:8.1f} Mpaths/s", flush=True)
if SYN:
    W, H = 3840, 2160  # the bench line's shape: 4K, 8 frames per step, consecutive launches overlapped
    d = rt.make_desc(W, H, mode=5, eye=eye, camera_rotate=cam)
    acc = torch.zeros((H, W, 3), dtype=torch.float32, device="cuda:0")
    def runk(**kw):
        sc.set_tuning()
        sc.set_tuning(**kw)
        sc.render_device(d, 0, 8, acc.data_ptr(), st); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(N):
            sc.render_device(d, (i + 1) * 8, 8, acc.data_ptr(), st)
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / N * 1e3
        print(f"{str(kw):60s} {ms:8.2f} ms per 8-frame 4K step  {W * H * 8 / ms / 1e3:8.1f} Mpaths/s", flush=True)
    N = 15  # odd, plus the warm-up launch: every figure starts on the slot the previous one started on; `flip` changes it
    for flip in (0, 1):
        if flip:
            sc.render_device(d, 0, 8, acc.data_ptr(), st)
        for kw in ({}, {"leaf_lanes": 16, "shade_lanes": 24}, {"leaf_lanes": 16, "shade_lanes": 20}, {"leaf_lanes": 16, "shade_lanes": 16},
                   {"leaf_lanes": 12, "shade_lanes": 24}, {"leaf_lanes": 12, "shade_lanes": 20}, {"leaf_lanes": 8, "shade_lanes": 24},
                   {"leaf_lanes": 16, "shade_lanes": 24, "wait_budget": 2000}, {"leaf_lanes": 16, "shade_lanes": 24, "wait_budget": 8000},
                   {"leaf_lanes": 16, "shade_lanes": 24, "unit_frames": 4}, {}):
            runk(**kw)
    runk(overlap_frames=1)
    runk(single_kernel=1)
    sys.exit(0)
run(64)
run(64)
for sl in (24, 32, 40, 56, 64):
    run(64, shade_lanes=sl)
for ll in (2, 4, 12, 16, 32):
    run(64, leaf_lanes=ll)
for wb in (1000, 2000, 8000, 16000):
    run(64, wait_budget=wb)
for uf in (1, 3, 4, 8):
    run(64, unit_frames=uf)
for lf in (32, 128, 256):
    run(256, launch_frames=lf)
run(64)
