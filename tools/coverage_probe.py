"""Where does the two-kernel form of the path loop stop paying?  Config-4 and config-5 scenes from orbit cameras at several
distances (the share of camera rays that hit anything falls with the distance), each timed with the unit size the library picks
from the view (rt_scene_screen_coverage) and with 2 / 4 / 8 frames per unit set by hand.  Prints wall ms per 64-frame 1080p call,
the share of camera rays that escape and the root box's share of the image."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import numpy as np, torch
import ray_tracing_amd as rt
from ray_tracing_amd import scenes

hdr, cache = scenes.hdr_and_cache()
W, H, SPP, N = 1920, 1080, 64, 4
st = torch.cuda.current_stream().cuda_stream
acc = torch.zeros((H, W, 3), dtype=torch.float32, device="cuda:0")


def wall(sc, d, **tuning):
    if tuning:
        sc.set_tuning(**tuning)
    sc.render_device(d, 0, SPP, acc.data_ptr(), st)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(N):
        sc.render_device(d, (i + 1) * SPP, SPP, acc.data_ptr(), st)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / N * 1e3


for name, (t, n), mode, cams in (
        ("config4", scenes.chap4_scene(), 4, [(0, 0, 4.0), (0, 0, 3.0), (0, 0, 2.5), (0, 0, 2.0), (90, 45, 4.0), (90, 45, 3.0), (90, 45, 2.5), (90, 45, 1.9)]),
        ("config5", scenes.chap5_scene(), 5, [(0, 20, 4.0), (0, 20, 8.0), (0, 60, 4.0), (0, -30, 6.0), (0, 5, 12.0)])):
    sc = rt.Scene(t, n, hdr, cache, device=0)
    for (rot, up, r) in cams:
        eye, cam = rt.camera(float(rot), float(up), float(r))
        d = rt.make_desc(W, H, mode=mode, eye=eye, camera_rotate=cam)
        sc.set_tuning()
        cs = sc.render_stats_camera(d, 0, 1, executed=True)
        esc = cs["paths"] / max(cs["rays"], 1)
        cov, uf = sc.screen_coverage(d)
        auto = wall(sc, d)
        sc.set_tuning()
        a = wall(sc, d, unit_frames=2)
        b = wall(sc, d, unit_frames=4)
        c = wall(sc, d, unit_frames=8)
        sc.set_tuning()
        print(f"{name} cam({rot:g},{up:g},{r:g}): escaped {esc * 100:5.1f} %  box coverage {cov:.3f} -> {uf} frames per unit: {auto:7.2f} ms   "
              f"(2: {a:7.2f}  4: {b:7.2f}  8: {c:7.2f})", flush=True)
