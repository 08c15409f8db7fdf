"""Chains of overlapped launches that start on an idle device, with 2 and 3 warm-up launches before the timed ones (the slot the
timed chain would start on alternates unless the library picks it): config 5 at 8 / 32 frames per call, the synthetic scene at 4K x 8."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import numpy as np, torch
import ray_tracing_amd as rt
from ray_tracing_amd import scenes
hdr, cache = scenes.hdr_and_cache()
st = torch.cuda.current_stream().cuda_stream


def run(sc, d, W, H, fpc, N, label):
    acc = torch.zeros((H, W, 3), dtype=torch.float32, device="cuda:0")
    for warm in (2, 3, 2, 3):
        for i in range(warm):
            sc.render_device(d, i * fpc, fpc, acc.data_ptr(), st)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(N):
            sc.render_device(d, (i + 3) * fpc, fpc, acc.data_ptr(), st)
        torch.cuda.synchronize()
        w = (time.perf_counter() - t0) / N * 1e3
        print(f"{label} frames per call {fpc:2d}, {warm} warm-up launches: {w:7.3f} ms per call  {W * H * fpc / w / 1e3:7.1f} Mpaths/s", flush=True)


t, n = scenes.chap5_scene()
sc = rt.Scene(t, n, hdr, cache, device=0)
eye, cam = rt.camera(0.0, 20.0, 4.0)
d = rt.make_desc(1920, 1080, mode=5, eye=eye, camera_rotate=cam)
for fpc in (1, 8, 32):
    run(sc, d, 1920, 1080, fpc, 40, "config5")
t, n = scenes.synthetic_scene_cached("/tmp", 256, 100)
sc2 = rt.Scene(t, n, hdr, cache, device=0)
d2 = rt.make_desc(3840, 2160, mode=5, eye=eye, camera_rotate=cam)
run(sc2, d2, 3840, 2160, 8, 32, "synthetic 4K")
