"""Frames per call on config 5 at 1080p: ms per call / per frame in both forms of the path loop, and with the overlap of
consecutive short launches (rt_tuning.overlap_frames) reaching further than its default of 8 frames."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import numpy as np, torch
import ray_tracing_amd as rt
from ray_tracing_amd import scenes
hdr, cache = scenes.hdr_and_cache()
t, n = scenes.chap5_scene()
sc = rt.Scene(t, n, hdr, cache, device=0)
eye, cam = rt.camera(0.0, 20.0, 4.0)
W, H = 1920, 1080
d = rt.make_desc(W, H, mode=5, eye=eye, camera_rotate=cam)
st = torch.cuda.current_stream().cuda_stream
acc = torch.zeros((H, W, 3), dtype=torch.float32, device="cuda:0")
for single, ov in ((0, 8), (1, 8), (0, 17), (0, 33), (0, 65)):
    for fpc in (1, 2, 4, 7, 8, 16, 32, 64):
        sc.set_tuning(single_kernel=single, overlap_frames=ov)
        for i in range(5):
            sc.render_device(d, i * fpc, fpc, acc.data_ptr(), st)
        torch.cuda.synchronize()
        N = 40
        t0 = time.perf_counter()
        for i in range(N):
            sc.render_device(d, (i + 5) * fpc, fpc, acc.data_ptr(), st)
        torch.cuda.synchronize()
        w = (time.perf_counter() - t0) / N * 1e3
        print(f"single_kernel={single} overlap_frames={ov:2d} frames per call {fpc:2d}: {w:7.3f} ms per call, {w / fpc:6.3f} ms per frame, {W * H * fpc / w / 1e3:7.1f} Mpaths/s", flush=True)
