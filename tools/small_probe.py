"""Small images (the reference's 512 x 512 window, 1280 x 720), 1 ... 64 frames per call: ms per call with the library's choice of form,
with one kernel forced and with overlap off.  The numbers behind kTwoKernelMinSamples (rt_device.hip)."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, torch
import ray_tracing_amd as rt
from ray_tracing_amd import scenes
hdr, cache = scenes.hdr_and_cache()
st = torch.cuda.current_stream().cuda_stream
t, n = scenes.chap5_scene()
sc = rt.Scene(t, n, hdr, cache, device=0)
eye, cam = rt.camera(0.0, 20.0, 4.0)
for (W, H) in ((512, 512), (1280, 720)):
    d = rt.make_desc(W, H, mode=5, eye=eye, camera_rotate=cam)
    acc = torch.zeros((H, W, 3), dtype=torch.float32, device="cuda:0")
    for fpc in (1, 2, 4, 8, 64):
        for kw in ({}, {"single_kernel": 1}, {"overlap_frames": 1}):
            sc.set_tuning(); sc.set_tuning(**kw) if kw else None
            for k in range(6): sc.render_device(d, k * fpc, fpc, acc.data_ptr(), st)
            torch.cuda.synchronize()
            N = 200 if fpc < 8 else 40
            t0 = time.perf_counter()
            for k in range(N): sc.render_device(d, (k + 6) * fpc, fpc, acc.data_ptr(), st)
            torch.cuda.synchronize()
            w = (time.perf_counter() - t0) / N * 1e3
            print(f"{W}x{H} frames per call {fpc:2d} {str(kw):24s} {w:7.3f} ms per call  {1e3 / w:8.0f} calls/s  {W*H*fpc/w/1e3:7.1f} Mpaths/s", flush=True)
