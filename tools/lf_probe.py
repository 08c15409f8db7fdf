"""Frames per launch on small images: config 5 at 512 x 512 and 1280 x 720, one 512-frame call cut into launches of 64 ... 512 frames."""
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
for (W, H) in ((512, 512), (1280, 720), (1920, 1080)):
    d = rt.make_desc(W, H, mode=5, eye=eye, camera_rotate=cam)
    acc = torch.zeros((H, W, 3), dtype=torch.float32, device="cuda:0")
    F = 512
    for lf in (32, 64, 128, 256, 512):
        sc.set_tuning(); sc.set_tuning(launch_frames=lf)
        sc.render_device(d, 0, F, acc.data_ptr(), st); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for k in range(3): sc.render_device(d, (k + 1) * F, F, acc.data_ptr(), st)
        torch.cuda.synchronize()
        w = (time.perf_counter() - t0) / 3 * 1e3
        print(f"{W}x{H} {F}-frame call, launches of {lf:3d} frames ({W*H*lf/1e6:6.1f} M samples): {w:8.2f} ms  {W*H*F/w/1e3:7.1f} Mpaths/s", flush=True)
