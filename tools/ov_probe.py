"""Does overlap pay by frames or by samples?  Config 5 at 512 x 512 and 1280 x 720, 32 ... 256 frames per call, overlap off / on."""
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
    for fpc in (32, 64, 128, 256):
        for ov in (1, 1025):
            sc.set_tuning(); sc.set_tuning(overlap_frames=ov)
            for k in range(3): sc.render_device(d, k * fpc, fpc, acc.data_ptr(), st)
            torch.cuda.synchronize()
            N = 24
            t0 = time.perf_counter()
            for k in range(N): sc.render_device(d, (k + 3) * fpc, fpc, acc.data_ptr(), st)
            torch.cuda.synchronize()
            w = (time.perf_counter() - t0) / N * 1e3
            print(f"{W}x{H} frames per call {fpc:3d} ({W*H*fpc/1e6:5.1f} M samples) overlap {'on ' if ov > 1 else 'off'}: {w:7.3f} ms  {W*H*fpc/w/1e3:7.1f} Mpaths/s", flush=True)
