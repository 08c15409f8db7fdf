"""Frames per work unit on the config-4 scene from three orbit cameras (coverage 0.05 / 0.17 / 0.37 of the image): wall, path-loop and
camera-ray kernel ms per 64-frame 1080p call at 2 ... 64 frames per unit.  The numbers behind effective_unit_frames (rt_device.hip)."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import numpy as np, torch
import ray_tracing_amd as rt
from ray_tracing_amd import scenes
hdr, cache = scenes.hdr_and_cache()
W, H, SPP, N = 1920, 1080, 64, 4
st = torch.cuda.current_stream().cuda_stream
acc = torch.zeros((H, W, 3), dtype=torch.float32, device="cuda:0")
t, n = scenes.chap4_scene()
sc = rt.Scene(t, n, hdr, cache, device=0)
for (rot, up, r) in ((0, 0, 4.0), (0, 0, 2.5), (90, 45, 3.0)):
    eye, cam = rt.camera(float(rot), float(up), float(r))
    d = rt.make_desc(W, H, mode=4, eye=eye, camera_rotate=cam)
    print("cam", rot, up, r, "coverage", sc.screen_coverage(d))
    for uf in (2, 4, 8, 16, 32, 64):
        sc.set_tuning(unit_frames=uf)
        sc.render_device(d, 0, SPP, acc.data_ptr(), st); torch.cuda.synchronize()
        sc.kernel_timing(reset=True)
        t0 = time.perf_counter()
        for i in range(N):
            sc.render_device(d, (i + 1) * SPP, SPP, acc.data_ptr(), st)
        torch.cuda.synchronize()
        w = (time.perf_counter() - t0) / N * 1e3
        kt = sc.kernel_timing(reset=True)
        print(f"  uf {uf:3d}: wall {w:6.2f} ms  path loop {kt.avg_ms:6.2f}  camera {kt.first_avg_ms:6.2f}")
    sc.set_tuning()
