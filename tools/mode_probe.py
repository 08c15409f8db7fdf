"""Scheduler thresholds on the other modes: config 4 with the sphere filling the frame (chapter-4 mode, anisotropic BRDF) and the
config-5 scene in chapter-3 mode, 1080p x 64 frames per call."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, torch
import ray_tracing_amd as rt
from ray_tracing_amd import scenes
hdr, cache = scenes.hdr_and_cache()
st = torch.cuda.current_stream().cuda_stream
W, H, F = 1920, 1080, 64
acc = torch.zeros((H, W, 3), dtype=torch.float32, device="cuda:0")
for name, (t, n), mode, camargs in (("config4 fill", scenes.chap4_scene(), 4, (90.0, 45.0, 1.9)), ("config5 scene mode 3", scenes.chap5_scene(), 3, (0.0, 20.0, 4.0)),
                                     ("config5 scene mode 4", scenes.chap5_scene(), 4, (0.0, 20.0, 4.0))):
    sc = rt.Scene(t, n, hdr, cache, device=0)
    eye, cam = rt.camera(*camargs)
    d = rt.make_desc(W, H, mode=mode, eye=eye, camera_rotate=cam)
    for kw in ({}, {"shade_lanes": 32}, {"shade_lanes": 40}, {"shade_lanes": 56}, {"leaf_lanes": 4}, {"leaf_lanes": 12}, {"leaf_lanes": 12, "shade_lanes": 40}, {}):
        sc.set_tuning(); sc.set_tuning(**kw) if kw else None
        sc.render_device(d, 0, F, acc.data_ptr(), st); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for k in range(4): sc.render_device(d, (k + 1) * F, F, acc.data_ptr(), st)
        torch.cuda.synchronize()
        w = (time.perf_counter() - t0) / 4 * 1e3
        print(f"{name:22s} {str(kw):44s} {w:7.2f} ms  {W*H*F/w/1e3:7.1f} Mpaths/s", flush=True)
