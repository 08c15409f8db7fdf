"""Scheduler thresholds (leaf_lanes / shade_lanes) on the synthetic generator's scenes of 0.26 M, 0.64 M and 2.56 M triangles (all get an
entry BVH), at 4K x 8 frames and 1080p x 64 frames per call: the 16 / 24 default of entry-BVH scenes against config 5's 8 / 48."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, torch
import ray_tracing_amd as rt
from ray_tracing_amd import scenes
hdr, cache = scenes.hdr_and_cache()
st = torch.cuda.current_stream().cuda_stream
eye, cam = rt.camera(0.0, 20.0, 4.0)
for nobj, sub in ((16, 100), (64, 100), (64, 32)):
    t, n = scenes.synthetic_scene(nobj, sub)
    sc = rt.Scene(t, n, hdr, cache, device=0)
    i = sc.info()
    for (W, H, fpc) in ((3840, 2160, 8), (1920, 1080, 64)):
        d = rt.make_desc(W, H, mode=5, eye=eye, camera_rotate=cam)
        acc = torch.zeros((H, W, 3), dtype=torch.float32, device="cuda:0")
        for kw in ({"leaf_lanes": 8, "shade_lanes": 48}, {"leaf_lanes": 16, "shade_lanes": 24}, {"leaf_lanes": 12, "shade_lanes": 32}):
            sc.set_tuning(); sc.set_tuning(**kw)
            for k in range(2): sc.render_device(d, k * fpc, fpc, acc.data_ptr(), st)
            torch.cuda.synchronize()
            N = 8
            t0 = time.perf_counter()
            for k in range(N): sc.render_device(d, (k + 2) * fpc, fpc, acc.data_ptr(), st)
            torch.cuda.synchronize()
            w = (time.perf_counter() - t0) / N * 1e3
            print(f"{t.shape[0]:9d} triangles entry_bvh={i.entry_bvh} cost {i.cost_reference_tree:.0f}/{i.cost_entry_bvh:.0f} {W}x{H}x{fpc} {kw}: {w:7.2f} ms  {W*H*fpc/w/1e3:7.1f} Mpaths/s", flush=True)
