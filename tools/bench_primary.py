"""Config 2: chapter-2 closest-hit query for the 1080p camera rays; prints Mrays/s of primary_kernel."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import numpy as np, torch
import ray_tracing_amd as rt
from ray_tracing_amd import scenes

tris, nodes = scenes.chap2_scene()
sc = rt.Scene(tris, nodes, device=0)
W, H = 1920, 1080
for r in (4.0, 1.7):
    d = rt.make_desc(W, H, eye=(0, 0, r))
    tri = torch.empty((H, W), dtype=torch.int32, device="cuda:0")
    dist = torch.empty((H, W), dtype=torch.float32, device="cuda:0")
    for _ in range(3):
        sc.primary_hits_device(d, tri.data_ptr(), dist.data_ptr(), torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    rt.kernel_time_ms(reset=True)
    for _ in range(20):
        sc.primary_hits_device(d, tri.data_ptr(), dist.data_ptr(), torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    ms, n = rt.kernel_time_ms(reset=True)
    print(f"r={r}: hits {(tri >= 0).sum().item()}, {ms:.4f} ms per launch over {n} launches, {W * H / ms / 1e3:.0f} Mrays/s")
