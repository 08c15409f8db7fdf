"""Synthetic large scene: build once (cached under --cache-dir), then time a few launches and print the
traversal statistics.  Meant to be run several times in one gpurun call (plain, under rocprofv3 --pmc, ...)."""
import argparse, os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import numpy as np, torch
import ray_tracing_amd as rt
from ray_tracing_amd import scenes

ap = argparse.ArgumentParser()
ap.add_argument("--objects", type=int, default=256)
ap.add_argument("--subdiv", type=int, default=100)
ap.add_argument("--width", type=int, default=1920)
ap.add_argument("--height", type=int, default=1080)
ap.add_argument("--spp", type=int, default=4)
ap.add_argument("--launches", type=int, default=2)
ap.add_argument("--cache-dir", default="/tmp")
ap.add_argument("--stats", action="store_true")
a = ap.parse_args()
t0 = time.time()
tris, nodes = scenes.synthetic_scene_cached(a.cache_dir, a.objects, a.subdiv)
print(f"scene: {tris.shape[0]} triangles, {nodes.shape[0]} nodes, {time.time() - t0:.1f} s", flush=True)
hdr, cache = scenes.hdr_and_cache()
sc = rt.Scene(tris, nodes, hdr, cache, device=0)
eye, cam = rt.camera(0.0, 20.0, 4.0)
d = rt.make_desc(a.width, a.height, mode=5, eye=eye, camera_rotate=cam)
acc = torch.zeros((a.height, a.width, 3), dtype=torch.float32, device="cuda:0")
st = torch.cuda.current_stream().cuda_stream
sc.render_device(d, 0, a.spp, acc.data_ptr(), st)
torch.cuda.synchronize()
rt.kernel_time_ms(reset=True)
for i in range(a.launches):
    sc.render_device(d, (i + 1) * a.spp, a.spp, acc.data_ptr(), st)
torch.cuda.synchronize()
ms, n = rt.kernel_time_ms(reset=True)
paths = a.width * a.height * a.spp
print(f"render_kernel {ms:.2f} ms per {a.spp}-frame launch ({n} launches): {paths / ms / 1e3:.1f} Mpaths/s", flush=True)
if a.stats:
    s = sc.render_stats(d, a.spp, a.spp)
    r = max(s["rays"], 1)
    print({k: (v / r if k not in ("paths", "rays") else v) for k, v in s.items()}, f"Mrays/s {s['rays'] / ms / 1e3:.1f}", flush=True)
