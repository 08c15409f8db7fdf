"""Times rt_scene_create on the 10 M-triangle synthetic scene (host preparation + uploads).  python tools/time_scene_create.py"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import ray_tracing_amd as rt
from ray_tracing_amd import scenes
t0 = time.time()
t, n = scenes.synthetic_scene_cached("/tmp", 256, 100)
print(f"scene arrays: {time.time() - t0:.2f} s ({t.shape[0]} triangles)", flush=True)
t0 = time.time()
t2, n2 = scenes.synthetic_scene_cached("/tmp", 256, 100)
print(f"scene arrays from the on-disk cache: {time.time() - t0:.2f} s", flush=True)
hdr, cache = scenes.hdr_and_cache()
for k in range(2):
    t0 = time.time()
    sc = rt.Scene(t, n, hdr, cache, device=0)
    print(f"rt_scene_create: {time.time() - t0:.2f} s  entry_bvh={sc.info().entry_bvh}", flush=True)
    del sc
