"""Pass statistics of the executed walk on the 10 M-triangle scene (RT_AMD_DIAG prints them from the statistics build)."""
import os, sys
os.environ["RT_AMD_DIAG"] = "1"
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import ray_tracing_amd as rt
from ray_tracing_amd import scenes
hdr, cache = scenes.hdr_and_cache()
t, n = scenes.synthetic_scene_cached("/tmp", 256, 100)
sc = rt.Scene(t, n, hdr, cache, device=0)
eye, cam = rt.camera(0.0, 20.0, 4.0)
d = rt.make_desc(1920, 1080, mode=5, eye=eye, camera_rotate=cam)
for kw in ({}, {"leaf_lanes": 8, "shade_lanes": 48}):
    sc.set_tuning(); sc.set_tuning(**kw) if kw else None
    print("tuning", kw, flush=True)
    s = sc.render_stats(d, 0, 4, executed=True)
    print({k: round(v / max(s["rays"], 1), 2) for k, v in s.items()}, flush=True)
