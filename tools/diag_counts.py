"""Prints the per-pass / per-shade-step lane occupancy of the path loop (RT_AMD_DIAG, executed walk) on config 5, or with --syn on
the 10 M-triangle synthetic scene."""
import os, sys
os.environ["RT_AMD_DIAG"] = "1"
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import ray_tracing_amd as rt
from ray_tracing_amd import scenes
syn = "--syn" in sys.argv
t, n = scenes.synthetic_scene_cached("/tmp", 256, 100) if syn else scenes.chap5_scene()
hdr, cache = scenes.hdr_and_cache()
sc = rt.Scene(t, n, hdr, cache, device=0)
eye, cam = rt.camera(0.0, 20.0, 4.0)
d = rt.make_desc(1920, 1080, mode=5, eye=eye, camera_rotate=cam)
print(sc.render_stats(d, 0, 2 if syn else 8, executed=True))
