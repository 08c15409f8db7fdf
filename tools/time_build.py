import time, sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import ray_tracing_amd as rt
from ray_tracing_amd import scenes
import ray_tracing_amd.scenes as S
n_obj, sub = int(sys.argv[1]), int(sys.argv[2])
# replicate synthetic_scene with timing around the BVH build
orig = rt.Mesh.buildBVHwithSAH
def timed(self, n):
    t0 = time.time(); r = orig(self, n); print(f"buildBVHwithSAH: {time.time()-t0:.2f} s", flush=True); return r
rt.Mesh.buildBVHwithSAH = timed
t0 = time.time()
t, n = S.synthetic_scene(n_obj, sub)
print(f"total synthetic_scene: {time.time()-t0:.2f} s, {t.shape[0]} triangles, {n.shape[0]} nodes", flush=True)
