"""One-off, full-size parity run: BASELINE config 5 (1920x1080, frames 0..1023, mode 5) or config 4 (frames 0..255, mode 4)
on the GPU against the CPU oracle on the same inputs, every float compared bitwise.  Takes minutes of host time."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'tests'))
import numpy as np
import oracle_lib as O
import ray_tracing_amd as rt
from ray_tracing_amd import scenes

cfg = int(sys.argv[1]) if len(sys.argv) > 1 else 5
chunk = int(sys.argv[2]) if len(sys.argv) > 2 else 128
W, H = 1920, 1080
if cfg == 5:
    tris, nodes = scenes.chap5_scene(); F = 1024; eye, cam = rt.camera(0.0, 20.0, 4.0)
else:
    tris, nodes = scenes.chap4_scene(); F = 256; eye, cam = rt.camera(0.0, 0.0, 4.0)
hdr, cache = scenes.hdr_and_cache()
sc = rt.Scene(tris, nodes, hdr, cache)
d = rt.make_desc(W, H, mode=cfg, eye=eye, camera_rotate=cam)
t0 = time.time(); img = sc.render(d, 0, F); tg = time.time() - t0
print(f"GPU: config {cfg}, {W}x{H} x {F} frames in {tg:.2f} s incl. host copies", flush=True)
ref = None
t0 = time.time()
for fb in range(0, F, chunk):  # the oracle in chunks, so that progress is visible; the running mean carries over exactly
    ref, _ = O.render(tris, nodes, hdr, cache, W, H, cfg, eye, cam, fb, min(chunk, F - fb), accum=ref)
    print(f"oracle: frames < {min(fb + chunk, F)} after {time.time() - t0:.0f} s", flush=True)
a, b = img.view(np.uint32), ref.view(np.uint32)
bad = (a != b) & ~(np.isnan(img) & np.isnan(ref))
print(f"config {cfg}: {bad.sum()} of {bad.size} floats differ; NaN pixels {int(np.isnan(ref).any(-1).sum())}; mean radiance {np.nanmean(ref):.6f}")
sys.exit(1 if bad.any() else 0)
