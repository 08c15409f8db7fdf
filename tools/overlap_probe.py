"""Config 5: 64-frame launches back to back, with and without overlapping consecutive launches on the two internal streams."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import torch
import ray_tracing_amd as rt
from ray_tracing_amd import scenes
t, n = scenes.chap5_scene()
hdr, cache = scenes.hdr_and_cache()
sc = rt.Scene(t, n, hdr, cache, device=0)
eye, cam = rt.camera(0.0, 20.0, 4.0)
W, H = 1920, 1080
d = rt.make_desc(W, H, mode=5, eye=eye, camera_rotate=cam)
acc = torch.zeros((H, W, 3), dtype=torch.float32, device="cuda:0")
st = torch.cuda.current_stream().cuda_stream
for lf, ov in ((64, 8), (64, 1025), (32, 1025), (16, 1025)):
    sc.set_tuning(launch_frames=lf, overlap_frames=ov)
    sc.render_device(d, 0, 128, acc.data_ptr(), st); torch.cuda.synchronize()
    t0 = time.perf_counter()
    sc.render_device(d, 128, 1024, acc.data_ptr(), st)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f"launch_frames {lf} overlap_frames {ov}: {W*H*1024/dt/1e6:8.1f} Mpaths/s", flush=True)
