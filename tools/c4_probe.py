"""Config 4 at its start-up camera (97 % of the camera rays escape): path-loop time against rt_tuning.unit_frames."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import numpy as np, torch
import ray_tracing_amd as rt
from ray_tracing_amd import scenes
hdr, cache = scenes.hdr_and_cache()
W, H = 1920, 1080
st = torch.cuda.current_stream().cuda_stream
acc = torch.zeros((H, W, 3), dtype=torch.float32, device="cuda:0")
t, n = scenes.chap4_scene()
sc = rt.Scene(t, n, hdr, cache, device=0)
for (r, label) in ((4.0, "start-up camera r=4"), (1.25, "sphere fills the frame r=1.25")):
    eye, cam = rt.camera(0.0, 0.0, r)
    d = rt.make_desc(W, H, mode=4, eye=eye, camera_rotate=cam)
    for single in (0, 1):
        for uf in (2, 8, 32):
            sc.set_tuning(unit_frames=uf, single_kernel=single)
            sc.render_device(d, 0, 64, acc.data_ptr(), st)
            torch.cuda.synchronize()
            sc.kernel_timing(reset=True)
            t0 = time.perf_counter()
            for i in range(5):
                sc.render_device(d, (i + 1) * 64, 64, acc.data_ptr(), st)
            torch.cuda.synchronize()
            wall = (time.perf_counter() - t0) / 5 * 1e3
            kt = sc.kernel_timing(reset=True)
            print(f"{label:32s} single={single} unit_frames={uf:2d}: path loop {kt.avg_ms:7.3f} ms  camera rays {kt.first_avg_ms:6.3f} ms  wall {wall:7.3f} ms  {W*H*64/wall/1e3:9.1f} Mpaths/s", flush=True)
    ex = sc.render_stats(d, 0, 2, executed=True)
    print("   executed walk:", {k: round(v / max(ex['rays'], 1), 2) for k, v in ex.items() if k not in ('paths', 'rays')}, "rays/path", round(ex['rays'] / ex['paths'], 3), flush=True)
