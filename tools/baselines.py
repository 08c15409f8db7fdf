"""BASELINE.md section 3 baselines that bench.py's line does not carry (run on the GPU box: host cores + one MI355X):
  config 1  rt_chap1_render at 256x256 @ 4096 spp (host only): wall time, cores, Mpaths/s, and SURVEY 8d's statistical parity
            against the reference's own sample_4096.png (fixture tests/golden/chap1_sample4096_linear_64x64.npy: the PNG
            linearised with its gamma 1.9 and box-filtered 512 -> 64; the render is box-filtered 256 -> 64 the same way):
            relative mean <= 2 %, RMSE of the gamma-encoded values <= 0.03
  config 2  chapter-2 closest hit for the 1080p camera rays at r = 1.7 (model fills the frame) and r = 4: the CPU oracle
            (c2:490-513 restated) on 1 thread and on all cores, beside primary_kernel
usage: python tools/baselines.py [--skip-chap1]          (prints a log; commit it under profiles/)"""
import os, subprocess, sys, time
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..')
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np


def chap2_cpu():
    import oracle_lib as O
    from ray_tracing_amd import scenes
    tris, nodes = scenes.chap2_scene()
    ident = np.eye(4, dtype=np.float32).reshape(16)
    for r in (1.7, 4.0):
        eye = np.array([0, 0, r], np.float32)
        O.primary_hits(tris, nodes, 480, 270, eye, ident)
        t0 = time.perf_counter()
        tri, _ = O.primary_hits(tris, nodes, 1920, 1080, eye, ident)
        dt = time.perf_counter() - t0
        print(f"config 2 CPU oracle, OMP_NUM_THREADS={os.environ.get('OMP_NUM_THREADS')}: r={r}: {dt * 1e3:8.1f} ms = {1920 * 1080 / dt / 1e6:8.2f} Mrays/s, hits {(tri >= 0).sum()}", flush=True)


if len(sys.argv) > 1 and sys.argv[1] == "--chap2-cpu":
    chap2_cpu()
    sys.exit(0)

import ray_tracing_amd as rt
cores = int(os.environ.get("OMP_NUM_THREADS", "0")) or min(os.cpu_count() or 1, 16)
print(f"host cores used: {cores} (os.cpu_count() = {os.cpu_count()})", flush=True)
if "--skip-chap1" not in sys.argv:
    W = H = 256; S = 4096
    t0 = time.perf_counter()
    img = rt.chap1_render(W, H, S, seed=1)
    dt = time.perf_counter() - t0
    print(f"config 1: rt_chap1_render {W}x{H} @ {S} spp: {dt:.1f} s on {cores} cores = {W * H * S / dt / 1e6:.2f} Mpaths/s", flush=True)
    ref = np.load(os.path.join(ROOT, "tests", "golden", "chap1_sample4096_linear_64x64.npy")).astype(np.float64)
    q = rt.chap1_to_rgb8(img).astype(np.float64)            # imshow(): gamma 1/1.9, truncation (c1:340-342)
    small = ((q / 255.0) ** 1.9).reshape(64, 4, 64, 4, 3).mean(axis=(1, 3))
    rel = abs(small.mean() - ref.mean()) / ref.mean()
    g = lambda x: x ** (1 / 1.9)
    rmse = float(np.sqrt(np.mean((g(small) - g(ref)) ** 2)))
    ok = rel <= 0.02 and rmse <= 0.03
    print(f"config 1 vs the reference's sample_4096.png (64x64 linear): relative mean {rel * 100:.2f} % (<= 2 %), per-channel "
          f"{(np.abs(small.mean((0, 1)) - ref.mean((0, 1))) / ref.mean((0, 1)) * 100).round(2).tolist()} %, gamma-encoded RMSE {rmse:.4f} (<= 0.03): "
          f"{'PASS' if ok else 'FAIL'}", flush=True)
    rt.write_png(os.path.join(ROOT, "gpurun_out", "chap1_256x256_4096spp.png") if os.path.isdir(os.path.join(ROOT, "gpurun_out")) else "/tmp/chap1.png", rt.chap1_to_rgb8(img))
for th in ("1", str(cores)):
    env = dict(os.environ); env["OMP_NUM_THREADS"] = th
    subprocess.run([sys.executable, os.path.abspath(__file__), "--chap2-cpu"], env=env, check=True)
try:
    import torch
    gpu = torch.cuda.is_available()
except Exception:
    gpu = False
if gpu:
    from ray_tracing_amd import scenes
    tris, nodes = scenes.chap2_scene()
    sc = rt.Scene(tris, nodes, device=0)
    W, H = 1920, 1080
    tri = torch.empty((H, W), dtype=torch.int32, device="cuda:0"); dist = torch.empty((H, W), dtype=torch.float32, device="cuda:0")
    st = torch.cuda.current_stream().cuda_stream
    for r in (1.7, 4.0):
        d = rt.make_desc(W, H, eye=(0, 0, r))
        for _ in range(3):
            sc.primary_hits_device(d, tri.data_ptr(), dist.data_ptr(), st)
        torch.cuda.synchronize(); rt.kernel_time_ms(reset=True)
        for _ in range(20):
            sc.primary_hits_device(d, tri.data_ptr(), dist.data_ptr(), st)
        torch.cuda.synchronize()
        ms, n = rt.kernel_time_ms(reset=True)
        print(f"config 2 primary_kernel (MI355X): r={r}: {ms:.4f} ms per 1080p query = {W * H / ms / 1e3:.0f} Mrays/s, hits {(tri >= 0).sum().item()}", flush=True)
