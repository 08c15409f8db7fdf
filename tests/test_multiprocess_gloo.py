"""N > 1 path on CPU: world_size-2 gloo run of the spp-split + single reduce (ray_tracing_amd/parallel.py, the code
bench.py runs over RCCL).  The per-rank renderer here is the CPU oracle in SUM mode (test stand-in for rt_render_device,
which is bit-identical to it on a GPU: tests/test_gpu_parity.py::test_sum_mode_matches_oracle_and_mean)."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, outdir):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["OMP_NUM_THREADS"] = "2"
    import torch
    import torch.distributed as dist

    import oracle_lib as O
    import ray_tracing_amd as rt
    from ray_tracing_amd import parallel, scenes

    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    tris, nodes = scenes.small_scene()
    hdr, cache = scenes.hdr_and_cache()
    eye, cam = rt.camera(10.0, 12.0, 3.2)
    W, H, spp, steps = 40, 32, 3, 2
    total = torch.zeros((H, W, 3), dtype=torch.float32)
    for step in range(steps):
        fb, fc = parallel.frame_range(step, rank, world, spp)
        acc, _ = O.render(tris, nodes, hdr, cache, W, H, 5, eye, cam, fb, fc, accumulate=1)
        t = torch.from_numpy(acc)
        parallel.reduce_sum(t, dist, dst=0)          # ONE collective per step
        if rank == 0:
            total += t
    if rank == 0:
        mean = parallel.normalize(total, parallel.frames_covered(steps, world, spp)).numpy()
        np.save(os.path.join(outdir, "mean.npy"), mean)
    dist.barrier()
    dist.destroy_process_group()


def test_spp_split_two_ranks_matches_single_process(tmp_path):
    import torch.multiprocessing as mp

    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib as O
    import ray_tracing_amd as rt
    from ray_tracing_amd import scenes

    world = 2
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    got = np.load(tmp_path / "mean.npy")
    tris, nodes = scenes.small_scene()
    hdr, cache = scenes.hdr_and_cache()
    eye, cam = rt.camera(10.0, 12.0, 3.2)
    n = 2 * world * 3
    ref_sum, _ = O.render(tris, nodes, hdr, cache, 40, 32, 5, eye, cam, 0, n, accumulate=1)
    ref_mean, _ = O.render(tris, nodes, hdr, cache, 40, 32, 5, eye, cam, 0, n)
    # same samples, different fp32 summation order: equal to rounding, not to the bit
    assert np.allclose(got, ref_sum / n, rtol=1e-5, atol=1e-6)
    assert np.allclose(got, ref_mean, rtol=1e-4, atol=1e-5)      # and to the reference's running mean (f5:955)


def test_bench_launches_its_own_ranks_and_verifies(tmp_path):
    """`bench.py --gpus 2` without a torch.distributed environment starts its own 2 ranks (torch.distributed.run on 127.0.0.1),
    partitions the frames, reduces once per step, forms the image on rank 0, checks it against a single-process render of the
    same frames and prints ONE JSON line with n_gpus = 2 — here over gloo with the oracle as the stand-in renderer."""
    import json
    import subprocess

    env = dict(os.environ)
    env.pop("WORLD_SIZE", None), env.pop("RANK", None), env.pop("LOCAL_RANK", None)
    env["OMP_NUM_THREADS"] = "2"
    env["PYTHONPATH"] = os.pathsep.join([os.path.join(ROOT, "tests"), ROOT, env.get("PYTHONPATH", "")])
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--spp", "2",
                        "--width", "40", "--height", "32", "--backend", "gloo", "--render-plugin", "rehearsal_plugin:make", "--verify"],
                       env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["world_size_seen"] == 2 and out["scaling"] == "weak"
    assert out["steps"] == 2 and out["warmup"] == 1 and out["value"] > 0
    assert out["verify"]["ok"] and out["verify"]["frames"] == 2 * 2 * 2


def _shared_worker(rank, world, path, outdir):
    sys.path.insert(0, ROOT)
    os.environ["OMP_NUM_THREADS"] = "2"
    import time

    from ray_tracing_amd import scenes
    built = []

    def builder():
        built.append(rank)
        if rank == 0:
            time.sleep(1.0)  # the other rank is already polling for the file by now
        return scenes.synthetic_scene(3, 6)

    tris, nodes = scenes.shared_build(path, 0x5EED0001, builder, rank, world, wait_s=120.0, poll_s=0.05)
    np.save(os.path.join(outdir, f"tris{rank}.npy"), tris)
    np.save(os.path.join(outdir, f"nodes{rank}.npy"), nodes)
    with open(os.path.join(outdir, f"built{rank}.txt"), "w") as f:
        f.write(str(len(built)))


def test_one_scene_build_per_node_rank0_builds_the_others_load(tmp_path):
    """bench.py --gpus N (one process per GPU): rank 0 builds the scene and saves it to the on-disk cache, every other rank waits
    for the file and loads it (scenes.shared_build) — here with two processes and a stale cache file of another key in place."""
    import torch.multiprocessing as mp

    path = str(tmp_path / "scene.rtsc")
    sys.path.insert(0, ROOT)
    import ray_tracing_amd as rt
    from ray_tracing_amd import scenes
    t0, n0 = scenes.small_scene()
    rt.scene_cache_save(path, 0x0BADBEEF, t0, n0)  # left over by an earlier run: must not be picked up
    mp.spawn(_shared_worker, args=(2, path, str(tmp_path)), nprocs=2, join=True)
    assert (tmp_path / "built0.txt").read_text() == "1" and (tmp_path / "built1.txt").read_text() == "0"
    want_t, want_n = scenes.synthetic_scene(3, 6)
    for r in range(2):
        assert np.array_equal(np.load(tmp_path / f"tris{r}.npy").view(np.uint32), want_t.view(np.uint32))
        assert np.array_equal(np.load(tmp_path / f"nodes{r}.npy")[1:].view(np.uint32), want_n[1:].view(np.uint32))
    assert not [p for p in os.listdir(tmp_path) if ".tmp." in p]  # the temporary file was renamed into place
