#!/usr/bin/env python3
"""bench.py — Mpaths/s of the per-pixel path loop on MI355X (BASELINE.json metric).

A "step" is one pass of the hot path over one batch: `--spp` frames (samples per pixel) of the whole image on every
rank.  N=1 workload = BASELINE configs[3]: path_tracing_5chap scene (nanosuit + floor slab, 19 070 triangles),
Sobol + BRDF/env importance sampling + MIS, 1920x1080, bundled 1024x512 HDR.  N>1: the scene is replicated, each rank
renders its own frameCounter range into a private fp32 sum buffer and the buffers are combined with ONE RCCL reduce per
step (spp split); per-GPU work is fixed, so scaling is weak.

Usage: python bench.py --gpus N --steps K --warmup W   (N>1 under torch.distributed.run, one rank per GPU)
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def _pmc_entry(key):
    try:
        with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "hbm_traffic.json")) as f:
            return json.load(f).get(key, {})
    except (OSError, ValueError):
        return {}


def _pmc_traffic(key):
    """HBM bytes per launch of the path-loop kernel from the committed rocprofv3 PMC passes (FETCH_SIZE + WRITE_SIZE, in
    bytes; collected by tools/profile_round.sh, stored by hand in profiles/hbm_traffic.json).  None when this workload
    was not profiled."""
    try:
        with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "hbm_traffic.json")) as f:
            return json.load(f).get(key, {}).get("bytes_per_launch")
    except (OSError, ValueError):
        return None


def _valu_block(flops_per_launch, kern_ms, wave_insts):
    """The roofline that does bind this kernel: fp32 VALU issue.  `issue_frac` = VALU wave-instructions of one launch (PMC
    SQ_INSTS_VALU, committed profile) x the measured issue ceiling of dependent fp32 VALU code at this occupancy (2.6 cycles
    per wave-instruction per SIMD, tools/ubench/valu_rate.hip) / (kernel time x 2.4 GHz x 1024 SIMDs)."""
    out = {"achieved_tflops": flops_per_launch / (kern_ms * 1e-3) / 1e12 if kern_ms > 0 else 0.0, "peak_tflops_no_fma": 78.6}
    if wave_insts and kern_ms > 0:
        simd_cycles = kern_ms * 1e-3 * 2.4e9 * 1024.0
        out.update({"wave_instructions_per_launch": wave_insts, "issue_cycles_per_instruction": simd_cycles / wave_insts,
                    "issue_ceiling_cycles_per_instruction": 2.6, "issue_frac": wave_insts * 2.6 / simd_cycles})
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=4)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--spp", type=int, default=32, help="frames (samples per pixel) per step per GPU")
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--mode", type=int, default=5, choices=[3, 4, 5])
    ap.add_argument("--scene", default="chap5", choices=["chap5", "chap4", "small", "synthetic"])
    ap.add_argument("--synthetic-objects", type=int, default=64)
    ap.add_argument("--synthetic-subdiv", type=int, default=48)
    ap.add_argument("--up-angle", type=float, default=20.0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-stats", action="store_true")
    ap.add_argument("--cpu-sample", default="1920x1080x128", help="WxHxSPP sample of the same workload for the CPU oracle")
    args = ap.parse_args()

    import numpy as np
    import torch

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the path loop has no CPU fallback)")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    # RT_BENCH_FORCE_DIST=1: run the multi-GPU code path (sum buffers + reduce) with a single rank, to rehearse it on a 1-GPU box
    multi = world > 1 or os.environ.get("RT_BENCH_FORCE_DIST") == "1"
    if multi:
        import torch.distributed as dist
        dist.init_process_group(backend="nccl", device_id=dev)

    import ray_tracing_amd as rt
    from ray_tracing_amd import parallel, scenes

    if args.scene == "chap5":
        tris, nodes = scenes.chap5_scene()
    elif args.scene == "chap4":
        tris, nodes = scenes.chap4_scene()
    elif args.scene == "small":
        tris, nodes = scenes.small_scene()
    else:
        tris, nodes = scenes.synthetic_scene(args.synthetic_objects, args.synthetic_subdiv)
    hdr, cache = scenes.hdr_and_cache()
    scene = rt.Scene(tris, nodes, hdr, cache, device=local_rank)
    # orbit camera of c5:716-720 at r = 4 (c5:149), raised 20 degrees: at the start-up upAngle = 0 the eye sits under the
    # floor slab (y = 0.2) and sees only its underside, so the first thing a user does is drag the view up
    eye, cam = rt.camera(0.0, args.up_angle, 4.0)
    W, H, spp = args.width, args.height, args.spp
    acc_mode = rt.RT_ACCUM_SUM if multi else rt.RT_ACCUM_RUNNING_MEAN
    desc = rt.make_desc(W, H, mode=args.mode, eye=eye, camera_rotate=cam, accumulate=acc_mode)

    # N = 1: one accumulation buffer (the running mean).  N > 1: two sum buffers, so that the RCCL reduce of step s (on its
    # own stream) overlaps the path loop of step s + 1 on the other buffer.
    bufs = [torch.zeros((H, W, 3), dtype=torch.float32, device=dev) for _ in range(2 if multi else 1)]
    main_stream = torch.cuda.current_stream(dev)
    stream = main_stream.cuda_stream
    comm_stream = torch.cuda.Stream(dev) if multi else None
    reduced = [None, None]  # per buffer: event recorded when its last reduce finished

    def step(i):
        # rank r of `world` owns frames [ (i*world + r)*spp, +spp ): disjoint frameCounter ranges (seed f5:320-323)
        fb, fc = parallel.frame_range(i, rank, world, spp)
        accum = bufs[i % len(bufs)]
        if multi:
            if reduced[i % 2] is not None:
                main_stream.wait_event(reduced[i % 2])  # the reduce that last read this buffer
            accum.zero_()
        scene.render_device(desc, fb, fc, accum.data_ptr(), stream)
        if multi:
            rendered = torch.cuda.Event()
            rendered.record(main_stream)
            with torch.cuda.stream(comm_stream):
                comm_stream.wait_event(rendered)
                parallel.reduce_sum(accum, dist, dst=0)  # one RCCL reduce per step over xGMI
                reduced[i % 2] = torch.cuda.Event()
                reduced[i % 2].record(comm_stream)

    def fence():
        torch.cuda.synchronize(dev)
        if multi:
            dist.barrier()
        torch.cuda.synchronize(dev)

    for i in range(args.warmup):
        step(i)
    fence()
    rt.kernel_time_ms(reset=True)
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(args.warmup + i)
    fence()
    elapsed = time.perf_counter() - t0
    kern_ms, launches = rt.kernel_time_ms(reset=True)
    if multi:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    paths = float(W) * H * spp * args.steps * world
    result = {
        "metric": "Mpaths/s (whole node, 1080p Disney+MIS)",
        "value": paths / elapsed / 1e6,
        "unit": "Mpaths/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": elapsed / max(args.steps, 1) * 1e3,
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "f32",
        "data": "reference's bundled scene + HDR (assets/); synthetic camera rays",
        "config": {"workload": f"path_tracing_5chap scene ({tris.shape[0]} triangles, {nodes.shape[0]} nodes), mode chap{args.mode}, "
                               f"{W}x{H}, orbit camera r=4 up={args.up_angle:g} deg, {spp} spp per step per GPU, maxBounce {4 if args.mode == 5 else 6}"
                   if args.scene == "chap5" else f"{args.scene} scene ({tris.shape[0]} triangles), mode chap{args.mode}, {W}x{H}, {spp} spp/step/GPU",
                   "parallelism": f"spp-split x{world}, scene replicated, 1 RCCL reduce per step" if world > 1 else "single GPU"},
    }

    if rank == 0:
        if not args.no_stats:
            # untimed statistics pass over one step's frames: rays/path and the traversal means (I, Lf, T, H) per ray
            st = scene.render_stats(desc, args.warmup * world * spp, spp)
            rays = max(st["rays"], 1)
            I, Lf, T, Hh = (st[k] / rays for k in ("internal_nodes", "leaf_nodes", "triangles_tested", "hit_updates"))
            rays_per_path = st["rays"] / max(st["paths"], 1)
            # algorithmic bytes per ray = the reference's fetch pattern: 144 B per internal visit, 48 B
            # per leaf node, 72 B per triangle tested, 72 B per closest-hit improvement; + per path: 24 B framebuffer
            bytes_per_ray = 144.0 * I + 48.0 * Lf + 72.0 * T + 72.0 * Hh
            flops_per_ray = 50.0 * I + 90.0 * T + 40.0 * Hh
            paths_per_launch = float(W) * H * spp
            bytes_per_launch = bytes_per_ray * rays_per_path * paths_per_launch + 24.0 * W * H
            achieved = bytes_per_launch / (kern_ms * 1e-3) / 1e9 if kern_ms > 0 else 0.0
            result["roofline"] = {
                "bound": "hbm", "achieved": achieved, "peak": 8000.0, "unit": "GB/s", "frac": achieved / 8000.0,
                "traffic": _pmc_traffic(f"{args.scene}_chap{args.mode}_{W}x{H}_spp{spp}"),
                "algorithmic_bytes_per_launch": bytes_per_launch,
                "kernel": f"render_kernel<chap{args.mode}>", "kernel_ms": kern_ms, "launches": launches,
                "bytes_per_ray": bytes_per_ray, "rays_per_path": rays_per_path,
                "per_ray": {"I": I, "Lf": Lf, "T": T, "H": Hh},
                "valu": _valu_block(flops_per_ray * rays_per_path * paths_per_launch, kern_ms,
                                    _pmc_entry(f"{args.scene}_chap{args.mode}_{W}x{H}_spp{spp}").get("valu_wave_instructions_per_launch")),
                "note": f"scene arrays {(tris.nbytes + nodes.nbytes) / 1e6:.1f} MB; the path loop is VALU-issue-bound, HBM is not the binding resource: see DESIGN.md section 5",
            }
            result["mrays_per_s_per_gpu"] = result["value"] * rays_per_path / world
        if not args.no_cpu_baseline and world == 1:  # a reported baseline, timed on rank 0 at N = 1 only
            import oracle_lib as O
            cw, ch, cs = (int(x) for x in args.cpu_sample.split("x"))
            cores = int(os.environ.get("OMP_NUM_THREADS", "0")) or min(os.cpu_count() or 1, 16)
            tc = time.perf_counter()
            O.render(tris, nodes, hdr, cache, cw, ch, args.mode, eye, cam, 0, cs)
            dt = time.perf_counter() - tc
            result["cpu_baseline"] = {"value": cw * ch * cs / dt / 1e6, "unit": "Mpaths/s", "cores": cores, "kind": "port",
                                      "sample": f"same scene/camera/mode at {cw}x{ch}, {cs} spp (oracle, OpenMP over rows), {dt:.1f} s"}
        print(json.dumps(result), flush=True)
    if multi:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
