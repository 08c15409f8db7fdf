#!/usr/bin/env python3
"""bench.py — Mpaths/s of the per-pixel path loop on MI355X (BASELINE.json metric).

A "step" is one pass of the hot path over one batch: `--spp` frames (samples per pixel) of the whole image on every
rank.  N=1 workload = BASELINE configs[3]: path_tracing_5chap scene (nanosuit + floor slab, 19 070 triangles),
Sobol + BRDF/env importance sampling + MIS, 1920x1080, bundled 1024x512 HDR.  N>1: the scene is replicated, each rank
renders its own frameCounter range into a private fp32 sum buffer and the buffers are combined with ONE RCCL reduce per
step (spp split); rank 0 adds every reduced buffer to the running total and divides by the total spp at the
end, so the run produces the image.  Per-GPU work is fixed, so scaling is weak.

Usage: python bench.py --gpus N --steps K --warmup W
  * N > 1 without a torch.distributed environment: this process starts its own N ranks (python -m torch.distributed.run,
    one per GPU, rendezvous on 127.0.0.1) BEFORE anything touches the GPU, relays their output and exits with their code.
  * Under torch.distributed.run (RANK / WORLD_SIZE set) it is one rank.
  * N > 1 verifies by default (outside the timed region): rank 0 renders the same frames alone and compares the N-rank
    total with it (rtol 1e-4, reported as "verify" in the JSON line); --no-verify skips, --verify forces it at N = 1 too,
    --strict-verify turns a failed check into exit status 3.
"""
import argparse
import importlib
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

PEAK_VALU_TFLOPS_NO_FMA = 78.6   # 157.3 TF fp32 vector peak (MI355X_MICROARCH.md) / 2: contraction is off for parity
PEAK_HBM_GBS = 8000.0
SPEC_CLOCK_MHZ = 2400.0
N_SIMD = 1024                    # 256 CUs x 4 SIMD-32; a wave64 VALU instruction issues over 2 cycles
SHADE_FLOPS_PER_BOUNCE = 400.0


def _profile_entry(key):
    """Committed rocprofv3 PMC results for a workload (profiles/hbm_traffic.json, written by tools/summarize_profiles.py
    from the separate --pmc passes of tools/profile_round.sh); {} when that workload was not profiled."""
    try:
        with open(os.path.join(ROOT, "profiles", "hbm_traffic.json")) as f:
            return json.load(f).get(key, {})
    except (OSError, ValueError):
        return {}


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=64)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--spp", type=int, default=64, help="frames (samples per pixel) per step per GPU (default: one launch, rt_tuning.launch_frames)")
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--mode", type=int, default=5, choices=[3, 4, 5])
    ap.add_argument("--scene", default="chap5", choices=["chap5", "chap4", "small", "synthetic"])
    ap.add_argument("--synthetic-objects", type=int, default=256)
    ap.add_argument("--synthetic-subdiv", type=int, default=100)
    ap.add_argument("--cache-dir", default="/tmp", help="where the synthetic scene's BVH is cached between runs")
    ap.add_argument("--up-angle", type=float, default=20.0)
    ap.add_argument("--single-kernel", action="store_true", help="rt_tuning.single_kernel = 1: the one-kernel form of the path loop")
    ap.add_argument("--no-overlap", action="store_true", help="rt_tuning.overlap_frames = 1: every launch on its own (profiling passes: a kernel's "
                                                              "duration in a trace then is its own)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-stats", action="store_true")
    ap.add_argument("--verify", action="store_true", help="check the accumulated image against a 1-GPU render of the same frames")
    ap.add_argument("--no-verify", action="store_true")
    ap.add_argument("--strict-verify", action="store_true", help="exit with status 3 when the check fails")
    ap.add_argument("--cpu-sample", default="1920x1080x128", help="WxHxSPP sample of the same workload for the CPU oracle")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="gloo = CPU rehearsal of the launcher / partition / reduce / verify path (needs --render-plugin)")
    ap.add_argument("--render-plugin", default=None,
                    help="module:function giving a stand-in renderer for the gloo rehearsal (tests/ supplies one); never used on GPUs")
    return ap.parse_args(argv)


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def launch_ranks(args):
    """--gpus N without a torch.distributed environment: start N ranks of this script and hand their exit code on.  Nothing
    in this process has touched the GPU (no torch.cuda call, no library call), and the ranks are child processes."""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", str(max(1, min(os.cpu_count() or 1, 16) // max(args.gpus, 1))))
    return subprocess.run(cmd, env=env).returncode


class GpuRenderer:
    """The product path: one replicated scene per rank, rt_render_device into device buffers."""

    def __init__(self, args, local_rank, multi, rank=0, world=1):
        import torch

        import ray_tracing_amd as rt
        from ray_tracing_amd import scenes
        self.torch, self.rt = torch, rt
        if args.scene == "chap5":
            tris, nodes = scenes.chap5_scene()
        elif args.scene == "chap4":
            tris, nodes = scenes.chap4_scene()
        elif args.scene == "small":
            tris, nodes = scenes.small_scene()
        else:
            # N > 1: rank 0 builds (or loads) and saves, the others wait for the cache file and load it — before any GPU call
            tris, nodes = scenes.synthetic_scene_cached(args.cache_dir, args.synthetic_objects, args.synthetic_subdiv, rank=rank, world=world)
        self.tris, self.nodes = tris, nodes
        self.hdr, self.cache = scenes.hdr_and_cache()
        self.scene = rt.Scene(tris, nodes, self.hdr, self.cache, device=local_rank)
        if args.single_kernel:
            self.scene.set_tuning(single_kernel=1)
        if args.no_overlap:
            self.scene.set_tuning(overlap_frames=1)
        # orbit camera of c5:716-720 at r = 4 (c5:149), raised 20 degrees: at the start-up upAngle = 0 the eye sits under the
        # floor slab (y = 0.2) and sees only its underside, so the first thing a user does is drag the view up
        self.eye, self.cam = rt.camera(0.0, args.up_angle, 4.0)
        self.W, self.H, self.mode = args.width, args.height, args.mode
        self.dev = torch.device("cuda", local_rank)
        self.desc_sum = rt.make_desc(self.W, self.H, mode=args.mode, eye=self.eye, camera_rotate=self.cam, accumulate=rt.RT_ACCUM_SUM)
        self.desc_mean = rt.make_desc(self.W, self.H, mode=args.mode, eye=self.eye, camera_rotate=self.cam)
        self.multi = multi

    def new_buffer(self):
        return self.torch.zeros((self.H, self.W, 3), dtype=self.torch.float32, device=self.dev)

    def render(self, accum, frame_begin, frame_count, stream, sum_mode):
        self.scene.render_device(self.desc_sum if sum_mode else self.desc_mean, frame_begin, frame_count, accum.data_ptr(), stream)


def main():
    args = parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(launch_ranks(args))

    import numpy as np
    import torch

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    rehearsal = args.backend == "gloo"
    # RT_BENCH_FORCE_DIST=1: run the multi-GPU code path (sum buffers + reduce) with a single rank, to rehearse it on a 1-GPU box
    multi = world > 1 or os.environ.get("RT_BENCH_FORCE_DIST") == "1"
    dist = None
    from ray_tracing_amd import parallel

    if rehearsal:
        if not args.render_plugin:
            raise SystemExit("--backend gloo is a CPU rehearsal of the distributed path and needs --render-plugin module:function")
        mod, fn = args.render_plugin.split(":")
        plugin = getattr(importlib.import_module(mod), fn)(args, rank, world)
        import torch.distributed as dist
        dist.init_process_group(backend="gloo")
        W, H, spp = args.width, args.height, args.spp
        total = torch.zeros((H, W, 3), dtype=torch.float32)

        def step(i):
            fb, fc = parallel.frame_range(i, rank, world, spp)
            buf = torch.from_numpy(np.ascontiguousarray(plugin.render_sum(fb, fc), np.float32))
            parallel.reduce_sum(buf, dist, dst=0)
            if rank == 0 and i >= args.warmup:
                total.add_(buf)

        def fence():
            dist.barrier()

        renderer = None
    else:
        if not torch.cuda.is_available():
            raise SystemExit("bench.py needs a GPU (the path loop has no CPU fallback)")
        torch.cuda.set_device(local_rank)
        dev = torch.device("cuda", local_rank)
        if multi:
            import torch.distributed as dist
            if "RANK" not in os.environ:  # RT_BENCH_FORCE_DIST without a launcher: a one-rank group on this process
                os.environ.update({"MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(_free_port()), "RANK": "0", "WORLD_SIZE": "1"})
            dist.init_process_group(backend="nccl", device_id=dev)
        renderer = GpuRenderer(args, local_rank, multi, rank, world)
        rt = renderer.rt
        W, H, spp = args.width, args.height, args.spp
        # N = 1: one accumulation buffer (the running mean).  N > 1: two sum buffers, so that the RCCL reduce of step s (on its
        # own stream) overlaps the path loop of step s + 1 on the other buffer; rank 0 adds each reduced buffer to `total`.
        bufs = [renderer.new_buffer() for _ in range(2 if multi else 1)]
        total = renderer.new_buffer() if multi else bufs[0]
        if multi:
            # the first use of an operator loads its kernels (52 ms of host time for `add_`, during which the device sat idle too:
            # measured with RT_BENCH_TRACE=1): pay that here, not inside the first timed step, where only rank 0 would pay it
            total.add_(bufs[0])
            total.zero_()
        main_stream = torch.cuda.current_stream(dev)
        stream = main_stream.cuda_stream
        comm_stream = torch.cuda.Stream(dev) if multi else None
        reduced = [None, None]  # per buffer: event recorded when its last reduce (and the add into `total`) finished

        def step(i):
            # rank r of `world` owns frames [ (i*world + r)*spp, +spp ): disjoint frameCounter ranges (seed f5:320-323)
            fb, fc = parallel.frame_range(i, rank, world, spp)
            accum = bufs[i % len(bufs)]
            if multi:
                if reduced[i % 2] is not None:
                    main_stream.wait_event(reduced[i % 2])  # the reduce that last read this buffer
                accum.zero_()
            elif i == args.warmup:
                accum.zero_()  # the timed steps start a fresh running mean; frame weights stay 1/(frame+1) of their own index
            if not multi and i >= args.warmup:
                fb = (i - args.warmup) * spp  # N = 1: the timed steps are frames 0 .. steps*spp-1 of one running mean
            renderer.render(accum, fb, fc, stream, multi)
            if multi:
                rendered = torch.cuda.Event()
                rendered.record(main_stream)
                with torch.cuda.stream(comm_stream):
                    comm_stream.wait_event(rendered)
                    parallel.reduce_sum(accum, dist, dst=0)  # one RCCL reduce per step over xGMI
                    if rank == 0 and i >= args.warmup:
                        total.add_(accum)
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
    if renderer is not None:
        renderer.scene.kernel_timing(reset=True)
    t0 = time.perf_counter()
    trace = os.environ.get("RT_BENCH_TRACE") == "1"
    for i in range(args.steps):
        step(args.warmup + i)
        if trace:
            print(f"[trace] step {i} issued at {(time.perf_counter() - t0) * 1e3:.2f} ms", file=sys.stderr)
    if trace:
        torch.cuda.synchronize(dev)
        print(f"[trace] device idle at {(time.perf_counter() - t0) * 1e3:.2f} ms", file=sys.stderr)
    fence()
    elapsed = time.perf_counter() - t0
    if trace:
        print(f"[trace] fence done at {elapsed * 1e3:.2f} ms", file=sys.stderr)
    timing = renderer.scene.kernel_timing(reset=True) if renderer is not None else None
    if multi or rehearsal:
        t = torch.tensor([elapsed], dtype=torch.float64, device=None if rehearsal else dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    n_frames = args.steps * world * spp  # frames of the timed steps, all ranks
    paths = float(W) * H * n_frames
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
        "data": "reference's bundled scene + HDR (assets/); synthetic camera rays" if not rehearsal else "CPU rehearsal (stand-in renderer)",
        "world_size_seen": int(dist.get_world_size()) if dist is not None else 1,
        "timed_seconds": elapsed,
    }
    if renderer is not None:
        nt, nn = renderer.tris.shape[0], renderer.nodes.shape[0]
        names = {"chap5": "path_tracing_5chap scene", "chap4": "path_tracing_4chap scene", "small": "small test scene",
                 "synthetic": "synthetic tessellated scene"}
        result["config"] = {
            "workload": f"{names[args.scene]} ({nt} triangles, {nn} nodes), mode chap{args.mode}, {W}x{H}, orbit camera r=4 "
                        f"up={args.up_angle:g} deg, {spp} spp per step per GPU, maxBounce {4 if args.mode == 5 else 6}",
            "parallelism": f"spp-split x{world}, scene replicated, 1 RCCL reduce per step" if world > 1 else "single GPU"}
    else:
        result["config"] = {"workload": f"rehearsal {W}x{H}, {spp} spp per step per rank", "parallelism": f"spp-split x{world} (gloo)"}

    # ---- the accumulated image and its check (outside the timed region) ------------------------------------------------------
    do_verify = (args.verify or (world > 1 and not args.no_verify)) and rank == 0
    if rank == 0 and rehearsal:
        mean = parallel.normalize(total.numpy().astype(np.float64), n_frames)
        result["image_mean"] = float(np.nanmean(mean))
        if do_verify:
            ref = plugin.render_sum(args.warmup * world * spp, n_frames).astype(np.float64) / n_frames
            ok = np.isfinite(ref) & np.isfinite(mean)
            err = float(np.max(np.abs(mean[ok] - ref[ok]) / (np.abs(ref[ok]) + 1e-4))) if ok.any() else 0.0
            result["verify"] = {"ok": bool(err <= 1e-4 and ok.mean() > 0.99), "max_rel_err": err, "frames": n_frames}
    elif rank == 0:
        f0 = args.warmup * world * spp if multi else 0
        if multi:
            mean = total / float(n_frames)
        else:
            mean = total  # the running mean of frames 0 .. n_frames-1
        result["image_mean"] = float(torch.nanmean(mean).item())
        if do_verify:
            # the same frames, alone: step-sized SUM blocks added in float64, so the check measures the N-rank pipeline (partition,
            # reduce, total), not the rounding of one long fp32 sum
            blk = renderer.new_buffer()
            ref = torch.zeros((H, W, 3), dtype=torch.float64, device=dev)
            for f in range(f0, f0 + n_frames, spp):
                blk.zero_()
                renderer.render(blk, f, min(spp, f0 + n_frames - f), stream, True)
                ref += blk.double()
            torch.cuda.synchronize(dev)
            ref = (ref / float(n_frames)).float()
            ok = torch.isfinite(ref) & torch.isfinite(mean)
            same_special = (torch.isnan(ref) & torch.isnan(mean)) | (ref == mean)  # NaN / inf pixels (f5:533's unclamped sqrt) must coincide
            rel = (mean[ok] - ref[ok]).abs() / (ref[ok].abs() + 1e-4)
            err = float(rel.max().item()) if rel.numel() else 0.0
            tol = 1e-4 if multi else 5e-4  # N = 1 compares the fp32 running mean (f5:955) with sum / n: a different association
            result["verify"] = {"ok": bool(err <= tol and bool((ok | same_special).all().item())), "max_rel_err": err, "tol": tol,
                                "frames": n_frames, "non_finite_pixels": int((~ok).any(-1).sum().item())}

    if rank == 0 and renderer is not None:
        kern_ms, launches = timing.avg_ms, timing.launches
        first_ms = float(getattr(timing, "first_avg_ms", 0.0))  # two-kernel form: the camera-ray kernel that precedes the path loop
        clock = timing.shader_clock_mhz
        # Small launches (rt_scene_launch_shape: `overlapped`) overlap their neighbours (two internal streams): the timed region
        # above measures that pipeline, but a kernel's event-bracketed duration then covers a window it shares with the next
        # launch.  The roofline needs the kernel's own duration: a few extra steps with overlap off, outside the timed region.
        tun = renderer.scene.tuning()
        shape = renderer.scene.launch_shape(renderer.desc_mean, spp)  # what rt_tuning leaves to the library for this image and call
        overlapped = bool(shape.overlapped)
        overlapped_ms = None
        if overlapped and not args.no_stats:
            overlapped_ms = {"path_loop": kern_ms, "first_kernel": first_ms}
            renderer.scene.set_tuning(overlap_frames=1)
            scratch_buf = renderer.new_buffer()
            for i in range(4):
                renderer.render(scratch_buf, i * spp, spp, stream, multi)
                if i == 0:
                    torch.cuda.synchronize(dev)
                    renderer.scene.kernel_timing(reset=True)
            torch.cuda.synchronize(dev)
            alone = renderer.scene.kernel_timing(reset=True)
            renderer.scene.set_tuning(overlap_frames=int(tun.overlap_frames))
            kern_ms, first_ms, launches = alone.avg_ms, float(getattr(alone, "first_avg_ms", 0.0)), alone.launches
            timing, clock = alone, (alone.shader_clock_mhz or clock)
        key = f"{args.scene}_chap{args.mode}_{W}x{H}_spp{spp}"
        prof = _profile_entry(key)
        if not args.no_stats:
            # untimed statistics passes over a few frames: (a) the walk the timed kernels EXECUTE (culling, env-visibility
            # rays stop at the first hit) -> the work in `achieved`; (b) the reference's un-culled walk -> SURVEY §8d's
            # algorithmic fetch pattern, reported beside it; (c) the camera rays' share of (a): the first kernel's work
            fs = min(spp, 4)
            ex = renderer.scene.render_stats(renderer.desc_mean, 0, fs, executed=True)
            rf = renderer.scene.render_stats(renderer.desc_mean, 0, fs)
            cam = None
            if first_ms > 0.0:
                cam = renderer.scene.render_stats_camera(renderer.desc_mean, 0, fs, executed=True)
            rays_per_path = ex["rays"] / max(ex["paths"], 1)
            I, Lf, T, Hh = (ex[k] / max(ex["rays"], 1) for k in ("internal_nodes", "leaf_nodes", "triangles_tested", "hit_updates"))
            rI, rLf, rT, rH = (rf[k] / max(rf["rays"], 1) for k in ("internal_nodes", "leaf_nodes", "triangles_tested", "hit_updates"))
            lf = max(1, int(shape.launch_frames))  # frames per kernel launch
            paths_per_launch = float(W) * H * min(spp, lf)
            launches_per_step = max(1, (spp + lf - 1) // lf)
            scale = paths_per_launch / max(ex["paths"], 1)  # statistics frames -> one launch

            def walk_flops(st):  # §8d: 50 per internal node + 90 per triangle test + 40 per closest-hit improvement
                return 50.0 * st["internal_nodes"] + 90.0 * st["triangles_tested"] + 40.0 * st["hit_updates"]

            # every bounce emits at most one env-visibility and one BRDF ray, so bounces >= (rays - paths) / 2 (a lower bound): 400 each
            bounces = max(ex["rays"] - ex["paths"], 0) / 2.0
            flops_total = (walk_flops(ex) + SHADE_FLOPS_PER_BOUNCE * bounces) * scale
            flops_first = walk_flops(cam) * scale if cam else 0.0     # camera rays: traversal only (§8d prices no shading before a bounce)
            flops_loop = flops_total - flops_first                        # the path-loop kernel: every other ray + all bounce shading

            def tf(fl, ms):
                return fl / (ms * 1e-3) / 1e12 if ms > 0 else 0.0

            tflops = tf(flops_loop, kern_ms)
            # bytes: the executed walk on the device layout (64-B node record, 48-B triangle) and the reference's pattern (§8d)
            exec_bytes_per_launch = ((64.0 * I + 48.0 * T) * rays_per_path + 24.0) * paths_per_launch
            ref_bytes_per_launch = ((144.0 * rI + 48.0 * rLf + 72.0 * rT + 72.0 * rH) * rays_per_path) * paths_per_launch + 24.0 * W * H
            # PMC figures of this workload from the committed profile — only while they still describe the kernel that was timed
            prof_ms = prof.get("kernel_ms_when_profiled")
            fresh = bool(prof) and prof_ms and kern_ms > 0 and abs(kern_ms - prof_ms) / prof_ms <= 0.03
            traffic = prof.get("bytes_per_launch") if fresh else None
            hbm_gbs = (traffic / (kern_ms * 1e-3) / 1e9) if (traffic and kern_ms > 0) else None
            scene_mb = (renderer.tris.nbytes + renderer.nodes.nbytes) / 1e6
            l2_resident = scene_mb < 32.0
            roof = {
                "kernel": f"render_kernel<chap{args.mode}{', SEG' if first_ms > 0 else ''}>", "kernel_ms": kern_ms, "kernel_ms_max": float(timing.max_ms), "launches": launches,
                "traffic": traffic,
                "profile_head": prof.get("git_head"), "profile_kernel_ms": prof_ms,
                "shader_clock_mhz": clock,
                "rays_per_path": rays_per_path, "I": I, "Lf": Lf, "T": T, "H": Hh,
                "flops_per_launch": flops_loop, "valu_tflops": tflops,
                "valu_frac": tflops / PEAK_VALU_TFLOPS_NO_FMA,
                "valu_frac_at_measured_clock": (tflops / (PEAK_VALU_TFLOPS_NO_FMA * clock / SPEC_CLOCK_MHZ)) if clock else None,
                "executed_bytes_per_launch": exec_bytes_per_launch,
                "reference_pattern_bytes_per_launch": ref_bytes_per_launch,
                "reference_walk": {"I": rI, "Lf": rLf, "T": rT, "H": rH},
                "hbm_gbs": hbm_gbs, "hbm_frac": (hbm_gbs / PEAK_HBM_GBS) if hbm_gbs else None,
            }
            if overlapped_ms is not None:
                roof["overlapped_launches"] = {"kernel_ms_in_the_timed_region": overlapped_ms,
                                               "note": "the timed steps are launches small enough to overlap (rt_scene_launch_shape): consecutive launches "
                                                       "overlap, so `value` is higher than 1 / kernel_ms implies; kernel_ms (and everything derived from "
                                                       "it here) is the kernel's own duration, from 3 launches with overlap off after the timed region"}
            if prof and not fresh:
                roof["profile_note"] = (f"committed PMC profile is of a {prof_ms} ms kernel (head {prof.get('git_head')}); the kernel timed here "
                                        f"takes {kern_ms:.2f} ms: traffic / issue figures dropped")
            if first_ms > 0.0 and cam:
                # the two-kernel form of the path loop: the camera-ray kernel beside the dominant one, and both together (the
                # figure to hold against earlier rounds' one-kernel roofline)
                cr = max(cam["rays"], 1)
                roof["first_kernel"] = {"kernel": f"first_kernel<chap{args.mode}>", "kernel_ms": first_ms, "flops_per_launch": flops_first,
                                        "valu_tflops": tf(flops_first, first_ms), "frac": tf(flops_first, first_ms) / PEAK_VALU_TFLOPS_NO_FMA,
                                        "I": cam["internal_nodes"] / cr, "T": cam["triangles_tested"] / cr, "H": cam["hit_updates"] / cr,
                                        "escaped_fraction": cam["paths"] / cr,
                                        "traffic": prof.get("first_kernel_bytes_per_launch") if fresh else None}
                both = tf(flops_total, kern_ms + first_ms)
                roof["both_kernels"] = {"ms": kern_ms + first_ms, "flops_per_launch": flops_total, "valu_tflops": both,
                                        "frac": both / PEAK_VALU_TFLOPS_NO_FMA}
            wi = prof.get("valu_wave_instructions_per_launch") if fresh else None
            if wi and kern_ms > 0 and clock:
                # issue view: VALU wave-instructions of one launch (PMC SQ_INSTS_VALU, committed profile of this command) x 2
                # cycles (wave64 on a SIMD-32) / (kernel time x measured clock x 1024 SIMDs)
                roof["valu_issue_frac"] = wi * 2.0 / (kern_ms * 1e-3 * clock * 1e6 * N_SIMD)
                roof["valu_lane_util"] = prof.get("valu_lane_util")
            if l2_resident:
                roof.update({"bound": "valu", "achieved": tflops, "peak": PEAK_VALU_TFLOPS_NO_FMA, "unit": "TFLOP/s",
                             "frac": tflops / PEAK_VALU_TFLOPS_NO_FMA,
                             "note": f"scene arrays {scene_mb:.1f} MB are L2-resident: fp32 VALU is the binding resource (no FMA contraction: "
                                     f"peak = 157.3/2 TF). achieved = SURVEY 8d flops of the walk the timed kernel executes "
                                     f"({launches_per_step} launch(es) per step; the camera rays' share belongs to first_kernel). "
                                     f"HBM side: traffic / kernel time = hbm_gbs"})
            else:
                # a scene far beyond the caches: memory side binds.  achieved = measured traffic when this workload was profiled,
                # else the executed walk's bytes on the device layout (each record fetched once per visit)
                use = traffic if traffic else exec_bytes_per_launch
                gbs = use / (kern_ms * 1e-3) / 1e9 if kern_ms > 0 else 0.0
                roof.update({"bound": "hbm", "achieved": gbs, "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": gbs / PEAK_HBM_GBS,
                             "note": f"scene arrays {scene_mb:.0f} MB exceed L2 + Infinity Cache: random 64-B node / 48-B triangle gathers "
                                     f"bind (latency). achieved = {'PMC traffic (2 x FETCH_SIZE + WRITE_SIZE)' if traffic else 'bytes of the executed walk on the device layout'} / kernel time"})
            result["roofline"] = roof
            result["mrays_per_s_per_gpu"] = result["value"] * rays_per_path / world
        if not args.no_cpu_baseline and world == 1:  # a reported baseline, timed on rank 0 at N = 1 only
            import oracle_lib as O
            cw, ch, cs = (int(x) for x in args.cpu_sample.split("x"))
            cores = int(os.environ.get("OMP_NUM_THREADS", "0")) or min(os.cpu_count() or 1, 16)
            tc = time.perf_counter()
            O.render(renderer.tris, renderer.nodes, renderer.hdr, renderer.cache, cw, ch, args.mode, renderer.eye, renderer.cam, 0, cs)
            dt = time.perf_counter() - tc
            result["cpu_baseline"] = {"value": cw * ch * cs / dt / 1e6, "unit": "Mpaths/s", "cores": cores, "kind": "port",
                                      "sample": f"same scene/camera/mode at {cw}x{ch}, {cs} spp (oracle, OpenMP over rows), {dt:.1f} s"}
    if rank == 0:
        print(json.dumps(result), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    if args.strict_verify and rank == 0 and "verify" in result and not result["verify"]["ok"]:
        sys.exit(3)  # (by default a failed check is reported in the JSON line only: the measurement stays usable)


if __name__ == "__main__":
    main()
