"""Termination soak: many launches of random sizes / frame ranges / modes / scenes / tuning knobs; prints progress lines.
Every launch must drain its queue and return; the smaller device-buffer launches are also checked bit for bit against the same
frames rendered with default tuning through the host-buffer entry point (the oracle comparison itself is the parity tests' job)."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import numpy as np
import ray_tracing_amd as rt
from ray_tracing_amd import scenes

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
only = int(sys.argv[3]) if len(sys.argv) > 3 else -1  # replay: render only this launch (the random sequence is consumed as usual)
verbose = len(sys.argv) > 2  # print every launch's parameters before it starts (to find one that does not return)
rng = np.random.RandomState(12345)
hdr, cache = scenes.hdr_and_cache()
pool = [scenes.small_scene(), scenes.chap5_scene(), scenes.chap4_scene(), scenes.synthetic_scene(9, 10),
        (np.zeros((0, 36), np.float32), np.zeros((1, 12), np.float32))]
objs = [rt.Scene(t, n, hdr, cache) for t, n in pool] + [rt.Scene(t, n, hdr, cache, entry_bvh=1) for t, n in pool[:4]]
use_torch = True
try:
    import torch
    streams = [torch.cuda.current_stream(), torch.cuda.Stream(), torch.cuda.Stream()]
except Exception:  # noqa: BLE001
    use_torch = False
keep = []
t0 = time.time(); k = 0; paths = 0; checked = 0
while time.time() - t0 < budget:
    sc = objs[rng.randint(len(objs))]
    W, H = int(rng.randint(1, 700)), int(rng.randint(1, 400))
    F = int(rng.choice([1, 1, 2, 3, 7, 33, 70]))
    mode = int(rng.choice([3, 4, 5]))
    sc.set_tuning(launch_frames=rng.choice([0, 1, 2, 5, 32, 64]), unit_frames=rng.choice([0, 1, 2, 3, 8]),  # (0: the library's choice)
                  shade_lanes=rng.choice([0, 1, 16, 48, 64]), leaf_lanes=rng.choice([1, 8, 64]),
                  wait_budget=rng.choice([1, 100, 4000, 10**8]), overlap_frames=rng.choice([1, 8, 1025]),
                  no_cull=rng.choice([0, 0, 1]), no_entry_bvh=rng.choice([0, 0, 1]), single_kernel=rng.choice([0, 1, 2]))
    cam_args = (float(rng.uniform(0, 360)), float(rng.uniform(-89, 89)), float(rng.uniform(0.05, 8)))
    eye, cam = rt.camera(*cam_args)
    d = rt.make_desc(W, H, mode=mode, eye=eye, camera_rotate=cam, accumulate=int(rng.randint(2)))
    fb = int(rng.randint(0, 5000))
    if verbose:
        tn = sc.tuning()
        print(f"launch {k}: scene {objs.index(sc)} {W}x{H} F={F} fb={fb} mode={mode} acc={d.accumulate} camera={cam_args!r} tuning={[getattr(tn, f) for f, _ in tn._fields_]}", flush=True)
    if only >= 0 and k != only:
        k += 1
        if k > only:
            break
        continue
    how = int(rng.randint(5)) if use_torch else 0
    if how <= 1:
        img = sc.render(d, fb, F)                       # host buffer, the library's NULL stream
        assert img.shape == (H, W, 3)
    else:
        st = streams[int(rng.randint(len(streams)))]
        with torch.cuda.stream(st):
            acc = torch.zeros((H, W, 3), dtype=torch.float32, device="cuda:0")  # (zero fill on the stream that renders)
            if how == 2 or W * H * 12 > (40 << 20):
                sc.render_device(d, fb, F, acc.data_ptr(), st.cuda_stream)
                if W * H * F <= 2_000_000:  # image check: the same frames, default tuning, host-buffer entry point
                    torch.cuda.synchronize()
                    got = acc.cpu().numpy()
                    sc.set_tuning()
                    want = sc.render(d, fb, F)
                    same = (got.view(np.uint32) == want.view(np.uint32)) | (np.isnan(got) & np.isnan(want))
                    assert same.all(), f"launch {k}: {int((~same).sum())} floats differ from the default-tuning render"
                    checked += 1
            elif how == 3:
                try:
                    sc.render_device_noalloc(d, fb, min(F, 3), acc.data_ptr(), st.cuda_stream)
                except rt.RtError:
                    pass                                    # does not fit the static scratch array: a clean refusal
            else:
                sc.primary_hits(d)
        if rng.randint(4) == 0:
            torch.cuda.synchronize()
        keep.append(acc)
        if len(keep) > 8:
            torch.cuda.synchronize()
            keep.clear()
    k += 1; paths += W * H * F
    if k % 50 == 0:
        print(f"{k} launches, {paths / 1e6:.0f} Mpaths, {time.time() - t0:.0f} s", flush=True)
if use_torch:
    torch.cuda.synchronize()
print(f"soak ok: {k} launches ({checked} image-checked), {paths / 1e6:.0f} Mpaths in {time.time() - t0:.0f} s")
