"""-m gpu: rt_render_device_noalloc (static per-GPU scratch, no hipMalloc) against rt_render_device, through the C ABI.

Collected last on purpose (test_zz_*): the round-end run is `pytest -x`."""
import ctypes as C

import numpy as np
import pytest

import ray_tracing_amd as rt

pytestmark = pytest.mark.gpu


def _bits(t):
    return np.ascontiguousarray(t.cpu().numpy(), np.float32).view(np.uint32)


def _both(sc, desc, fb, fc, torch, dev):
    """(rt_render_device image, rt_render_device_noalloc image) for the same frames, both starting from the same lastFrame."""
    start = torch.rand((desc.height, desc.width, 3), dtype=torch.float32, device=dev)  # a non-trivial lastFrame to blend into
    a, b = start.clone(), start.clone()
    stream = torch.cuda.current_stream(dev).cuda_stream
    sc.render_device(desc, fb, fc, a.data_ptr(), stream)
    sc.render_device_noalloc(desc, fb, fc, b.data_ptr(), stream)
    torch.cuda.synchronize(dev)
    return a, b


def test_noalloc_equals_render_device_two_calls_second_smaller(scene_small, hdr_pair):
    import torch
    dev = torch.device("cuda", 0)
    tris, nodes = scene_small
    hdr, cache = hdr_pair
    sc = rt.Scene(tris, nodes, hdr, cache, device=0)
    eye, cam = rt.camera(20.0, 15.0, 3.5)
    # first call: a large input; second: a smaller one on the same GPU's scratch, which still holds the first call's queue
    # counters, Sobol pairs and samples.  launch_frames = 3 cuts the 40-frame call into 14 launches that reuse the one scratch
    # slot back to back on the caller's stream; the 1080p call is cut by the scratch itself (room for 2 frames of samples)
    for (W, H, mode, fb, fc, lf) in ((333, 187, 5, 3, 40, 3), (333, 187, 5, 3, 40, 64), (1920, 1080, 5, 1, 5, 64), (61, 45, 5, 0, 3, 64), (40, 24, 4, 7, 2, 64)):
        sc.set_tuning(launch_frames=lf)
        d = rt.make_desc(W, H, mode=mode, eye=eye, camera_rotate=cam)
        a, b = _both(sc, d, fb, fc, torch, dev)
        A, B = _bits(a), _bits(b)
        nan = np.isnan(a.cpu().numpy()) & np.isnan(b.cpu().numpy())
        assert ((A == B) | nan).all(), f"{W}x{H} mode {mode}: {int(((A != B) & ~nan).sum())} floats differ"
        assert float(torch.nan_to_num(b).abs().sum()) > 0.0


def test_noalloc_sum_mode_and_deep_tree_spill_from_scratch(hdr_pair):
    """A 30-deep hand-built chain needs the global stack spill: the no-alloc entry carves it out of the scratch array."""
    import torch
    dev = torch.device("cuda", 0)
    from scene_helpers import deep_chain
    tris, nodes = deep_chain(30, base_color=0.8, roughness=0.4)
    hdr, cache = hdr_pair
    sc = rt.Scene(tris, nodes, hdr, cache, device=0)
    d = rt.make_desc(48, 40, mode=5, eye=(0.0, 0.0, 2.0), accumulate=rt.RT_ACCUM_SUM)
    a, b = _both(sc, d, 0, 3, torch, dev)
    assert np.array_equal(_bits(a), _bits(b))


def test_noalloc_with_the_entry_bvh_and_interleaved_with_overlapped_calls(scene_small, hdr_pair):
    """A scene that walks the entry BVH (forced), rendered alternately through the allocating entry point (one-frame calls,
    overlapped on the internal streams) and the no-allocation one on the same caller stream: both accumulate the same frames."""
    import torch
    dev = torch.device("cuda", 0)
    tris, nodes = scene_small
    hdr, cache = hdr_pair
    sc = rt.Scene(tris, nodes, hdr, cache, device=0, entry_bvh=1)
    assert sc.info().entry_bvh == 1
    eye, cam = rt.camera(-30.0, 25.0, 3.0)
    d = rt.make_desc(120, 90, mode=5, eye=eye, camera_rotate=cam)
    stream = torch.cuda.current_stream(dev).cuda_stream
    a = torch.zeros((90, 120, 3), dtype=torch.float32, device=dev)
    b = torch.zeros_like(a)
    for f in range(9):
        sc.render_device(d, f, 1, a.data_ptr(), stream)
        sc.render_device_noalloc(d, f, 1, b.data_ptr(), stream)
    torch.cuda.synchronize(dev)
    assert np.array_equal(_bits(a), _bits(b))
    c = torch.zeros_like(a)
    sc.render_device_noalloc(d, 0, 9, c.data_ptr(), stream)
    torch.cuda.synchronize(dev)
    assert np.array_equal(_bits(a), _bits(c))


def test_noalloc_rejects_what_does_not_fit(scene_small, hdr_pair):
    import torch
    tris, nodes = scene_small
    hdr, cache = hdr_pair
    sc = rt.Scene(tris, nodes, hdr, cache, device=0)
    d = rt.make_desc(3840, 2160, mode=5)  # one frame of samples = 99.5 MB > the 64 MiB scratch
    buf = torch.zeros((4,), dtype=torch.float32, device="cuda:0")  # never written: the call must fail before any launch
    with pytest.raises(rt.RtError) as e:
        sc.render_device_noalloc(d, 0, 1, buf.data_ptr(), 0)
    assert e.value.code == -1 and "scratch" in str(e.value)


def test_noalloc_refuses_a_call_that_would_overlap_on_another_stream(scene_chap5, hdr_pair):
    """All calls on a GPU share its scratch array.  A call on another stream while the previous call's work is still in flight
    is refused (RT_ERR_INVALID_ARG) instead of racing; the same stream (ordered) and a later call (finished) are accepted, and
    the library allocates no device memory in any of them."""
    import torch
    dev = torch.device("cuda", 0)
    tris, nodes = scene_chap5
    hdr, cache = hdr_pair
    sc = rt.Scene(tris, nodes, hdr, cache, device=0)
    eye, cam = rt.camera(0.0, 20.0, 4.0)
    d = rt.make_desc(1920, 1080, mode=5, eye=eye, camera_rotate=cam)
    a = torch.zeros((1080, 1920, 3), dtype=torch.float32, device=dev)
    b = torch.zeros_like(a)
    s1, s2 = torch.cuda.Stream(dev), torch.cuda.Stream(dev)
    torch.cuda.synchronize(dev)
    free0 = torch.cuda.mem_get_info(dev)[0]
    sc.render_device_noalloc(d, 0, 16, a.data_ptr(), s1.cuda_stream)      # ~10 ms of GPU work in flight
    with pytest.raises(rt.RtError) as e:
        sc.render_device_noalloc(d, 0, 1, b.data_ptr(), s2.cuda_stream)
    assert e.value.code == -1 and "in use" in str(e.value)
    sc.render_device_noalloc(d, 16, 1, a.data_ptr(), s1.cuda_stream)      # same stream: ordered behind the first call
    torch.cuda.synchronize(dev)
    sc.render_device_noalloc(d, 0, 17, b.data_ptr(), s2.cuda_stream)      # the scratch is free again
    torch.cuda.synchronize(dev)
    assert torch.cuda.mem_get_info(dev)[0] == free0                        # hipMemGetInfo: nothing was allocated on the device
    assert np.array_equal(_bits(a), _bits(b))
