"""-m gpu: the stream contract of INTEGRATION.md ("Streams").  Collected after the parity tests on purpose: a call from a second
caller stream switches overlapped launches off on the device for the rest of the process (rt_render_device), so everything
before this file exercises the overlapped path from the one stream the suite uses (torch's current stream)."""
import numpy as np
import pytest

import oracle_lib as O  # noqa: F401
import ray_tracing_amd as rt

pytestmark = pytest.mark.gpu


def _bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


def _assert_images_equal(gpu, ora, what):
    same = _bits(gpu) == _bits(ora)
    both_nan = np.isnan(gpu) & np.isnan(ora)
    bad = ~(same | both_nan)
    assert not bad.any(), f"{what}: {bad.sum()} of {bad.size} floats differ"


@pytest.mark.timeout(120)
def test_second_caller_stream_reuses_the_workspace_in_order(scene_chap5, hdr_pair):
    """Slot reuse is ordered by events whichever stream the previous user ran on: a 64-frame call on one stream, then 4 frames
    on another (the caller orders ITS accumulation buffer; the library orders its workspace) == one 68-frame call."""
    import torch
    tris, nodes = scene_chap5
    hdr, cache = hdr_pair
    sc = rt.Scene(tris, nodes, hdr, cache, device=0)
    eye, cam = rt.camera(0.0, 20.0, 4.0)
    W, H = 1920, 1080
    d = rt.make_desc(W, H, mode=5, eye=eye, camera_rotate=cam)
    st = torch.cuda.current_stream().cuda_stream
    a, b = (torch.zeros((H, W, 3), dtype=torch.float32, device="cuda:0") for _ in range(2))
    sc.set_tuning(launch_frames=68)
    sc.render_device(d, 0, 68, b.data_ptr(), st)
    torch.cuda.synchronize()
    sc.set_tuning(launch_frames=64)
    other = torch.cuda.Stream()
    torch.cuda.synchronize()
    sc.render_device(d, 0, 64, a.data_ptr(), st)
    ev = torch.cuda.Event()
    ev.record()
    other.wait_event(ev)
    for f in range(64, 68):
        sc.render_device(d, f, 1, a.data_ptr(), other.cuda_stream)
    torch.cuda.synchronize()
    _assert_images_equal(a.cpu().numpy(), b.cpu().numpy(), "two caller streams")


@pytest.mark.timeout(120)
def test_overlapped_calls_from_priority_and_many_caller_streams(scene_small, hdr_pair):
    """INTEGRATION.md's stream contract: overlapped launches serve one ordinary caller stream per device; a caller that brings a
    priority stream, or several streams, gets its short launches in order on its own streams (the library adds no queue of its
    own).  Eight live scenes, chains of one-frame launches from a high-priority stream and from three ordinary streams in turn
    (the pattern that stalled while such callers were still overlapped): all finish, images equal one call."""
    import torch
    tris, nodes = scene_small
    hdr, cache = hdr_pair
    live = [rt.Scene(tris, nodes, hdr, cache, device=0) for _ in range(8)]
    eye, cam = rt.camera(33.0, 12.0, 3.2)
    W, H, F = 200, 120, 24
    d = rt.make_desc(W, H, mode=5, eye=eye, camera_rotate=cam)
    want = live[0].render(d, 0, F)
    hi = torch.cuda.Stream(priority=-1)
    plain = [torch.cuda.Stream() for _ in range(3)]
    bufs = [torch.zeros((H, W, 3), dtype=torch.float32, device="cuda:0") for _ in live]
    torch.cuda.synchronize()  # (the pool streams are non-blocking: they do not wait for the zero fill on the default stream)
    for k, sc in enumerate(live):
        buf = bufs[k]
        streams = [hi] if k % 2 == 0 else plain
        prev = None
        for f in range(F):
            s_ = streams[f % len(streams)]
            if prev is not None and prev is not s_:
                ev = torch.cuda.Event()
                ev.record(prev)
                s_.wait_event(ev)  # the caller orders its own accumulation buffer between its streams
            sc.render_device(d, f, 1, buf.data_ptr(), s_.cuda_stream)
            prev = s_
    torch.cuda.synchronize()
    for k, buf in enumerate(bufs):
        _assert_images_equal(buf.cpu().numpy(), want, f"scene {k}")


