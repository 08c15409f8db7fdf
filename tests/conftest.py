import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


if os.environ.get("RT_ABORT_TRACE"):  # debug aid: C backtrace on SIGABRT (tools/dbg/abort_trace.c, built by hand with gcc)
    import ctypes
    _shim = os.path.join(ROOT, "tools", "dbg", "abort_trace.so")
    if os.path.exists(_shim):
        ctypes.CDLL(_shim)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run on the GPU box with -m gpu)")
    config.addinivalue_line("markers", "library_default_form: leave the choice between the one- and two-kernel form of the path loop to the library")


@pytest.fixture(autouse=True)
def _pin_two_kernel_form(request):
    """The parity tests render small images, and launches of fewer than 2^22 samples run the one-kernel form of the path loop by
    default (rt_tuning.single_kernel = 0).  Pin the two-kernel form — what the full-size configurations run — for every Scene a
    test creates, unless the test is about the library's own choice; the one-kernel form is covered by the tests that set
    single_kernel = 1, by every no-allocation test and by tools/soak.py."""
    import ray_tracing_amd as rt
    rt.Scene.default_tuning = {} if request.node.get_closest_marker("library_default_form") else {"single_kernel": 2}
    yield
    rt.Scene.default_tuning = {}


@pytest.fixture(scope="session")
def hdr_pair():
    from ray_tracing_amd import scenes
    return scenes.hdr_and_cache()


@pytest.fixture(scope="session")
def scene_small():
    from ray_tracing_amd import scenes
    return scenes.small_scene()


@pytest.fixture(scope="session")
def scene_chap2():
    from ray_tracing_amd import scenes
    return scenes.chap2_scene()


@pytest.fixture(scope="session")
def scene_chap4():
    from ray_tracing_amd import scenes
    return scenes.chap4_scene()


@pytest.fixture(scope="session")
def scene_chap5():
    from ray_tracing_amd import scenes
    return scenes.chap5_scene()


@pytest.fixture(scope="session")
def scene_synth10m():
    """BASELINE configs[4]: the ~10 M-triangle tessellated scene (256 objects x subdivision 100 + the floor slab =
    10 240 012 triangles, ~3.6 M nodes, depth 27), built once per session with the product's bit-exact SAH builder."""
    from ray_tracing_amd import scenes
    return scenes.synthetic_scene(256, 100)
