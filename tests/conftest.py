import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run on the GPU box with -m gpu)")


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
