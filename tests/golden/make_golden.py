"""Generates tests/golden/*.npz from the CPU oracle (oracle/liboracle.so) in this container.

The reference ships no golden vectors and cannot run here (no GLM / no GL), so these fixtures are the oracle's own
outputs, committed so that a later change to the oracle, the host layer, the compiler or libstdc++ (std::sort decides
the triangle permutation) fails loudly.  What ties them to the reference: the SURVEY §8c values asserted in
tests/test_host_oracle.py (counts, the c2:605-609 test ray, 45 494 primary hits, cache[23]).

Run: python tests/golden/make_golden.py
"""
import hashlib
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import oracle_lib as O  # noqa: E402
import ray_tracing_amd as rt  # noqa: E402
from ray_tracing_amd import scenes  # noqa: E402


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def main():
    out = {}
    hdr, cache = scenes.hdr_and_cache()
    out["hdr_sha"], out["cache_sha"] = sha(hdr), sha(cache)
    for name, fn in (("chap2", scenes.chap2_scene), ("chap4", scenes.chap4_scene), ("chap5", scenes.chap5_scene), ("small", scenes.small_scene)):
        tris, nodes = fn()
        out[f"{name}_tris_sha"], out[f"{name}_nodes_sha"] = sha(tris), sha(nodes[1:])
        out[f"{name}_counts"] = np.array([tris.shape[0], nodes.shape[0], int((nodes[1:, 3] > 0).sum())])
    # config 2: 240x135 sub-grid of the 1080p primary-hit map at eye (0,0,4) + full-map digest
    tris, nodes = scenes.chap2_scene()
    eye = np.array([0, 0, 4], np.float32)
    cam = np.eye(4, dtype=np.float32).reshape(16)
    tri, dist = O.primary_hits(tris, nodes, 1920, 1080, eye, cam)
    out["chap2_hits_1080p_sha"], out["chap2_dist_1080p_sha"] = sha(tri), sha(dist)
    out["chap2_hits_sub"] = tri[4::8, 4::8].copy()
    out["chap2_hits_count"] = np.array([(tri >= 0).sum()])
    # small renders (oracle, deterministic math): per-mode images
    st, sn = scenes.small_scene()
    e, c = rt.camera(20.0, 15.0, 3.5)
    for mode in (3, 4, 5):
        img, stats = O.render(st, sn, hdr, cache, 48, 40, mode, e, c, 0, 4)
        out[f"small_mode{mode}_img"] = img
        out[f"small_mode{mode}_stats"] = np.array([stats[k] for k in ("paths", "rays", "internal_nodes", "leaf_nodes", "triangles_tested", "hit_updates")], np.int64)
    np.savez_compressed(os.path.join(HERE, "oracle_golden.npz"), **out)
    print("wrote", os.path.join(HERE, "oracle_golden.npz"), {k: (v if isinstance(v, str) else getattr(v, "shape", v)) for k, v in out.items()})


if __name__ == "__main__":
    main()
