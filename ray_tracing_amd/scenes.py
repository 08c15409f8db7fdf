"""The reference's hard-coded scenes (the `main()` literals of each chapter), built with the product host layer.

Assets under ``assets/`` are the reference's bundled data files (OBJ models, one Radiance HDR), copied as data.
"""
from __future__ import annotations

import os

import numpy as np

from . import ASSETS, Material, Mesh, calculateHdrCache, getTransformMatrix, loadHDR

HDR_PATH = os.path.join(ASSETS, "HDR", "peppermint_powerplant_4k.hdr")


def _m(name):
    return os.path.join(ASSETS, "models", name)


def chap2_scene():
    """path_tracing_2.cpp:589-602: bunny x5, y-0.5, + chapter-2 quad; SAH leaf 8."""
    mesh = Mesh()
    mesh.readObjChap2(_m("Stanford Bunny.obj"), 5.0, 0.5)
    mesh.readObjChap2(os.path.join(ASSETS, "models_chap2", "quad.obj"), 1.0, 0.0)
    nodes = mesh.buildBVHwithSAH(8)
    return mesh.triangles(), nodes


def chap4_scene():
    """path_tracing_4.cpp:686-722: sphere2.obj, metallic anisotropic material."""
    mesh = Mesh()
    m = Material(baseColor=(0.45, 0.75, 0.31), roughness=0.1, specular=1.0, metallic=1.0, anisotropic=0.1)
    mesh.readObj(_m("sphere2.obj"), m, getTransformMatrix((0, 0, 0), (-0.8, 0.8, 0), (1.0, 1.0, 1.0)), True)
    nodes = mesh.buildBVHwithSAH(8)
    return mesh.triangles(), nodes


def chap5_scene():
    """path_tracing_5chap.cpp:805-843: gold nanosuit + white glossy floor slab."""
    mesh = Mesh()
    m = Material(baseColor=(1.0, 0.84, 0.0), roughness=0.6, metallic=1.0, specular=0.8, clearcoat=0.2, clearcoatGloss=0.2)
    mesh.readObj(_m("nanosuit.obj"), m, getTransformMatrix((0, 0, 0), (-0.8, 0.2, 0), (3.0, 3.0, 3.0)), True)
    m.baseColor[:] = [1.0, 1.0, 1.0]
    m.roughness = 0.05
    m.clearcoat = 0.2
    m.metallic = 0.0
    m.clearcoatGloss = 1.0
    mesh.readObj(_m("quad.obj"), m, getTransformMatrix((0, 0, 0), (0, 0.2, 0), (18.83, 0.01, 18.83)), False)
    nodes = mesh.buildBVHwithSAH(8)
    return mesh.triangles(), nodes


def small_scene(seed: int = 7):
    """A small mixed-material scene for fast parity tests: sphere.obj (320 tris) over the floor slab + emitter."""
    mesh = Mesh()
    m = Material(baseColor=(0.8, 0.3, 0.2), roughness=0.35, metallic=0.5, specular=0.6, clearcoat=0.5, clearcoatGloss=0.4,
                 subsurface=0.2, sheen=0.3, sheenTint=0.5, specularTint=0.3, anisotropic=0.4)
    mesh.readObj(_m("sphere.obj"), m, getTransformMatrix((0, 30, 0), (0.1, 0.6, 0.2), (1.4, 1.4, 1.4)), True)
    m2 = Material(baseColor=(0.9, 0.9, 0.9), roughness=0.1, clearcoat=0.2, clearcoatGloss=1.0)
    mesh.readObj(_m("quad.obj"), m2, getTransformMatrix((0, 0, 0), (0, -0.3, 0), (6.0, 0.01, 6.0)), False)
    m3 = Material(baseColor=(1, 1, 1), emissive=(6.0, 5.0, 3.0))
    mesh.readObj(_m("quad.obj"), m3, getTransformMatrix((0, 0, 20), (-1.2, 1.5, -0.5), (0.6, 0.02, 0.6)), False)
    nodes = mesh.buildBVHwithSAH(8)
    return mesh.triangles(), nodes


_hdr_cache = {}


def hdr_and_cache():
    if "v" not in _hdr_cache:
        hdr = loadHDR(HDR_PATH)
        _hdr_cache["v"] = (hdr, calculateHdrCache(hdr))
    return _hdr_cache["v"]


def synthetic_scene(n_objects: int = 64, subdiv: int = 32, seed: int = 0x5EED):
    """Procedural tessellated scene (SURVEY §8d 'synthetic'): a grid of displaced UV-spheres over a floor, Disney
    materials drawn per object from a fixed table.  Triangles = n_objects * 2*subdiv*(subdiv-1) + 12."""
    rng = np.random.RandomState(seed)
    table = [(mt, r, cc) for mt in (0.0, 1.0) for r in (0.05, 0.3, 0.6) for cc in (0.0, 0.2, 1.0)]
    side = int(np.ceil(np.sqrt(n_objects)))
    tris = []
    th = np.linspace(0.0, np.pi, subdiv + 1)
    ph = np.linspace(0.0, 2 * np.pi, 2 * subdiv + 1)
    T, Pp = np.meshgrid(th, ph, indexing="ij")
    for k in range(n_objects):
        gx, gz = k % side, k // side
        c = np.array([(gx - (side - 1) / 2) * 20.0 / side, 0.9 + 0.4 * rng.rand(), (gz - (side - 1) / 2) * 20.0 / side - 4.0])
        rad = (0.35 + 0.1 * rng.rand()) * 20.0 / side
        bump = 1.0 + 0.06 * np.sin(7 * T + 3 * rng.rand()) * np.cos(5 * Pp + 3 * rng.rand())
        P = np.stack([np.sin(T) * np.cos(Pp), np.cos(T), np.sin(T) * np.sin(Pp)], -1) * (rad * bump)[..., None] + c
        Nn = P - c
        Nn /= np.maximum(np.linalg.norm(Nn, axis=-1, keepdims=True), 1e-20)
        a, b, cc_, d = P[:-1, :-1], P[1:, :-1], P[1:, 1:], P[:-1, 1:]
        na, nb, nc, ndd = Nn[:-1, :-1], Nn[1:, :-1], Nn[1:, 1:], Nn[:-1, 1:]
        mt, r, cc = table[rng.randint(len(table))]
        mat = Material(baseColor=tuple(0.2 + 0.8 * rng.rand(3)), metallic=mt, roughness=r, clearcoat=cc, clearcoatGloss=0.5,
                       specular=0.5).as_array()
        for (p1, p2, p3, n1, n2, n3) in ((a, b, cc_, na, nb, nc), (a, cc_, d, na, nc, ndd)):
            rec = np.concatenate([p1, p2, p3, n1, n2, n3], -1).reshape(-1, 18)
            # drop the degenerate pole triangles
            e1, e2 = rec[:, 3:6] - rec[:, 0:3], rec[:, 6:9] - rec[:, 0:3]
            keep = np.linalg.norm(np.cross(e1, e2), axis=1) > 1e-12
            rec = rec[keep]
            tris.append(np.concatenate([rec, np.broadcast_to(mat, (rec.shape[0], 18))], 1))
    mesh = Mesh()
    mesh.append(np.concatenate(tris, 0).astype(np.float32))
    m2 = Material(baseColor=(0.9, 0.9, 0.9), roughness=0.05, clearcoat=0.2, clearcoatGloss=1.0)
    mesh.readObj(_m("quad.obj"), m2, getTransformMatrix((0, 0, 0), (0, 0.2, 0), (18.83, 0.01, 18.83)), False)
    nodes = mesh.buildBVHwithSAH(8)
    return mesh.triangles(), nodes


def cached_build(path: str, key: int, builder):
    """``builder()`` -> (triangles, nodes) runs only when ``path`` is missing, stale (other ``key``) or corrupt;
    otherwise the arrays come from the cache file (the reference's own layouts, raw fp32).  ``key`` = hash of the inputs,
    e.g. ``rt.hash_fnv1a64(open(obj, 'rb').read(), rt.hash_fnv1a64(params_bytes))``.  The file appears atomically (written
    beside ``path``, then renamed), so a concurrent reader sees either no file or a complete one."""
    from . import RtError, scene_cache_load, scene_cache_save
    try:
        return scene_cache_load(path, key)
    except RtError:
        tris, nodes = builder()
        tmp = f"{path}.tmp.{os.getpid()}"
        try:
            scene_cache_save(tmp, key, tris, nodes)
            os.replace(tmp, path)
        except (RtError, OSError):
            # a cache that cannot be written (read-only directory) is not fatal: the arrays are valid
            try:
                os.unlink(tmp)
            except OSError:
                pass
        return tris, nodes


def shared_build(path: str, key: int, builder, rank: int, world: int, wait_s: float = 1800.0, poll_s: float = 0.2):
    """One build per node for a multi-process job (one process per GPU, bench.py --gpus N): rank 0 loads or builds-and-saves the
    scene (``cached_build``); every other rank waits for the cache file and loads it — N concurrent SAH builds of a 10 M-triangle
    scene would take N times the memory and, at 16 / N threads each, several times as long.  Runs before anything touches the
    GPU.  A rank that waits longer than ``wait_s`` (rank 0 died, unwritable directory) builds the scene itself."""
    import time

    from . import RtError, scene_cache_load
    if world <= 1 or rank == 0:
        return cached_build(path, key, builder)
    deadline = time.monotonic() + wait_s
    while time.monotonic() < deadline:
        if os.path.exists(path):
            try:
                return scene_cache_load(path, key)
            except RtError:
                pass  # a stale file of an earlier run (other key): rank 0 is about to replace it
        time.sleep(poll_s)
    return cached_build(path, key, builder)


def synthetic_scene_cached(cache_dir: str, n_objects: int = 64, subdiv: int = 32, seed: int = 0x5EED, rank: int = 0, world: int = 1):
    """synthetic_scene behind the on-disk cache: the 10 M-triangle build (8 s on 16 cores, 2.5 s of it numpy) is paid once —
    per node, by rank 0, when ``world`` > 1 (``shared_build``)."""
    from . import hash_fnv1a64
    key = hash_fnv1a64(np.array([n_objects, subdiv, seed, 8], np.int64), hash_fnv1a64(open(_m("quad.obj"), "rb").read()))
    path = os.path.join(cache_dir, f"synthetic_{n_objects}_{subdiv}_{seed:x}.rtsc")
    return shared_build(path, key, lambda: synthetic_scene(n_objects, subdiv, seed), rank, world)
