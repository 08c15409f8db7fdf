"""ctypes access to oracle/liboracle.so — the CPU restatement used as the parity checker (tests only)."""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_PATH = os.path.join(ROOT, "oracle", "liboracle.so")
REF_SVPNG_PATH = os.path.join(ROOT, "oracle", "_ref", "libref_svpng.so")

_o = C.CDLL(ORACLE_PATH)
_f32p = C.POINTER(C.c_float)
_o.orc_mesh_new.restype = C.c_void_p
_o.orc_mesh_free.argtypes = [C.c_void_p]
_o.orc_mesh_count.argtypes = [C.c_void_p]
_o.orc_mesh_copy.argtypes = [C.c_void_p, C.c_void_p]
_o.orc_transform_matrix.argtypes = [_f32p, _f32p, _f32p, _f32p]
_o.orc_mesh_read_obj.argtypes = [C.c_void_p, C.c_char_p, C.c_void_p, C.c_void_p, C.c_int]
_o.orc_mesh_read_obj_chap2.argtypes = [C.c_void_p, C.c_char_p, C.c_float, C.c_float]
_o.orc_bvh_build.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int]
_o.orc_hdr_load.argtypes = [C.c_char_p, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(_f32p)]
_o.orc_free.argtypes = [C.c_void_p]
_o.orc_hdr_cache.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p]
_o.orc_camera.argtypes = [C.c_float, C.c_float, C.c_float, _f32p, _f32p]
_o.orc_render.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int,
                          C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p,
                          C.c_int, C.c_void_p]
_o.orc_primary_hits.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p,
                                C.c_void_p, C.c_void_p]
_o.orc_hit_one.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(C.c_int32), C.POINTER(C.c_float)]
_o.orc_tonemap_rgb8.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int]
for _n in ("sin", "cos", "asin", "log"):
    getattr(_o, "orc_det_" + _n).argtypes = [C.c_float]
    getattr(_o, "orc_det_" + _n).restype = C.c_float
_o.orc_det_atan2.argtypes = [C.c_float, C.c_float]
_o.orc_det_atan2.restype = C.c_float
_o.orc_det_pow.argtypes = [C.c_float, C.c_float]
_o.orc_det_pow.restype = C.c_float
_o.orc_sobol.argtypes = [C.c_int, C.c_int]
_o.orc_sobol.restype = C.c_float
_o.orc_wang_hash.argtypes = [C.POINTER(C.c_uint32)]
_o.orc_wang_hash.restype = C.c_uint32
if hasattr(_o, "orc_chap1_render"):
    _o.orc_chap1_render.argtypes = [C.c_int, C.c_int, C.c_int, C.c_uint32, C.c_void_p]

raw = _o


def _fa(x, n):
    return (C.c_float * n)(*[float(v) for v in np.asarray(x).reshape(n)])


def transform_matrix(rot, tr, sc):
    out = (C.c_float * 16)()
    _o.orc_transform_matrix(_fa(rot, 3), _fa(tr, 3), _fa(sc, 3), out)
    return np.array(out[:], np.float32)


class Mesh:
    def __init__(self):
        self.h = C.c_void_p(_o.orc_mesh_new())

    def __del__(self):
        if self.h:
            _o.orc_mesh_free(self.h)
            self.h = None

    def read_obj(self, path, mat18, trans16, smooth):
        m = np.ascontiguousarray(mat18, np.float32)
        t = np.ascontiguousarray(trans16, np.float32)
        assert _o.orc_mesh_read_obj(self.h, path.encode(), m.ctypes.data, t.ctypes.data, int(smooth)) == 0

    def read_obj_chap2(self, path, scale, yoff):
        assert _o.orc_mesh_read_obj_chap2(self.h, path.encode(), scale, yoff) == 0

    def triangles(self):
        n = _o.orc_mesh_count(self.h)
        out = np.zeros((n, 36), np.float32)
        if n:
            _o.orc_mesh_copy(self.h, out.ctypes.data)
        return out

    def build(self, leaf_n=8, sah=True):
        n = _o.orc_mesh_count(self.h)
        cap = 2 * n + 8
        out = np.zeros((cap, 12), np.float32)
        cnt = _o.orc_bvh_build(self.h, leaf_n, int(sah), out.ctypes.data, cap)
        assert cnt > 0
        return out[:cnt].copy()


def hdr_load(path):
    w, h, p = C.c_int(), C.c_int(), _f32p()
    assert _o.orc_hdr_load(path.encode(), C.byref(w), C.byref(h), C.byref(p)) == 0
    a = np.ctypeslib.as_array(p, shape=(h.value, w.value, 3)).copy()
    _o.orc_free(p)
    return a


def hdr_cache(hdr):
    hdr = np.ascontiguousarray(hdr, np.float32)
    out = np.empty_like(hdr)
    _o.orc_hdr_cache(hdr.ctypes.data, hdr.shape[1], hdr.shape[0], out.ctypes.data)
    return out


def camera(rot, up, r):
    eye = (C.c_float * 3)()
    cam = (C.c_float * 16)()
    _o.orc_camera(rot, up, r, eye, cam)
    return np.array(eye[:], np.float32), np.array(cam[:], np.float32)


def render(tris, nodes, hdr, cache, width, height, mode, eye, cam, frame_begin, frame_count, accum=None, max_bounce=0,
           accumulate=0, use_libm=False):
    tris = np.ascontiguousarray(tris, np.float32).reshape(-1, 36)
    nodes = np.ascontiguousarray(nodes, np.float32).reshape(-1, 12)
    if accum is None:
        accum = np.zeros((height, width, 3), np.float32)
    eye = np.ascontiguousarray(eye, np.float32)
    cam = np.ascontiguousarray(cam, np.float32)
    hw = hh = 0
    hp = cp = None
    if hdr is not None:
        hdr = np.ascontiguousarray(hdr, np.float32)
        hh, hw = hdr.shape[:2]
        hp = hdr.ctypes.data
        if cache is not None:
            cache = np.ascontiguousarray(cache, np.float32)
            cp = cache.ctypes.data
    stats = np.zeros(6, np.uint64)
    rc = _o.orc_render(tris.ctypes.data if tris.size else None, tris.shape[0], nodes.ctypes.data if nodes.size else None,
                       nodes.shape[0], hp, cp, hw, hh, width, height, mode, max_bounce, accumulate, eye.ctypes.data,
                       cam.ctypes.data, frame_begin, frame_count, accum.ctypes.data, int(use_libm), stats.ctypes.data)
    assert rc == 0
    names = ("paths", "rays", "internal_nodes", "leaf_nodes", "triangles_tested", "hit_updates")
    return accum, {k: int(v) for k, v in zip(names, stats)}


def primary_hits(tris, nodes, width, height, eye, cam):
    tris = np.ascontiguousarray(tris, np.float32).reshape(-1, 36)
    nodes = np.ascontiguousarray(nodes, np.float32).reshape(-1, 12)
    eye = np.ascontiguousarray(eye, np.float32)
    cam = np.ascontiguousarray(cam, np.float32)
    tri = np.empty((height, width), np.int32)
    dist = np.empty((height, width), np.float32)
    _o.orc_primary_hits(tris.ctypes.data if tris.size else None, tris.shape[0], nodes.ctypes.data if nodes.size else None,
                        nodes.shape[0], width, height, eye.ctypes.data, cam.ctypes.data, tri.ctypes.data, dist.ctypes.data)
    return tri, dist


def hit_one(tris, nodes, origin, direction):
    tris = np.ascontiguousarray(tris, np.float32)
    nodes = np.ascontiguousarray(nodes, np.float32)
    o = np.ascontiguousarray(origin, np.float32)
    d = np.ascontiguousarray(direction, np.float32)
    t, dist = C.c_int32(), C.c_float()
    _o.orc_hit_one(tris.ctypes.data, nodes.ctypes.data, o.ctypes.data, d.ctypes.data, C.byref(t), C.byref(dist))
    return t.value, dist.value


def tonemap_rgb8(rgb, flip_y=True):
    rgb = np.ascontiguousarray(rgb, np.float32)
    h, w = rgb.shape[:2]
    out = np.empty((h, w, 3), np.uint8)
    _o.orc_tonemap_rgb8(rgb.ctypes.data, w, h, out.ctypes.data, int(flip_y))
    return out


def hitmap_hash(tri):
    """h=0xcbf29ce484222325; h=(h^uint32(idx))*0x100000001b3 per pixel, rows bottom-up, miss = -1."""
    h = 0xcbf29ce484222325
    for v in tri.astype(np.int32).reshape(-1).view(np.uint32).tolist():
        h = ((h ^ v) * 0x100000001b3) & 0xFFFFFFFFFFFFFFFF
    return h
