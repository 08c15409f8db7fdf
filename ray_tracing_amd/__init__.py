"""ray_tracing_amd — MI355X-native path tracer behind the reference's data layout.

Thin ctypes mirror of the C ABI in ``include/ray_tracing_amd.h`` (libray_tracing_amd.so: host scene layer in C++ +
hand-written HIP kernels for gfx950).  Function names follow the reference's host code
(path_tracing_5chap/path_tracing_5chap.cpp): ``readObj``, ``getTransformMatrix``, ``buildBVHwithSAH``,
``calculateHdrCache``, ``HDRLoader.load`` ...  There is no CPU fallback: if the shared library is missing the import
fails, and device entry points fail when no GPU is present.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("RT_AMD_LIB") or os.path.join(_HERE, "libray_tracing_amd.so")
ASSETS = os.path.join(os.path.dirname(_HERE), "assets")

if not os.path.exists(LIB_PATH):
    raise ImportError(
        f"{LIB_PATH} not found: build it with `make -C ray_tracing_amd/csrc` (or __graft_entry__.build()); "
        "ray_tracing_amd has no fallback path")



def _share_torch_hip_runtime():
    """PyTorch-ROCm wheels bundle their own HIP runtime (torch/lib/libamdhip64.so, SONAME libamdhip64.so.7) and ask for it by
    file name, so a process that loaded /opt/rocm's copy first ends up with TWO runtimes and torch then finds no GPU.  When
    torch is installed, load its copy before this library: both then share one runtime whatever the import order.  torch
    itself is not imported; without torch the system runtime is used."""
    import importlib.util
    import sys
    if "torch" in sys.modules:
        return
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        return
    if spec is None or not spec.submodule_search_locations:
        return
    cand = os.path.join(list(spec.submodule_search_locations)[0], "lib", "libamdhip64.so")
    if os.path.exists(cand):
        try:
            C.CDLL(cand, mode=C.RTLD_GLOBAL)
        except OSError:
            pass


_share_torch_hip_runtime()
_lib = C.CDLL(LIB_PATH)

RT_MODE_CHAP3, RT_MODE_CHAP4, RT_MODE_CHAP5 = 3, 4, 5
RT_ACCUM_RUNNING_MEAN, RT_ACCUM_SUM = 0, 1

_f32p = C.POINTER(C.c_float)
_i32p = C.POINTER(C.c_int32)


class Material(C.Structure):
    """struct Material, path_tracing_5chap.cpp:23-39 (same field order, same defaults)."""
    _fields_ = [("emissive", C.c_float * 3), ("baseColor", C.c_float * 3)] + [
        (n, C.c_float) for n in ("subsurface", "metallic", "specular", "specularTint", "roughness", "anisotropic",
                                 "sheen", "sheenTint", "clearcoat", "clearcoatGloss", "IOR", "transmission")]

    def __init__(self, **kw):
        super().__init__()
        _lib.rt_material_default(C.byref(self))
        for k, v in kw.items():
            if k in ("emissive", "baseColor"):
                getattr(self, k)[:] = list(v)
            else:
                setattr(self, k, v)

    def as_array(self) -> np.ndarray:
        return np.frombuffer(bytes(self), dtype=np.float32).copy()


class RenderDesc(C.Structure):
    _fields_ = [("width", C.c_int32), ("height", C.c_int32), ("mode", C.c_int32), ("max_bounce", C.c_int32),
                ("accumulate", C.c_int32), ("eye", C.c_float * 3), ("camera_rotate", C.c_float * 16)]


class TraceStats(C.Structure):
    _fields_ = [(n, C.c_uint64) for n in ("paths", "rays", "internal_nodes", "leaf_nodes", "triangles_tested",
                                          "hit_updates")]

    def as_dict(self):
        return {n: int(getattr(self, n)) for n, _ in self._fields_}


class Tuning(C.Structure):
    """rt_tuning: scheduling knobs of the path loop (they never change the image)."""
    _fields_ = [(n, C.c_int32) for n in ("launch_frames", "unit_frames", "leaf_lanes", "shade_lanes", "wait_budget", "no_cull", "overlap_frames", "no_entry_bvh", "single_kernel")]


class SceneInfo(C.Structure):
    """rt_scene_info."""
    _fields_ = [("n_triangles", C.c_int32), ("n_nodes", C.c_int32), ("bvh_depth", C.c_int32), ("stack_spill", C.c_int32),
                ("cull_enabled", C.c_int32), ("cull_eps", C.c_float), ("device", C.c_int32), ("entry_bvh", C.c_int32),
                ("cost_reference_tree", C.c_float), ("cost_entry_bvh", C.c_float)]


class KernelTiming(C.Structure):
    """rt_kernel_timing_info."""
    _fields_ = [("avg_ms", C.c_float), ("max_ms", C.c_float), ("launches", C.c_int32), ("shader_clock_mhz", C.c_float), ("first_avg_ms", C.c_float)]


class RtError(RuntimeError):
    def __init__(self, code: int, msg: str):
        super().__init__(f"[rt_status {code}] {msg}")
        self.code = code


_lib.rt_last_error.restype = C.c_char_p
_lib.rt_mesh_data.restype = _f32p
_lib.rt_mesh_data.argtypes = [C.c_void_p]
_lib.rt_mesh_count.argtypes = [C.c_void_p]
_lib.rt_mesh_destroy.argtypes = [C.c_void_p]
_lib.rt_mesh_destroy.restype = None
_lib.rt_mesh_append.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
_lib.rt_mesh_read_obj.argtypes = [C.c_void_p, C.c_char_p, C.POINTER(Material), _f32p, C.c_int]
_lib.rt_mesh_read_obj_chap2.argtypes = [C.c_void_p, C.c_char_p, C.c_float, C.c_float]
_lib.rt_bvh_build.argtypes = [C.c_void_p, C.c_int, C.c_int, C.POINTER(_f32p), C.POINTER(C.c_int)]
_lib.rt_hdr_load.argtypes = [C.c_char_p, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(_f32p)]
_lib.rt_hdr_cache.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p]
_lib.rt_hdr_cache_gpu.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p]
_lib.rt_camera.argtypes = [C.c_float, C.c_float, C.c_float, _f32p, _f32p]
_lib.rt_tonemap_rgb8.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int]
_lib.rt_write_png.argtypes = [C.c_char_p, C.c_int, C.c_int, C.c_void_p, C.c_int]
_lib.rt_free.argtypes = [C.c_void_p]
_lib.rt_free.restype = None
_lib.rt_scene_create.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_int,
                                 C.POINTER(C.c_void_p)]
_lib.rt_scene_create_ex.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_int,
                                    C.c_void_p, C.POINTER(C.c_void_p)]
_lib.rt_scene_destroy.argtypes = [C.c_void_p]
_lib.rt_scene_destroy.restype = None
_lib.rt_render.argtypes = [C.c_void_p, C.POINTER(RenderDesc), C.c_int, C.c_int, C.c_void_p]
_lib.rt_render_device.argtypes = [C.c_void_p, C.POINTER(RenderDesc), C.c_int, C.c_int, C.c_void_p, C.c_void_p]
_lib.rt_render_device_noalloc.argtypes = [C.c_void_p, C.POINTER(RenderDesc), C.c_int, C.c_int, C.c_void_p, C.c_void_p]
_lib.rt_scene_get_tuning.argtypes = [C.c_void_p, C.POINTER(Tuning)]
_lib.rt_scene_set_tuning.argtypes = [C.c_void_p, C.POINTER(Tuning)]
_lib.rt_scene_get_info.argtypes = [C.c_void_p, C.POINTER(SceneInfo)]
_lib.rt_scene_screen_coverage.argtypes = [C.c_void_p, C.POINTER(RenderDesc), C.POINTER(C.c_float), C.POINTER(C.c_int)]


class LaunchShape(C.Structure):
    """rt_launch_shape"""
    _fields_ = [("launch_frames", C.c_int32), ("unit_frames", C.c_int32), ("two_kernels", C.c_int32), ("overlapped", C.c_int32),
                ("screen_coverage", C.c_float)]


_lib.rt_scene_launch_shape.argtypes = [C.c_void_p, C.POINTER(RenderDesc), C.c_int, C.POINTER(LaunchShape)]
_lib.rt_primary_hits.argtypes = [C.c_void_p, C.POINTER(RenderDesc), C.c_void_p, C.c_void_p]
_lib.rt_primary_hits_device.argtypes = [C.c_void_p, C.POINTER(RenderDesc), C.c_void_p, C.c_void_p, C.c_void_p]
_lib.rt_render_stats.argtypes = [C.c_void_p, C.POINTER(RenderDesc), C.c_int, C.c_int, C.POINTER(TraceStats)]
_lib.rt_render_stats_executed.argtypes = [C.c_void_p, C.POINTER(RenderDesc), C.c_int, C.c_int, C.POINTER(TraceStats)]
_lib.rt_render_stats_camera.argtypes = [C.c_void_p, C.POINTER(RenderDesc), C.c_int, C.c_int, C.c_int, C.POINTER(TraceStats)]
_lib.rt_kernel_timing.argtypes = [C.c_void_p, C.POINTER(KernelTiming), C.c_int]
_lib.rt_kernel_time_ms.argtypes = [_f32p, C.POINTER(C.c_int), C.c_int]
_lib.rt_hash_fnv1a64.argtypes = [C.c_void_p, C.c_uint64, C.c_uint64]
_lib.rt_hash_fnv1a64.restype = C.c_uint64
_lib.rt_scene_cache_save.argtypes = [C.c_char_p, C.c_uint64, C.c_void_p, C.c_int, C.c_void_p, C.c_int]
_lib.rt_scene_cache_load.argtypes = [C.c_char_p, C.c_uint64, C.POINTER(_f32p), C.POINTER(C.c_int), C.POINTER(_f32p), C.POINTER(C.c_int)]


def _check(rc: int):
    if rc != 0:
        raise RtError(rc, _lib.rt_last_error().decode(errors="replace"))


def lib() -> C.CDLL:
    """The raw C ABI (for callers that want to bind it themselves)."""
    return _lib


def getTransformMatrix(rotateCtrl, translateCtrl, scaleCtrl) -> np.ndarray:
    """getTransformMatrix, c5:249-265.  Returns the column-major 4x4 as 16 floats."""
    out = (C.c_float * 16)()
    _check(_lib.rt_transform_matrix((C.c_float * 3)(*rotateCtrl), (C.c_float * 3)(*translateCtrl),
                                    (C.c_float * 3)(*scaleCtrl), out))
    return np.array(out[:], dtype=np.float32)


class Mesh:
    """std::vector<Triangle> of the reference, stored as Triangle_encoded records (c5:54-64)."""

    def __init__(self):
        self._h = C.c_void_p()
        _check(_lib.rt_mesh_create(C.byref(self._h)))

    def __del__(self):
        if getattr(self, "_h", None) and _lib is not None:
            _lib.rt_mesh_destroy(self._h)
            self._h = None

    def __len__(self):
        return _lib.rt_mesh_count(self._h)

    def readObj(self, filepath: str, material: Material, trans, smoothNormal: bool):
        """readObj, c5:267-397."""
        t = np.ascontiguousarray(trans, dtype=np.float32).reshape(16)
        _check(_lib.rt_mesh_read_obj(self._h, filepath.encode(), C.byref(material), t.ctypes.data_as(_f32p),
                                     1 if smoothNormal else 0))

    def readObjChap2(self, filepath: str, scale: float = 1.0, y_offset: float = 0.0):
        """readObj, c2:142-183 (+ the vertex edit of c2:590-594)."""
        _check(_lib.rt_mesh_read_obj_chap2(self._h, filepath.encode(), scale, y_offset))

    def append(self, tris36: np.ndarray):
        a = np.ascontiguousarray(tris36, dtype=np.float32).reshape(-1, 36)
        _check(_lib.rt_mesh_append(self._h, a.ctypes.data, a.shape[0]))

    def triangles(self) -> np.ndarray:
        """Triangle_encoded[n] as an (n, 36) float32 array (a copy)."""
        n = len(self)
        if n == 0:
            return np.zeros((0, 36), np.float32)
        return np.ctypeslib.as_array(_lib.rt_mesh_data(self._h), shape=(n, 36)).copy()

    def buildBVHwithSAH(self, n: int = 8) -> np.ndarray:
        """buildBVHwithSAH(triangles, nodes, 0, size-1, n), c5:462-593 + encode c5:869-875.  Permutes the mesh."""
        return self._build(n, 1)

    def buildBVH(self, n: int = 8) -> np.ndarray:
        """buildBVH, c5:399-460."""
        return self._build(n, 0)

    def _build(self, n, sah):
        p = _f32p()
        cnt = C.c_int()
        _check(_lib.rt_bvh_build(self._h, n, sah, C.byref(p), C.byref(cnt)))
        try:
            return np.ctypeslib.as_array(p, shape=(cnt.value, 12)).copy()
        finally:
            _lib.rt_free(p)


def loadHDR(path: str) -> np.ndarray:
    """HDRLoader::load, lib/hdrloader.cpp:29-97 -> (H, W, 3) float32, top scanline first."""
    w, h, p = C.c_int(), C.c_int(), _f32p()
    _check(_lib.rt_hdr_load(path.encode(), C.byref(w), C.byref(h), C.byref(p)))
    try:
        return np.ctypeslib.as_array(p, shape=(h.value, w.value, 3)).copy()
    finally:
        _lib.rt_free(p)


def calculateHdrCache(hdr: np.ndarray) -> np.ndarray:
    """calculateHdrCache, c5:595-698."""
    hdr = np.ascontiguousarray(hdr, dtype=np.float32)
    out = np.empty_like(hdr)
    _check(_lib.rt_hdr_cache(hdr.ctypes.data, hdr.shape[1], hdr.shape[0], out.ctypes.data))
    return out


def calculateHdrCacheGpu(hdr: np.ndarray) -> np.ndarray:
    """calculateHdrCache, c5:595-698, on the GPU (rt_hdr_cache_gpu): bit-identical to calculateHdrCache."""
    hdr = np.ascontiguousarray(hdr, dtype=np.float32)
    out = np.empty_like(hdr)
    _check(_lib.rt_hdr_cache_gpu(hdr.ctypes.data, hdr.shape[1], hdr.shape[0], out.ctypes.data))
    return out


def camera(rotatAngle: float = 0.0, upAngle: float = 0.0, r: float = 4.0):
    """eye / cameraRotate of display(), c5:716-720."""
    eye = (C.c_float * 3)()
    cam = (C.c_float * 16)()
    _check(_lib.rt_camera(rotatAngle, upAngle, r, eye, cam))
    return np.array(eye[:], np.float32), np.array(cam[:], np.float32)


def tonemap_rgb8(rgb: np.ndarray, flip_y: bool = True) -> np.ndarray:
    """pass3.fsh:14-25 + 8-bit conversion."""
    rgb = np.ascontiguousarray(rgb, dtype=np.float32)
    h, w = rgb.shape[:2]
    out = np.empty((h, w, 3), np.uint8)
    _check(_lib.rt_tonemap_rgb8(rgb.ctypes.data, w, h, out.ctypes.data, 1 if flip_y else 0))
    return out


def write_png(path: str, img: np.ndarray):
    """svpng(), path_tracing_1chap/svpng.inc:77-108."""
    img = np.ascontiguousarray(img, dtype=np.uint8)
    h, w, c = img.shape
    _check(_lib.rt_write_png(path.encode(), w, h, img.ctypes.data, 1 if c == 4 else 0))


def make_desc(width, height, mode=RT_MODE_CHAP5, eye=(0, 0, 4), camera_rotate=None, max_bounce=0,
              accumulate=RT_ACCUM_RUNNING_MEAN) -> RenderDesc:
    d = RenderDesc()
    d.width, d.height, d.mode, d.max_bounce, d.accumulate = width, height, mode, max_bounce, accumulate
    d.eye[:] = [float(x) for x in eye]
    if camera_rotate is None:
        camera_rotate = np.eye(4, dtype=np.float32).reshape(16)
    d.camera_rotate[:] = [float(x) for x in np.asarray(camera_rotate).reshape(16)]
    return d


class Scene:
    """Device-resident scene: what c5:881-909 uploads (triangles TBO, nodes TBO, hdrMap, hdrCache)."""

    def __init__(self, triangles: np.ndarray, nodes: np.ndarray, hdr: np.ndarray | None = None,
                 hdr_cache: np.ndarray | None = None, device: int | None = None, entry_bvh: int = 0):
        if device is not None:
            _check(_lib.rt_init(device))
        tris = np.ascontiguousarray(triangles, dtype=np.float32).reshape(-1, 36)
        nd = np.ascontiguousarray(nodes, dtype=np.float32).reshape(-1, 12)
        self.n_triangles, self.n_nodes = tris.shape[0], nd.shape[0]
        hw = hh = 0
        hp = cp = None
        if hdr is not None:
            hdr = np.ascontiguousarray(hdr, dtype=np.float32)
            hh, hw = hdr.shape[:2]
            hp = hdr.ctypes.data
            if hdr_cache is not None:
                hdr_cache = np.ascontiguousarray(hdr_cache, dtype=np.float32)
                assert hdr_cache.shape == hdr.shape
                cp = hdr_cache.ctypes.data
        self._h = C.c_void_p()
        opt = (C.c_int32 * 1)(entry_bvh)  # rt_scene_options
        _check(_lib.rt_scene_create_ex(tris.ctypes.data if tris.size else None, tris.shape[0],
                                       nd.ctypes.data if nd.size else None, nd.shape[0], hp, cp, hw, hh,
                                       C.cast(opt, C.c_void_p), C.byref(self._h)))
        if Scene.default_tuning:
            self.set_tuning(**Scene.default_tuning)

    def __del__(self):
        if getattr(self, "_h", None) and _lib is not None:  # (_lib is None while the interpreter shuts down)
            _lib.rt_scene_destroy(self._h)
            self._h = None

    def render(self, desc: RenderDesc, frame_begin: int, frame_count: int, accum: np.ndarray | None = None) -> np.ndarray:
        """pass1.draw() for frameCounter in [frame_begin, frame_begin+frame_count); accum is lastFrame (host)."""
        if accum is None:
            accum = np.zeros((desc.height, desc.width, 3), np.float32)
        assert accum.dtype == np.float32 and accum.flags.c_contiguous and accum.size == desc.width * desc.height * 3
        _check(_lib.rt_render(self._h, C.byref(desc), frame_begin, frame_count, accum.ctypes.data))
        return accum

    def render_device(self, desc: RenderDesc, frame_begin: int, frame_count: int, accum_ptr: int, stream: int = 0):
        """Same on device memory (accum_ptr = device address of W*H*3 fp32), asynchronous on `stream`."""
        _check(_lib.rt_render_device(self._h, C.byref(desc), frame_begin, frame_count, C.c_void_p(accum_ptr),
                                     C.c_void_p(stream)))

    def render_device_noalloc(self, desc: RenderDesc, frame_begin: int, frame_count: int, accum_ptr: int, stream: int = 0):
        """render_device for callers that cannot allocate: workspace = the library's static per-GPU scratch array."""
        _check(_lib.rt_render_device_noalloc(self._h, C.byref(desc), frame_begin, frame_count, C.c_void_p(accum_ptr),
                                             C.c_void_p(stream)))

    def info(self) -> SceneInfo:
        i = SceneInfo()
        _check(_lib.rt_scene_get_info(self._h, C.byref(i)))
        return i

    def screen_coverage(self, desc: RenderDesc):
        """(share of the image the scene's root box can cover from desc's camera, frames per work unit a render uses there)."""
        c, u = C.c_float(0.0), C.c_int(0)
        _check(_lib.rt_scene_screen_coverage(self._h, C.byref(desc), C.byref(c), C.byref(u)))
        return float(c.value), int(u.value)

    def launch_shape(self, desc: RenderDesc, frame_count: int) -> LaunchShape:
        """What a render call of frame_count frames with desc does: frames per launch / per unit, two kernels?, overlapped?"""
        ls = LaunchShape()
        _check(_lib.rt_scene_launch_shape(self._h, C.byref(desc), frame_count, C.byref(ls)))
        return ls

    def tuning(self) -> Tuning:
        t = Tuning()
        _check(_lib.rt_scene_get_tuning(self._h, C.byref(t)))
        return t

    default_tuning: dict = {}  # applied to every new Scene and by set_tuning() with no argument (the GPU tests pin the two-kernel
                               # form of the path loop with it: their images are small, and small launches run the one-kernel form)

    def set_tuning(self, **kw):
        """Set some knobs of rt_tuning (others keep their value); set_tuning() with no argument restores the defaults."""
        if not kw:
            _check(_lib.rt_scene_set_tuning(self._h, None))
            if Scene.default_tuning:
                self.set_tuning(**Scene.default_tuning)
            return
        t = self.tuning()
        for k, v in kw.items():
            setattr(t, k, int(v))
        _check(_lib.rt_scene_set_tuning(self._h, C.byref(t)))

    def kernel_timing(self, reset: bool = True) -> KernelTiming:
        """Average / longest path-loop launch (HIP events) and the shader clock it ran at, since the last reset."""
        t = KernelTiming()
        _check(_lib.rt_kernel_timing(self._h, C.byref(t), 1 if reset else 0))
        return t

    def primary_hits(self, desc: RenderDesc):
        """Chapter-2 hitBVH (c2:490-513) for the un-jittered camera rays: (tri index int32, distance fp32)."""
        tri = np.empty((desc.height, desc.width), np.int32)
        dist = np.empty((desc.height, desc.width), np.float32)
        _check(_lib.rt_primary_hits(self._h, C.byref(desc), tri.ctypes.data, dist.ctypes.data))
        return tri, dist

    def primary_hits_device(self, desc: RenderDesc, tri_ptr: int, dist_ptr: int, stream: int = 0):
        _check(_lib.rt_primary_hits_device(self._h, C.byref(desc), C.c_void_p(tri_ptr), C.c_void_p(dist_ptr),
                                           C.c_void_p(stream)))

    def render_stats_camera(self, desc: RenderDesc, frame_begin: int, frame_count: int, executed: bool = False) -> dict:
        """The same counters for the camera rays alone (the first kernel of the two-kernel form)."""
        st = TraceStats()
        _check(_lib.rt_render_stats_camera(self._h, C.byref(desc), frame_begin, frame_count, 1 if executed else 0, C.byref(st)))
        return st.as_dict()

    def render_stats(self, desc: RenderDesc, frame_begin: int, frame_count: int, executed: bool = False) -> dict:
        """Traversal counters of the reference's walk (default) or of the walk the product kernel executes."""
        st = TraceStats()
        fn = _lib.rt_render_stats_executed if executed else _lib.rt_render_stats
        _check(fn(self._h, C.byref(desc), frame_begin, frame_count, C.byref(st)))
        return st.as_dict()


_lib.rt_viewer_create.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_void_p)]
_lib.rt_viewer_destroy.argtypes = [C.c_void_p]
_lib.rt_viewer_destroy.restype = None
_lib.rt_viewer_display.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
_lib.rt_viewer_mouse_down.argtypes = [C.c_void_p, C.c_double, C.c_double]
_lib.rt_viewer_mouse_drag.argtypes = [C.c_void_p, C.c_double, C.c_double]
_lib.rt_viewer_mouse_wheel.argtypes = [C.c_void_p, C.c_int]
_lib.rt_viewer_set_orbit.argtypes = [C.c_void_p, C.c_float, C.c_float, C.c_float]
_lib.rt_viewer_state.argtypes = [C.c_void_p, C.POINTER(C.c_int), _f32p, _f32p, _f32p]
_lib.rt_viewer_last_frame_device.argtypes = [C.c_void_p]
_lib.rt_viewer_last_frame_device.restype = C.c_void_p
_lib.rt_viewer_read.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
_lib.rt_viewer_save_png.argtypes = [C.c_void_p, C.c_char_p, C.c_void_p]
_lib.rt_viewer_render_progressive.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_char_p, C.c_void_p]


class Viewer:
    """The reference's interactive loop without the window (rt_viewer): display() / mouse() / mouseWheel() of
    path_tracing_5chap.cpp:706-791 on a device-resident lastFrame."""

    def __init__(self, scene: Scene, width: int = 512, height: int = 512, mode: int = RT_MODE_CHAP5):
        self._scene = scene  # keep the scene alive
        self.width, self.height = width, height
        self._h = C.c_void_p()
        _check(_lib.rt_viewer_create(scene._h, width, height, mode, C.byref(self._h)))

    def __del__(self):
        if getattr(self, "_h", None) and _lib is not None:
            _lib.rt_viewer_destroy(self._h)
            self._h = None

    def display(self, n_frames: int = 1, stream: int = 0):
        _check(_lib.rt_viewer_display(self._h, n_frames, C.c_void_p(stream)))

    def mouseDown(self, x, y):
        _check(_lib.rt_viewer_mouse_down(self._h, x, y))

    def mouse(self, x, y):
        _check(_lib.rt_viewer_mouse_drag(self._h, x, y))

    def mouseWheel(self, direction: int):
        _check(_lib.rt_viewer_mouse_wheel(self._h, direction))

    def set_orbit(self, rotatAngle: float, upAngle: float, r: float):
        _check(_lib.rt_viewer_set_orbit(self._h, rotatAngle, upAngle, r))

    def state(self):
        fc, ra, ua, r = C.c_int(), C.c_float(), C.c_float(), C.c_float()
        _check(_lib.rt_viewer_state(self._h, C.byref(fc), C.byref(ra), C.byref(ua), C.byref(r)))
        return {"frameCounter": fc.value, "rotatAngle": ra.value, "upAngle": ua.value, "r": r.value}

    def read(self, stream: int = 0) -> np.ndarray:
        out = np.empty((self.height, self.width, 3), np.float32)
        _check(_lib.rt_viewer_read(self._h, out.ctypes.data, C.c_void_p(stream)))
        return out

    def save_png(self, path: str, stream: int = 0):
        _check(_lib.rt_viewer_save_png(self._h, os.fsencode(path), C.c_void_p(stream)))

    def render_progressive(self, total_frames: int, dump_every: int = 0, path_pattern: str | None = None, stream: int = 0):
        _check(_lib.rt_viewer_render_progressive(self._h, total_frames, dump_every,
                                                 os.fsencode(path_pattern) if path_pattern else None, C.c_void_p(stream)))


_lib.rt_render_multi.argtypes = [C.POINTER(C.c_void_p), C.c_int, C.POINTER(RenderDesc), C.c_int, C.c_int, C.c_void_p]


def render_multi(scenes, desc: RenderDesc, frame_begin: int, frames_each: int, sum_host: np.ndarray | None = None) -> np.ndarray:
    """rt_render_multi: replicas of one scene (one per GPU) render disjoint frame ranges at once; returns / updates the SUM buffer."""
    if sum_host is None:
        sum_host = np.zeros((desc.height, desc.width, 3), np.float32)
    assert sum_host.dtype == np.float32 and sum_host.flags.c_contiguous
    arr = (C.c_void_p * len(scenes))(*[s._h for s in scenes])
    _check(_lib.rt_render_multi(arr, len(scenes), C.byref(desc), frame_begin, frames_each, sum_host.ctypes.data))
    return sum_host


def kernel_time_ms(reset: bool = True):
    """(average ms, launches) of this thread's path-loop kernel launches since the last reset (HIP events)."""
    ms, n = C.c_float(), C.c_int()
    _check(_lib.rt_kernel_time_ms(C.byref(ms), C.byref(n), 1 if reset else 0))
    return ms.value, n.value


def chap1_render(width: int = 256, height: int = 256, samples: int = 64, seed: int = 1) -> np.ndarray:
    """Chapter 1 CPU Cornell box (path_tracing_1chap/path_tracing.cpp main(), c1:351-489): (H, W, 3) float64, row 0 = top."""
    img = np.zeros((height, width, 3), np.float64)
    _check(_lib.rt_chap1_render(width, height, samples, C.c_uint(seed), img.ctypes.data_as(C.c_void_p)))
    return img


def chap1_to_rgb8(image: np.ndarray) -> np.ndarray:
    """imshow(), c1:326-347."""
    image = np.ascontiguousarray(image, np.float64)
    out = np.empty(image.shape, np.uint8)
    _check(_lib.rt_chap1_to_rgb8(image.ctypes.data_as(C.c_void_p), image.shape[1], image.shape[0], out.ctypes.data_as(C.c_void_p)))
    return out


def hash_fnv1a64(data, seed: int = 0) -> int:
    """FNV-1a 64 of a bytes-like object or a C-contiguous array; pass the previous result as ``seed`` to chain."""
    b = np.ascontiguousarray(data).view(np.uint8).reshape(-1) if isinstance(data, np.ndarray) else np.frombuffer(bytes(data), np.uint8)
    return int(_lib.rt_hash_fnv1a64(b.ctypes.data_as(C.c_void_p), b.size, seed))


def scene_cache_save(path: str, key: int, triangles: np.ndarray, nodes: np.ndarray):
    """Write Triangle_encoded[] + BVHNode_encoded[] (raw fp32) with ``key`` = hash of the inputs they were built from."""
    t = np.ascontiguousarray(triangles, np.float32).reshape(-1, 36)
    n = np.ascontiguousarray(nodes, np.float32).reshape(-1, 12)
    _check(_lib.rt_scene_cache_save(os.fsencode(path), key, t.ctypes.data_as(C.c_void_p), t.shape[0],
                                    n.ctypes.data_as(C.c_void_p), n.shape[0]))


def scene_cache_load(path: str, key: int):
    """(triangles, nodes) from a cache file; raises RtError if the file is missing, stale (other key) or corrupt."""
    tp, np_ = _f32p(), _f32p()
    nt, nn = C.c_int(0), C.c_int(0)
    _check(_lib.rt_scene_cache_load(os.fsencode(path), key, C.byref(tp), C.byref(nt), C.byref(np_), C.byref(nn)))
    try:
        tris = np.ctypeslib.as_array(tp, shape=(nt.value, 36)).copy() if nt.value else np.zeros((0, 36), np.float32)
        nodes = np.ctypeslib.as_array(np_, shape=(nn.value, 12)).copy() if nn.value else np.zeros((0, 12), np.float32)
    finally:
        _lib.rt_free(tp), _lib.rt_free(np_)
    return tris, nodes
