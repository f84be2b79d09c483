"""ctypes binding of libpt_amd.so, the C ABI declared in include/pt_amd.h.

Thin plumbing only: every method is one C call.  There is no CPU fallback: if
the HIP library is missing or no gfx950 device is present, calls raise.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# PT_AMD_LIB selects another build of the same library (A/B experiments).
LIB_PATH = os.environ.get("PT_AMD_LIB") or os.path.join(_HERE, "lib", "libpt_amd.so")
HOST_LIB_PATH = os.path.join(_HERE, "lib", "libpt_host.so")

SURFACE = np.dtype([("kind", "<i4"), ("material", "<i4"), ("a", "<f4", 3),
                    ("b", "<f4", 3), ("c", "<f4", 3)])
MATERIAL = np.dtype([("type", "<i4"), ("color", "<f4", 3), ("emittance", "<f4", 3)])
CAMERA = np.dtype([("pos", "<f4", 3), ("rot", "<f4", 9), ("focal", "<f4"),
                   ("width", "<i4"), ("height", "<i4"), ("dof_on", "<i4"),
                   ("dof_focal_length", "<f4"), ("dof_aperture", "<f4")])

DIFFUSE, MIRROR, GLASS = 0, 1, 2
TRIANGLE, SPHERE = 0, 1
MODE_MEGAKERNEL, MODE_WAVEFRONT = 0x0, 0x1
MODE_COUNT_STEPS = 0x100
NUM_EXACT, NUM_FAST = 0x0, 0x10
MAX_DEPTH = 1024
ACCEL_AUTO, ACCEL_NONE, ACCEL_BVH, ACCEL_LDS_TILES = 0, 1, 2, 3
STATIC_SCRATCH_BYTES = 32 << 20      # PT_STATIC_SCRATCH_BYTES

SCENE_NAMES = ("dof_demo", "test", "orange_plane", "red_corner", "sunrise",
               "cornell_2", "cornell")

# Every symbol include/pt_amd.h declares.
ABI_SYMBOLS = (
    "pt_global_error", "pt_create", "pt_destroy", "pt_last_error", "pt_set_stream",
    "pt_set_scene", "pt_set_accel", "pt_set_max_depth", "pt_set_camera", "pt_clear", "pt_render", "pt_render_into",
    "pt_render_into_static", "pt_render_rows",
    "pt_read_accum", "pt_map_accum", "pt_write_accum", "pt_tonemap_rgb8", "pt_primary_hits", "pt_intersect_rays",
    "pt_camera_rays", "pt_eval_bsdf", "pt_eval_brdf", "pt_rng_uniform", "pt_get_counters",
    "pt_group_create", "pt_group_destroy", "pt_group_last_error", "pt_group_size", "pt_group_context",
    "pt_group_share", "pt_group_set_scene", "pt_group_set_accel", "pt_group_set_max_depth",
    "pt_group_set_camera", "pt_group_clear", "pt_group_render", "pt_group_reduce",
    "pt_group_read_accum", "pt_group_tonemap_rgb8",
)


class Counters(C.Structure):
    _fields_ = [("samples", C.c_uint64), ("segments", C.c_uint64), ("tests", C.c_uint64),
                ("redone", C.c_uint64),
                ("kernel_ms", C.c_double), ("total_ms", C.c_double), ("flops", C.c_double),
                ("vgprs", C.c_uint32), ("grid_blocks", C.c_uint32),
                ("accel", C.c_uint32), ("reserved", C.c_uint32),
                ("node_steps", C.c_uint64), ("leaf_steps", C.c_uint64)]


class PtError(RuntimeError):
    def __init__(self, code, message):
        super().__init__("pt_amd error %d: %s" % (code, message))
        self.code = code


_lib = None
_host = None


def load():
    """Loads libpt_amd.so; raises if it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise FileNotFoundError(
                "%s is missing: build it with `make -C path-tracer_amd/csrc` "
                "(or __graft_entry__.build()); there is no CPU fallback" % LIB_PATH)
        lib = C.CDLL(LIB_PATH)
        lib.pt_global_error.restype = C.c_char_p
        lib.pt_last_error.restype = C.c_char_p
        lib.pt_last_error.argtypes = [C.c_void_p]
        lib.pt_destroy.restype = None
        lib.pt_destroy.argtypes = [C.c_void_p]
        lib.pt_group_last_error.restype = C.c_char_p
        lib.pt_group_last_error.argtypes = [C.c_void_p]
        lib.pt_group_destroy.restype = None
        lib.pt_group_destroy.argtypes = [C.c_void_p]
        lib.pt_group_context.restype = C.c_void_p
        lib.pt_group_context.argtypes = [C.c_void_p, C.c_int]
        lib.pt_group_share.restype = None
        for name in ABI_SYMBOLS:
            fn = getattr(lib, name)
            if name not in ("pt_global_error", "pt_last_error", "pt_destroy", "pt_group_last_error",
                            "pt_group_destroy", "pt_group_context", "pt_group_share"):
                fn.restype = C.c_int
        _lib = lib
    return _lib


def load_host():
    """Loads libpt_host.so, the C++ host mirror of the reference's scene types."""
    global _host
    if _host is None:
        if not os.path.exists(HOST_LIB_PATH):
            raise FileNotFoundError("%s is missing: build it with `make -C path-tracer_amd/host`"
                                    % HOST_LIB_PATH)
        _host = C.CDLL(HOST_LIB_PATH)
        _host.pt_host_builtin_scene.restype = C.c_int
    return _host


def _p(a):
    return C.c_void_p(a.ctypes.data)


class SceneData:
    """A flattened scene: surfaces, materials, background, camera."""

    def __init__(self, surfaces, materials, bg, camera):
        self.surfaces = np.ascontiguousarray(surfaces, dtype=SURFACE)
        self.materials = np.ascontiguousarray(materials, dtype=MATERIAL)
        self.bg = np.ascontiguousarray(bg, dtype=np.float32)
        self.camera = np.ascontiguousarray(camera, dtype=CAMERA).reshape(1)

    @property
    def width(self):
        return int(self.camera["width"][0])

    @property
    def height(self):
        return int(self.camera["height"][0])

    @property
    def n_triangles(self):
        return int((self.surfaces["kind"] == TRIANGLE).sum())

    @property
    def n_spheres(self):
        return int((self.surfaces["kind"] == SPHERE).sum())


def builtin_scene(name, width, height):
    """A scene of the C++ host mirror by name, flattened, with focal length = height:
    one of the reference's Scenes::* (src/scene.hpp:32-203), "glass_stress", or
    "heightfield[:quads]"."""
    host = load_host()
    bg = np.zeros(3, dtype=np.float32)
    cam = np.zeros(1, dtype=CAMERA)
    n = C.c_int32(0)
    rc = host.pt_host_builtin_scene(name.encode(), int(width), int(height), None, None,
                                    C.c_int32(0), C.byref(n), _p(bg), _p(cam))
    if rc == -1:
        raise KeyError("unknown scene %r" % name)
    if rc == -3:
        host.pt_host_last_error.restype = C.c_char_p
        raise RuntimeError("pt_host_builtin_scene(%r): %s" % (name, host.pt_host_last_error().decode()))
    surf = np.zeros(max(n.value, 1), dtype=SURFACE)
    mats = np.zeros(max(n.value, 1), dtype=MATERIAL)
    rc = host.pt_host_builtin_scene(name.encode(), int(width), int(height), _p(surf), _p(mats),
                                    C.c_int32(len(surf)), C.byref(n), _p(bg), _p(cam))
    if rc != 0:
        host.pt_host_last_error.restype = C.c_char_p
        raise RuntimeError("pt_host_builtin_scene(%r) failed with %d: %s" % (
            name, rc, host.pt_host_last_error().decode() if rc == -3 else ""))
    return SceneData(surf[:n.value], mats[:n.value], bg, cam)


def load_scene_file(path, width, height):
    """A scene file (host/pt_scene_io.hpp) through the C++ host mirror, flattened."""
    host = load_host()
    host.pt_host_last_error.restype = C.c_char_p
    bg = np.zeros(3, dtype=np.float32)
    cam = np.zeros(1, dtype=CAMERA)
    n = C.c_int32(0)
    rc = host.pt_host_load_scene(str(path).encode(), int(width), int(height), None, None, C.c_int32(0),
                                 C.byref(n), _p(bg), _p(cam))
    if rc == -3:
        raise ValueError(host.pt_host_last_error().decode())
    surf = np.zeros(max(n.value, 1), dtype=SURFACE)
    mats = np.zeros(max(n.value, 1), dtype=MATERIAL)
    rc = host.pt_host_load_scene(str(path).encode(), int(width), int(height), _p(surf), _p(mats),
                                 C.c_int32(len(surf)), C.byref(n), _p(bg), _p(cam))
    if rc != 0:
        raise ValueError(host.pt_host_last_error().decode())
    return SceneData(surf[:n.value], mats[:n.value], bg, cam)


def save_scene_file(name, width, height, path):
    """Writes built-in or synthetic scene `name` as a scene file."""
    rc = load_host().pt_host_save_scene(name.encode(), int(width), int(height), str(path).encode())
    if rc == -1:
        raise KeyError("unknown scene %r" % name)
    if rc != 0:
        raise OSError("cannot write %s" % path)


class Context:
    """One GPU's render context (pt_ctx)."""

    def __init__(self, device=0):
        self._lib = load()
        self._h = C.c_void_p()
        rc = self._lib.pt_create(int(device), C.byref(self._h))
        if rc != 0:
            raise PtError(rc, self._lib.pt_global_error().decode())
        self.width = self.height = 0

    def close(self):
        if self._h:
            self._lib.pt_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def _check(self, rc):
        if rc != 0:
            raise PtError(rc, self._lib.pt_last_error(self._h).decode())

    def set_stream(self, stream_ptr):
        self._check(self._lib.pt_set_stream(self._h, C.c_void_p(stream_ptr)))

    def set_scene(self, surfaces, materials, bg):
        surfaces = np.ascontiguousarray(surfaces, dtype=SURFACE)
        materials = np.ascontiguousarray(materials, dtype=MATERIAL)
        bg = np.ascontiguousarray(bg, dtype=np.float32)
        self._check(self._lib.pt_set_scene(self._h, _p(surfaces), C.c_int32(len(surfaces)),
                                           _p(materials), C.c_int32(len(materials)), _p(bg)))

    def set_accel(self, accel):
        self._check(self._lib.pt_set_accel(self._h, C.c_int(accel)))

    def set_max_depth(self, max_segments):
        self._check(self._lib.pt_set_max_depth(self._h, C.c_int(max_segments)))

    def set_camera(self, camera):
        camera = np.ascontiguousarray(camera, dtype=CAMERA).reshape(1)
        self._check(self._lib.pt_set_camera(self._h, _p(camera)))
        self.width = int(camera["width"][0])
        self.height = int(camera["height"][0])

    def load(self, scene):
        """set_scene + set_camera from an object with surfaces/materials/bg/camera."""
        self.set_scene(scene.surfaces, scene.materials, scene.bg)
        self.set_camera(scene.camera)
        return self

    def clear(self):
        self._check(self._lib.pt_clear(self._h))

    def render(self, spp, first_sample=0, seed=1, mode=0):
        self._check(self._lib.pt_render(self._h, C.c_uint32(spp), C.c_uint32(first_sample),
                                        C.c_uint64(seed), C.c_int(mode)))

    def render_into(self, device_ptr, spp, first_sample=0, seed=1, mode=0):
        self._check(self._lib.pt_render_into(self._h, C.c_void_p(device_ptr), C.c_uint32(spp),
                                             C.c_uint32(first_sample), C.c_uint64(seed),
                                             C.c_int(mode)))

    def render_rows(self, device_ptr, first_row, n_rows, spp, first_sample=0, seed=1, mode=0):
        """render_into for rows first_row .. first_row + n_rows - 1 of the image only."""
        self._check(self._lib.pt_render_rows(self._h, C.c_void_p(device_ptr), C.c_int32(first_row),
                                             C.c_int32(n_rows), C.c_uint32(spp), C.c_uint32(first_sample),
                                             C.c_uint64(seed), C.c_int(mode)))

    def render_into_static(self, device_ptr, spp, first_sample=0, seed=1, mode=0):
        """render_into over the library's static device scratch area: allocates nothing."""
        self._check(self._lib.pt_render_into_static(self._h, C.c_void_p(device_ptr), C.c_uint32(spp),
                                                    C.c_uint32(first_sample), C.c_uint64(seed),
                                                    C.c_int(mode)))

    def read_accum(self):
        out = np.empty((self.height, self.width, 3), dtype=np.float32)
        total = C.c_uint64(0)
        self._check(self._lib.pt_read_accum(self._h, _p(out), C.byref(total)))
        return out, total.value

    def map_accum(self):
        """The running sums as a (height, width, 3) view of page-locked memory the context
        owns (valid until the next map_accum), and the sample count."""
        ptr = C.POINTER(C.c_float)()
        total = C.c_uint64(0)
        self._check(self._lib.pt_map_accum(self._h, C.byref(ptr), C.byref(total)))
        return np.ctypeslib.as_array(ptr, shape=(self.height, self.width, 3)), int(total.value)

    def write_accum(self, sums, total_samples):
        sums = np.ascontiguousarray(sums, dtype=np.float32)
        if sums.size != self.width * self.height * 3:
            raise ValueError("sums must hold height*width*3 floats")
        self._check(self._lib.pt_write_accum(self._h, _p(sums), C.c_uint64(total_samples)))

    def write_accum_device(self, device_ptr, total_samples):
        """The same from a device buffer of height*width*3 floats (e.g. the reduced image of several GPUs)."""
        self._check(self._lib.pt_write_accum(self._h, C.c_void_p(device_ptr), C.c_uint64(total_samples)))

    def tonemap_rgb8(self):
        out = np.empty((self.height, self.width, 3), dtype=np.uint8)
        self._check(self._lib.pt_tonemap_rgb8(self._h, _p(out)))
        return out

    def primary_hits(self, numeric=NUM_EXACT):
        idx = np.empty((self.height, self.width), dtype=np.int32)
        dist = np.empty((self.height, self.width), dtype=np.float32)
        self._check(self._lib.pt_primary_hits(self._h, _p(idx), _p(dist), C.c_int(numeric)))
        return idx, dist

    def intersect_rays(self, origin, direction, numeric=NUM_EXACT):
        origin = np.ascontiguousarray(origin, dtype=np.float32)
        direction = np.ascontiguousarray(direction, dtype=np.float32)
        n = origin.shape[0] if origin.ndim == 2 else 0
        idx = np.empty(n, dtype=np.int32)
        dist = np.empty(n, dtype=np.float32)
        point = np.empty((n, 3), dtype=np.float32)
        self._check(self._lib.pt_intersect_rays(self._h, _p(origin), _p(direction), C.c_int64(n),
                                                _p(idx), _p(dist), _p(point), C.c_int(numeric)))
        return idx, dist, point

    def camera_rays(self, px, xi, numeric=NUM_EXACT):
        px = np.ascontiguousarray(px, dtype=np.int32)
        xi = np.ascontiguousarray(xi, dtype=np.float32)
        n = px.shape[0]
        origin = np.empty((n, 3), dtype=np.float32)
        direction = np.empty((n, 3), dtype=np.float32)
        self._check(self._lib.pt_camera_rays(self._h, _p(px), _p(xi), C.c_int64(n), _p(origin),
                                             _p(direction), C.c_int(numeric)))
        return origin, direction

    def eval_bsdf(self, mtype, color, normal, incident, xi, numeric=NUM_EXACT):
        mtype = np.ascontiguousarray(mtype, dtype=np.int32)
        color = np.ascontiguousarray(color, dtype=np.float32)
        normal = np.ascontiguousarray(normal, dtype=np.float32)
        incident = np.ascontiguousarray(incident, dtype=np.float32)
        xi = np.ascontiguousarray(xi, dtype=np.float32)
        n = mtype.shape[0]
        out_dir = np.empty((n, 3), dtype=np.float32)
        out_brdf = np.empty((n, 3), dtype=np.float32)
        self._check(self._lib.pt_eval_bsdf(self._h, _p(mtype), _p(color), _p(normal), _p(incident),
                                           _p(xi), C.c_int64(n), _p(out_dir), _p(out_brdf),
                                           C.c_int(numeric)))
        return out_dir, out_brdf

    def eval_brdf(self, mtype, color, normal, reflected):
        mtype = np.ascontiguousarray(mtype, dtype=np.int32)
        color = np.ascontiguousarray(color, dtype=np.float32)
        normal = np.ascontiguousarray(normal, dtype=np.float32)
        reflected = np.ascontiguousarray(reflected, dtype=np.float32)
        n = mtype.shape[0]
        out = np.empty((n, 3), dtype=np.float32)
        self._check(self._lib.pt_eval_brdf(self._h, _p(mtype), _p(color), _p(normal), _p(reflected),
                                           C.c_int64(n), _p(out)))
        return out

    def rng_uniform(self, seed, pixel, sample, draw):
        pixel = np.ascontiguousarray(pixel, dtype=np.uint32)
        sample = np.ascontiguousarray(sample, dtype=np.uint32)
        draw = np.ascontiguousarray(draw, dtype=np.uint32)
        out = np.empty(pixel.shape[0], dtype=np.float32)
        self._check(self._lib.pt_rng_uniform(self._h, C.c_uint64(seed), _p(pixel), _p(sample),
                                             _p(draw), C.c_int64(pixel.shape[0]), _p(out)))
        return out

    def counters(self):
        c = Counters()
        self._check(self._lib.pt_get_counters(self._h, C.byref(c)))
        return c


def group_share(spp_total, n, rank, first_sample=0):
    """pt_group_share: (first, count) of rank `rank` of `n`."""
    first, count = C.c_uint32(0), C.c_uint32(0)
    load().pt_group_share(C.c_uint32(spp_total), C.c_int(n), C.c_int(rank), C.c_uint32(first_sample),
                          C.byref(first), C.byref(count))
    return first.value, count.value


class Group:
    """Several GPUs of one node rendering one image (pt_group): samples partitioned by global
    sample number, one RCCL reduce onto the first device."""

    def __init__(self, devices):
        self._lib = load()
        self._h = C.c_void_p()
        devs = (C.c_int * len(devices))(*devices)
        rc = self._lib.pt_group_create(devs, C.c_int(len(devices)), C.byref(self._h))
        if rc != 0:
            raise PtError(rc, self._lib.pt_group_last_error(None).decode())
        self.width = self.height = 0

    def close(self):
        if self._h:
            self._lib.pt_group_destroy(self._h)
            self._h = C.c_void_p()

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc):
        if rc != 0:
            raise PtError(rc, self._lib.pt_group_last_error(self._h).decode())

    @property
    def size(self):
        return int(self._lib.pt_group_size(self._h))

    def load(self, scene):
        self._check(self._lib.pt_group_set_scene(self._h, _p(scene.surfaces), C.c_int32(len(scene.surfaces)),
                                                 _p(scene.materials), C.c_int32(len(scene.materials)), _p(scene.bg)))
        self._check(self._lib.pt_group_set_camera(self._h, _p(scene.camera)))
        self.width, self.height = scene.width, scene.height
        return self

    def set_accel(self, accel):
        self._check(self._lib.pt_group_set_accel(self._h, C.c_int(accel)))

    def set_max_depth(self, max_segments):
        self._check(self._lib.pt_group_set_max_depth(self._h, C.c_int(max_segments)))

    def clear(self):
        self._check(self._lib.pt_group_clear(self._h))

    def render(self, spp_total, first_sample=0, seed=1, mode=0):
        self._check(self._lib.pt_group_render(self._h, C.c_uint32(spp_total), C.c_uint32(first_sample),
                                              C.c_uint64(seed), C.c_int(mode)))

    def reduce(self):
        self._check(self._lib.pt_group_reduce(self._h))

    def read_accum(self):
        out = np.empty((self.height, self.width, 3), dtype=np.float32)
        total = C.c_uint64(0)
        self._check(self._lib.pt_group_read_accum(self._h, _p(out), C.byref(total)))
        return out, total.value

    def tonemap_rgb8(self):
        out = np.empty((self.height, self.width, 3), dtype=np.uint8)
        self._check(self._lib.pt_group_tonemap_rgb8(self._h, _p(out)))
        return out

    def counters(self, member=0):
        c = Counters()
        h = self._lib.pt_group_context(self._h, C.c_int(member))
        rc = self._lib.pt_get_counters(C.c_void_p(h), C.byref(c))
        if rc != 0:
            raise PtError(rc, self._lib.pt_last_error(C.c_void_p(h)).decode())
        return c
