"""ctypes binding of the CPU oracle (oracle/pt_oracle.c).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg, never by the product package.  PARITY: pinned at
image level by the five renders the reference ships (tests/test_reference_images.py);
unpinned at bit level (GLM's operation order): the reference holds no golden vectors for
this path and cannot be built here, so the oracle is a restatement from the reference's
source text (file:line cited in pt_oracle.c).
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libpt_oracle.so")

SURFACE = np.dtype([("kind", "<i4"), ("material", "<i4"), ("a", "<f4", 3),
                    ("b", "<f4", 3), ("c", "<f4", 3)])
MATERIAL = np.dtype([("type", "<i4"), ("color", "<f4", 3), ("emittance", "<f4", 3)])
CAMERA = np.dtype([("pos", "<f4", 3), ("rot", "<f4", 9), ("focal", "<f4"),
                   ("width", "<i4"), ("height", "<i4"), ("dof_on", "<i4"),
                   ("dof_focal_length", "<f4"), ("dof_aperture", "<f4")])

DIFFUSE, MIRROR, GLASS = 0, 1, 2
TRIANGLE, SPHERE = 0, 1
MAX_BUILTIN = 16
SCENE_NAMES = ("dof_demo", "test", "orange_plane", "red_corner", "sunrise",
               "cornell_2", "cornell")


class _Scene(C.Structure):
    _fields_ = [("n_surfaces", C.c_int32), ("n_materials", C.c_int32),
                ("surfaces", C.c_void_p), ("materials", C.c_void_p),
                ("bg", C.c_float * 3)]


class Counters(C.Structure):
    _fields_ = [("samples", C.c_uint64), ("segments", C.c_uint64),
                ("max_depth", C.c_uint64)]


def build(force=False):
    """Compile libpt_oracle.so with the Makefile next to this file."""
    if force or not os.path.exists(_LIB_PATH) or (
            os.path.getmtime(_LIB_PATH) < os.path.getmtime(os.path.join(_HERE, "pt_oracle.c"))):
        subprocess.check_call(["make", "-C", _HERE, "-s", "-B"])
    return _LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            build()
        _lib = C.CDLL(_LIB_PATH)
        _lib.orc_rng_uniform.restype = C.c_float
        _lib.orc_rng_uniform.argtypes = [C.c_uint64, C.c_uint32, C.c_uint32, C.c_uint32]
        _lib.orc_builtin_scene.restype = C.c_int
    return _lib


def _p(a):
    return C.c_void_p(a.ctypes.data)


class Scene:
    """Surfaces + materials + background + camera, as flat numpy records."""

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

    def c_scene(self):
        s = _Scene()
        s.n_surfaces = len(self.surfaces)
        s.n_materials = len(self.materials)
        s.surfaces = self.surfaces.ctypes.data
        s.materials = self.materials.ctypes.data
        s.bg[:] = [float(x) for x in self.bg]
        return s


def builtin_scene(name, width, height):
    surf = np.zeros(MAX_BUILTIN, dtype=SURFACE)
    mats = np.zeros(MAX_BUILTIN, dtype=MATERIAL)
    bg = np.zeros(3, dtype=np.float32)
    cam = np.zeros(1, dtype=CAMERA)
    n = C.c_int32(0)
    rc = lib().orc_builtin_scene(name.encode(), int(width), int(height), _p(surf), _p(mats),
                                 C.byref(n), _p(bg), _p(cam))
    if rc != 0:
        raise KeyError(name)
    return Scene(surf[:n.value].copy(), mats[:n.value].copy(), bg, cam)


def camera_init(pos, rot_x, rot_y, focal, add_offset, width, height):
    cam = np.zeros(1, dtype=CAMERA)
    pos = np.ascontiguousarray(pos, dtype=np.float32)
    lib().orc_camera_init(_p(cam), _p(pos), C.c_float(rot_x), C.c_float(rot_y),
                          C.c_float(focal), int(add_offset), int(width), int(height))
    return cam


def camera_walk(pos, rot, moves):
    """camera.hpp's rotate / move / update for every row of moves (d_rot_x, d_rot_y, dx, dy, dz)
    from the stored pose: rows of pos (3), rot (2), update()'s result, matrix (9)."""
    pos = np.ascontiguousarray(pos, dtype=np.float32)
    rot = np.ascontiguousarray(rot, dtype=np.float32)
    moves = np.ascontiguousarray(moves, dtype=np.float32).reshape(-1, 5)
    out = np.empty((len(moves), 15), dtype=np.float32)
    lib().orc_camera_walk(_p(pos), _p(rot), C.c_int(len(moves)), _p(moves), _p(out))
    return out


def primary_hits(scene):
    n = scene.width * scene.height
    idx = np.empty(n, dtype=np.int32)
    dist = np.empty(n, dtype=np.float32)
    cs = scene.c_scene()
    lib().orc_primary_hits(C.byref(cs), _p(scene.camera), _p(idx), _p(dist))
    return idx.reshape(scene.height, scene.width), dist.reshape(scene.height, scene.width)


def intersect_rays(scene, o, d):
    o = np.ascontiguousarray(o, dtype=np.float32)
    d = np.ascontiguousarray(d, dtype=np.float32)
    n = o.shape[0]
    idx = np.empty(n, dtype=np.int32)
    dist = np.empty(n, dtype=np.float32)
    point = np.empty((n, 3), dtype=np.float32)
    cs = scene.c_scene()
    lib().orc_intersect_rays(C.byref(cs), _p(o), _p(d), C.c_int64(n), _p(idx), _p(dist), _p(point))
    return idx, dist, point


def camera_rays(camera, px, xi):
    camera = np.ascontiguousarray(camera, dtype=CAMERA)
    px = np.ascontiguousarray(px, dtype=np.int32)
    xi = np.ascontiguousarray(xi, dtype=np.float32)
    n = px.shape[0]
    o = np.empty((n, 3), dtype=np.float32)
    d = np.empty((n, 3), dtype=np.float32)
    lib().orc_camera_rays(_p(camera), _p(px), _p(xi), C.c_int64(n), _p(o), _p(d))
    return o, d


def eval_bsdf(mtype, color, normal, incident, xi):
    mtype = np.ascontiguousarray(mtype, dtype=np.int32)
    color = np.ascontiguousarray(color, dtype=np.float32)
    normal = np.ascontiguousarray(normal, dtype=np.float32)
    incident = np.ascontiguousarray(incident, dtype=np.float32)
    xi = np.ascontiguousarray(xi, dtype=np.float32)
    n = mtype.shape[0]
    out_dir = np.empty((n, 3), dtype=np.float32)
    out_brdf = np.empty((n, 3), dtype=np.float32)
    lib().orc_eval_bsdf(_p(mtype), _p(color), _p(normal), _p(incident), _p(xi), C.c_int64(n),
                        _p(out_dir), _p(out_brdf))
    return out_dir, out_brdf


def rng_uniform(seed, pixel, sample, draw):
    return float(lib().orc_rng_uniform(int(seed), int(pixel), int(sample), int(draw)))


def render(scene, spp, first_sample=0, seed=1, threads=1, accum=None):
    """Adds spp samples per pixel into accum (H x W x 3 float32 sums)."""
    if accum is None:
        accum = np.zeros((scene.height, scene.width, 3), dtype=np.float32)
    assert accum.dtype == np.float32 and accum.flags.c_contiguous
    cnt = Counters()
    cs = scene.c_scene()
    lib().orc_render(C.byref(cs), _p(scene.camera), C.c_uint32(spp), C.c_uint32(first_sample),
                     C.c_uint64(seed), int(threads), _p(accum), C.byref(cnt))
    return accum, cnt


def set_path_cap(segments):
    """Paths of later renders stop after `segments` segments (0 = no cap, the default)."""
    lib().orc_set_path_cap(int(segments))


def set_rr_compensation(on):
    """Russian roulette of later renders with (True: the source) or without the 1/p factor."""
    lib().orc_set_rr_compensation(1 if on else 0)


def set_rr_start(depth):
    """Later renders play russian roulette from this depth on (5: the source; negative restores it)."""
    lib().orc_set_rr_start(int(depth))


def render_libc_rand(scene, spp, seed=1, accum=None):
    if accum is None:
        accum = np.zeros((scene.height, scene.width, 3), dtype=np.float32)
    cnt = Counters()
    cs = scene.c_scene()
    lib().orc_render_libc_rand(C.byref(cs), _p(scene.camera), C.c_uint32(spp), C.c_uint(seed),
                               _p(accum), C.byref(cnt))
    return accum, cnt


def tonemap(accum, total_samples):
    accum = np.ascontiguousarray(accum, dtype=np.float32)
    n = accum.size // 3
    out = np.empty((n, 3), dtype=np.uint8)
    lib().orc_tonemap(_p(accum), C.c_int64(n), C.c_uint32(total_samples), _p(out))
    return out.reshape(accum.shape)
