"""CPU tests of the C++ host mirror (Scenes::*, Camera, flatten) against the oracle's
independently written scene tables and camera code."""
import ctypes as C

import numpy as np
import pytest

SIZES = [(800, 800), (356, 200), (1600, 1600), (3840, 2160)]


@pytest.mark.parametrize("name", ["dof_demo", "test", "orange_plane", "red_corner", "sunrise",
                                  "cornell_2", "cornell"])
def test_builtin_scenes_equal_oracle_tables(pt, orc, name):
    for w, h in SIZES:
        a, b = pt.builtin_scene(name, w, h), orc.builtin_scene(name, w, h)
        assert a.surfaces.tobytes() == b.surfaces.tobytes()
        assert a.materials.tobytes() == b.materials.tobytes()
        assert a.bg.tobytes() == b.bg.tobytes()
        assert a.camera.tobytes() == b.camera.tobytes()
        assert a.camera["focal"][0] == h


def test_scene_catalogue_shape(pt):
    counts = {n: (pt.builtin_scene(n, 64, 64).n_triangles, pt.builtin_scene(n, 64, 64).n_spheres)
              for n in pt.SCENE_NAMES}
    assert counts == {"dof_demo": (0, 6), "test": (0, 11), "orange_plane": (1, 4),
                      "red_corner": (2, 3), "sunrise": (1, 2), "cornell_2": (5, 3), "cornell": (15, 1)}
    with pytest.raises(KeyError):
        pt.builtin_scene("no_such_scene", 64, 64)


def test_camera_matrix_equals_oracle(pt, orc):
    host = pt.load_host()
    rng = np.random.default_rng(3)
    poses = [(0.0, 0.0), (-np.pi / 4, 0.0), (0.0, 5.78)] + [tuple(p) for p in rng.uniform(-3, 3, (50, 2))]
    for rx, ry in poses:
        for off in (0, 1):
            out = np.zeros(9, dtype=np.float32)
            host.pt_host_camera_matrix(C.c_float(rx), C.c_float(ry), off, C.c_void_p(out.ctypes.data))
            want = orc.camera_init([0, 0, 0], rx, ry, 1.0, off, 8, 8)["rot"][0]
            assert out.tobytes() == want.tobytes(), (rx, ry, off)
