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


def test_camera_walk_equals_oracle(pt, orc):
    """Camera::rotate / move / update (src/camera.hpp:41-58, 76-89) in the C++ mirror against the
    oracle's restatement, driven as the main loop drives them, bit for bit: random walks, steps
    that change nothing, pitch past the clamp, yaw below 0 and beyond 2 pi."""
    host = pt.load_host()
    rng = np.random.default_rng(8)

    def walk(pos, rot, moves):
        pos = np.float32(pos)
        rot = np.float32(rot)
        moves = np.ascontiguousarray(moves, dtype=np.float32).reshape(-1, 5)
        out = np.empty((len(moves), 15), dtype=np.float32)
        host.pt_host_camera_walk(C.c_void_p(pos.ctypes.data), C.c_void_p(rot.ctypes.data), C.c_int(len(moves)),
                                 C.c_void_p(moves.ctypes.data), C.c_void_p(out.ctypes.data))
        return out, orc.camera_walk(pos, rot, moves)

    for trial in range(20):
        moves = rng.normal(0, 0.3, (200, 5)).astype(np.float32)
        moves[rng.random(200) < 0.3, :2] = 0          # frames without rotation
        moves[rng.random(200) < 0.3, 2:] = 0          # frames without motion
        got, want = walk(rng.uniform(-3, 3, 3), rng.uniform(-1, 1, 2), moves)
        assert got.tobytes() == want.tobytes(), trial
        assert (got[:, 5] == (moves != 0).any(axis=1)).all()      # update() reports exactly the frames that changed something
    # the clamp and the wrap act on the stored pose after the matrix was built from the raw one
    got, want = walk([0, 0, 0], [1.5, 0.1], [[0.5, -0.3, 0, 0, 0], [0, 0, 0, 0, 0], [0.0, 7.0, 1, 0, 1]])
    assert got.tobytes() == want.tobytes()
    assert got[0, 3] == np.float32(np.pi / 2) and got[0, 4] == np.float32(float(np.float32(0.1) + np.float32(-0.3)) + 2 * np.pi)
    raw = orc.camera_init([0, 0, 0], np.float32(2.0), np.float32(0.1) + np.float32(-0.3), 1.0, 0, 8, 8)["rot"][0]
    assert got[0, 6:].tobytes() == raw.tobytes()                 # pitch 2.0 in the matrix, pi/2 in the pose
    assert got[1, 5] == 0 and got[1, :5].tobytes() == got[0, :5].tobytes()
    assert 0 <= got[2, 4] <= np.float32(2 * np.pi)


def test_scene_files_round_trip(pt, tmp_path):
    """save_scene -> load_scene reproduces every flattened scene bit for bit."""
    for name in list(pt.SCENE_NAMES) + ["glass_stress", "heightfield:12"]:
        path = tmp_path / (name.replace(":", "_") + ".txt")
        pt.save_scene_file(name, 320, 200, path)
        a, b = pt.builtin_scene(name, 320, 200), pt.load_scene_file(path, 320, 200)
        assert a.surfaces.tobytes() == b.surfaces.tobytes(), name
        assert a.materials.tobytes() == b.materials.tobytes(), name
        assert a.bg.tobytes() == b.bg.tobytes() and a.camera.tobytes() == b.camera.tobytes(), name


def test_scene_file_by_hand(pt, orc, tmp_path):
    path = tmp_path / "hand.txt"
    path.write_text("""# two surfaces
camera 0 0 -3  0 0  auto
background 0.1 0.2 0.3
dof 0.25 4
material red diffuse 1 0.05 0.05
material lamp diffuse 1 1 1  2 2 2     # emissive
material glass glass 1 1 1
sphere 0 0 0 0.5 glass
triangle -2 1 -1  0 1 10  2 1 -1 red
sphere 0 -3 0 1 lamp
""")
    sc = pt.load_scene_file(path, 64, 48)
    assert len(sc.surfaces) == 3 and list(sc.surfaces["kind"]) == [pt.SPHERE, pt.TRIANGLE, pt.SPHERE]
    assert list(sc.materials["type"]) == [pt.GLASS, pt.DIFFUSE, pt.DIFFUSE]
    assert np.allclose(sc.materials["emittance"][2], 2) and np.allclose(sc.bg, [0.1, 0.2, 0.3])
    cam = sc.camera[0]
    assert cam["focal"] == 48 and cam["dof_on"] == 1 and cam["dof_aperture"] == np.float32(0.25)
    # the plain camera statement adds OFFSET, as the reference's constructor does
    want = orc.camera_init([0, 0, -3], 0, 0, 48, 1, 64, 48)["rot"][0]
    assert cam["rot"].tobytes() == want.tobytes()


@pytest.mark.parametrize("text,fragment", [
    ("sphere 0 0 0 1 nomat\ncamera 0 0 0 0 0 auto\n", "unknown material"),
    ("camera 0 0 0 0 0 auto\nmaterial m plastic 1 1 1\n", "unknown material type"),
    ("camera 0 0 0 0 0 auto\ntriangle 0 0 0 1 1 1 m\n", "too few"),
    ("background 0 0 0\n", "no camera"),
    ("camera 0 0 0 0 0 auto\nteapot 1 2 3\n", "unknown statement"),
])
def test_scene_file_errors(pt, tmp_path, text, fragment):
    path = tmp_path / "bad.txt"
    path.write_text(text)
    with pytest.raises(ValueError) as e:
        pt.load_scene_file(path, 32, 32)
    assert fragment in str(e.value) and "line" in str(e.value)
    with pytest.raises(ValueError):
        pt.load_scene_file(tmp_path / "missing.txt", 32, 32)


CUBE_OBJ = """# a unit cube: quads, a pentagon-free mix of reference styles
v 0 0 0
v 1 0 0
v 1 1 0
v 0 1 0
v 0 0 1
v 1 0 1
v 1 1 1
v 0 1 1
vt 0 0
vn 0 0 1
f 1 2 3 4
f 5/1 6/1 7/1 8/1
f 1//1 2//1 6//1 5//1
f 2/1/1 3/1/1 7/1/1 6/1/1
f -6 -5 -1 -2          # 3 4 8 7 counted back from the last vertex
o ignored
usemtl ignored
f 4 1 5
f 4 5 8
"""


def test_scene_file_with_obj_mesh(pt, tmp_path):
    """The `mesh` statement: faces of an OBJ file as triangles, quads as fans, scaled and moved,
    in file order between the other surfaces."""
    (tmp_path / "cube.obj").write_text(CUBE_OBJ)
    path = tmp_path / "scene.txt"
    path.write_text("""camera 0 0 -3  0 0  auto
material grey diffuse 0.5 0.5 0.5
material lamp diffuse 1 1 1  3 3 3
sphere 0 -3 0 1 lamp
mesh cube.obj grey 2 -1 -1 4
sphere 0 0 0 0.25 grey
""")
    sc = pt.load_scene_file(path, 64, 48)
    kinds = list(sc.surfaces["kind"])
    assert kinds == [pt.SPHERE] + [pt.TRIANGLE] * 12 + [pt.SPHERE]
    tri = sc.surfaces[1:13]
    assert (tri["material"] == tri["material"][0]).all() or len(set(tri["material"])) == 12
    # first quad 1 2 3 4 -> (1,2,3) and (1,3,4), vertex * 2 + (-1,-1,4)
    assert np.array_equal(tri["a"][0], [-1, -1, 4]) and np.array_equal(tri["b"][0], [1, -1, 4])
    assert np.array_equal(tri["c"][0], [1, 1, 4])
    assert np.array_equal(tri["a"][1], [-1, -1, 4]) and np.array_equal(tri["b"][1], [1, 1, 4])
    assert np.array_equal(tri["c"][1], [-1, 1, 4])
    # the face written with negative references is 3 4 8 7
    assert np.array_equal(tri["a"][8], [1, 1, 4]) and np.array_equal(tri["b"][8], [-1, 1, 4])
    assert np.array_equal(tri["c"][8], [-1, 1, 6])
    corners = np.concatenate([tri["a"], tri["b"], tri["c"]])
    assert corners.min(0).tolist() == [-1, -1, 4] and corners.max(0).tolist() == [1, 1, 6]


@pytest.mark.parametrize("obj,fragment", [
    ("v 0 0 0\nv 1 0 0\nf 1 2 3\n", "out of range"),
    ("v 0 0 0\nv 1 0 0\nv 0 1 0\nf 1 2\n", "three vertices"),
    ("v 0 0\n", "three coordinates"),
    ("v 0 0 0\nv 1 0 0\nv 0 1 0\nf 1 x 3\n", "bad vertex reference"),
    (None, "cannot open mesh"),
])
def test_obj_mesh_errors(pt, tmp_path, obj, fragment):
    if obj is not None:
        (tmp_path / "m.obj").write_text(obj)
    path = tmp_path / "scene.txt"
    path.write_text("camera 0 0 0 0 0 auto\nmaterial g diffuse 1 1 1\nmesh m.obj g\n")
    with pytest.raises(ValueError) as e:
        pt.load_scene_file(path, 32, 32)
    assert fragment in str(e.value) and "scene file line 3" in str(e.value)
