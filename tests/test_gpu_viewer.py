"""The interactive loop (host/pt_viewer.hpp) replayed through pt_headless on the
GPU: -log / -exit semantics of the reference's log() (src/main.cpp:275-308),
the image restarting when the camera or a depth-of-field key changes it
(src/main.cpp:258-269), and the saved BMP."""
import os
import re
import struct
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "path-tracer_amd", "lib", "pt_headless")
pytestmark = pytest.mark.gpu


def headless(*args):
    p = subprocess.run([EXE] + [str(a) for a in args], capture_output=True, text=True, timeout=120)
    assert p.returncode == 0, p.stdout + p.stderr
    fields = dict(re.findall(r"(\w+)=(\S+)", p.stdout))
    return fields, p.stderr


def test_log_and_exit_stop_at_100_passes_with_a_bmp(tmp_path):
    fields, err = headless("-scene", "cornell_2", "-width", 96, "-height", 64, "-passes", 1000,
                           "-log", "-exit", "-images", tmp_path)
    assert fields["samples_per_pixel"] == "100" and fields["frames"] == "100"
    assert "will exit at 100" in err and "image saving" not in err
    assert "saved image to %s/100.bmp" % tmp_path in err
    assert [l for l in err.splitlines() if l.isdigit()] == [str(n) for n in range(10, 101, 10)]
    data = (tmp_path / "100.bmp").read_bytes()
    assert data[:2] == b"BM" and len(data) == 54 + 96 * 3 * 64
    size, _, offset, header, width, height, planes, bits = struct.unpack_from("<IIIIiiHH", data, 2)
    assert (size, offset, header, width, height, planes, bits) == (len(data), 54, 40, 96, 64, 1, 24)
    # the Cornell box is lit: the image is neither black nor saturated
    body = data[54:]
    assert 10 < sum(body) / len(body) < 245


def test_saved_image_is_the_tone_mapped_sums(tmp_path):
    """The BMP at 100 passes holds the same pixels as -out's PPM of the same run."""
    headless("-scene", "cornell_2", "-width", 50, "-height", 40, "-log", "-exit", "-images", tmp_path,
             "-out", tmp_path / "final.ppm")
    bmp = (tmp_path / "100.bmp").read_bytes()
    ppm = (tmp_path / "final.ppm").read_bytes()
    assert ppm.startswith(b"P6\n50 40\n255\n")
    rgb = ppm[len(b"P6\n50 40\n255\n"):]
    row = (50 * 3 + 3) & ~3
    for y in (0, 17, 39):
        line = bmp[54 + (39 - y) * row:54 + (39 - y) * row + 150]
        want = rgb[y * 150:(y + 1) * 150]
        assert line[2::3] == want[0::3] and line[1::3] == want[1::3] and line[0::3] == want[2::3]


def test_without_log_the_targets_do_not_advance(tmp_path):
    """-exit without -log: no file, stops at 100 all the same."""
    fields, err = headless("-scene", "red_corner", "-width", 48, "-height", 32, "-passes", 500, "-exit",
                           "-images", tmp_path)
    assert fields["samples_per_pixel"] == "100" and not list(tmp_path.iterdir()) and "saved" not in err


def test_log_without_exit_saves_at_doubling_targets(tmp_path):
    fields, err = headless("-scene", "cornell_2", "-width", 32, "-height", 24, "-passes", 450, "-log",
                           "-images", tmp_path)
    assert fields["samples_per_pixel"] == "450"
    assert sorted(p.name for p in tmp_path.iterdir()) == ["100.bmp", "200.bmp", "400.bmp"]


def test_batched_frames_cross_targets(tmp_path):
    """32 passes per frame: the first frame at or past 100 is 128, the next target after it 200 -> 224, then 400 -> 416."""
    fields, _ = headless("-scene", "cornell_2", "-width", 32, "-height", 24, "-passes", 14, "-batch", 32, "-log",
                         "-images", tmp_path)
    assert fields["samples_per_pixel"] == "448"
    assert sorted(p.name for p in tmp_path.iterdir()) == ["128.bmp", "224.bmp", "416.bmp"]


def test_moving_the_camera_restarts_the_image(tmp_path):
    script = tmp_path / "keys.txt"
    script.write_text("# frames ms keys\n30 16 -\n1 500 q     # half a unit up\n20 16 -\n")
    still, _ = headless("-scene", "cornell_2", "-width", 64, "-height", 48, "-passes", 51, "-images", tmp_path)
    moved, _ = headless("-scene", "cornell_2", "-width", 64, "-height", 48, "-keys", script)
    assert still["samples_per_pixel"] == "51" and moved["samples_per_pixel"] == "21" and moved["frames"] == "51"
    y0 = float(still["camera_pos"].split(",")[1])
    assert float(moved["camera_pos"].split(",")[1]) == y0 + 0.5
    # a different view of the same box: the mean radiance differs, and both are lit
    a = [float(v) for v in still["mean_rgb"].split(",")]
    b = [float(v) for v in moved["mean_rgb"].split(",")]
    assert min(a) > 0.01 and min(b) > 0.01 and a != b


def test_depth_of_field_keys_restart_the_image_even_when_nothing_changes(tmp_path):
    script = tmp_path / "keys.txt"
    script.write_text("10 16 -\n1 16 u\n5 16 -\n1 16 i+o\n3 16 -\n")
    fields, err = headless("-scene", "orange_plane", "-width", 64, "-height", 36, "-keys", script)   # DOF is on already
    assert fields["samples_per_pixel"] == "4" and fields["frames"] == "20"
    on, aperture, focal = fields["dof"].split(",")
    assert on == "1" and abs(float(aperture) - 0.16) < 1e-6 and abs(float(focal) - 5.1) < 1e-6
    assert "depth of field enabled" in err and "DoF focal length: 5.1" in err


def test_viewer_passes_equal_one_batched_launch(tmp_path):
    """100 frames of one pass use sample numbers 0..99: the same paths as one launch of 100."""
    one, _ = headless("-scene", "cornell_2", "-width", 40, "-height", 30, "-exit")
    batch, _ = headless("-scene", "cornell_2", "-width", 40, "-height", 30, "-passes", 1, "-batch", 100)
    a = [float(v) for v in one["mean_rgb"].split(",")]
    b = [float(v) for v in batch["mean_rgb"].split(",")]
    assert all(abs(x - y) <= 2e-5 * abs(y) for x, y in zip(a, b)), (a, b)


def test_bad_script_is_refused(tmp_path):
    script = tmp_path / "keys.txt"
    script.write_text("10 16 w+teleport\n")
    p = subprocess.run([EXE, "-keys", str(script)], capture_output=True, text=True, timeout=60)
    assert p.returncode == 2 and "line 1" in p.stderr
