"""CPU tests of the oracle itself: the known answers recorded from the reference, the committed
golden vectors, and properties of the restated algorithm."""
import json
import os

import numpy as np
import pytest

from conftest import GOLDEN

SIZE = 800


def test_reference_known_answers(orc):
    """14 primary-hit answers recorded from the reference's own intersection code."""
    with open(os.path.join(GOLDEN, "reference_known_answers.json")) as f:
        kat = json.load(f)
    checked = 0
    for name, rows in kat.items():
        if name == "note":
            continue
        idx, dist = orc.primary_hits(orc.builtin_scene(name, SIZE, SIZE))
        for u, v, want_idx, want_dist in rows:
            assert idx[v, u] == want_idx, (name, u, v)
            assert dist[v, u] == np.float32(want_dist), (name, u, v, float(dist[v, u]))
            checked += 1
    assert checked == 14


@pytest.mark.parametrize("name", ["dof_demo", "test", "orange_plane", "red_corner", "sunrise",
                                  "cornell_2", "cornell"])
def test_primary_hits_match_golden(orc, golden, name):
    g = golden("primary_hits_800.npz")
    idx, dist = orc.primary_hits(orc.builtin_scene(name, SIZE, SIZE))
    assert np.array_equal(idx, g[name + "_idx"].astype(np.int32))
    assert np.array_equal(dist[::8, ::8], g[name + "_dist"])


def test_bsdf_matches_golden(orc, golden):
    g = golden("bsdf_table.npz")
    d, f = orc.eval_bsdf(g["type"], g["color"], g["normal"], g["incident"], g["xi"])
    assert np.array_equal(d, g["out_dir"], equal_nan=True)
    assert np.array_equal(f, g["out_brdf"], equal_nan=True)


def test_bsdf_properties(orc, golden):
    g = golden("bsdf_table.npz")
    t, n, i, d, f = g["type"], g["normal"], g["incident"], g["out_dir"], g["out_brdf"]
    dif, mir, gla = t == orc.DIFFUSE, t == orc.MIRROR, t == orc.GLASS
    # diffuse: unit direction in the normal's hemisphere; weight 2*color*cos >= 0
    assert np.allclose(np.linalg.norm(d[dif], axis=1), 1, atol=1e-5)
    assert ((d[dif] * n[dif]).sum(1) >= 0).all()
    cos = (d[dif] * n[dif]).sum(1, keepdims=True)
    assert np.allclose(f[dif], 2 * g["color"][dif] * cos, rtol=1e-5, atol=1e-6)
    # mirror: reflection keeps length and flips the normal component
    assert np.allclose((d[mir] * n[mir]).sum(1), -(i[mir] * n[mir]).sum(1), atol=1e-5)
    assert np.array_equal(f[mir], g["color"][mir])
    # glass: forced xi = 2^-24 reflects, xi = 1-2^-24 refracts (when refraction exists)
    refl = d[:1000]
    assert np.allclose(np.abs((refl * n[:1000]).sum(1)), np.abs((i[:1000] * n[:1000]).sum(1)), atol=1e-5)
    assert np.isfinite(d[gla]).all()
    # total internal reflection rows (grazing, from inside): always a reflection
    k = slice(2000, 2500)
    assert np.allclose((d[k] * n[k]).sum(1), -(i[k] * n[k]).sum(1), atol=1e-5)


def test_camera_rays_match_golden(orc, golden):
    g = golden("camera_rays.npz")
    for name in ("orange_plane", "cornell_2", "red_corner", "sunrise", "dof_demo"):
        o, d = orc.camera_rays(g[name + "_camera"], g[name + "_px"], g[name + "_xi"])
        assert np.array_equal(o, g[name + "_origin"])
        assert np.array_equal(d, g[name + "_dir"])
        assert np.allclose(np.linalg.norm(d, axis=1), 1, atol=1e-6)


def test_rng_table(orc):
    with open(os.path.join(GOLDEN, "rng_table.json")) as f:
        rows = json.load(f)
    for seed, pixel, sample, draw, want in rows:
        got = orc.rng_uniform(seed, pixel, sample, draw)
        assert got == float.fromhex(want)
        assert 0.0 < got < 1.0


def test_rng_uniformity(orc):
    """Mean and variance of 200k variates, and independence of neighbouring streams."""
    n = 200000
    a = np.array([orc.rng_uniform(1, i % 977, i // 977, i % 7) for i in range(n)])
    assert abs(a.mean() - 0.5) < 0.005
    assert abs(a.var() - 1 / 12) < 0.002
    b = np.array([orc.rng_uniform(1, p, 0, 0) for p in range(20000)])
    c = np.array([orc.rng_uniform(1, p, 1, 0) for p in range(20000)])
    assert abs(np.corrcoef(b, c)[0, 1]) < 0.03
    assert abs(np.corrcoef(b[:-1], b[1:])[0, 1]) < 0.03


def test_camera_matrix_is_rotation(orc):
    for rx, ry in [(0, 0), (-np.pi / 4, 0), (0, 5.78), (0.3, -1.2)]:
        cam = orc.camera_init([0, 0, 0], rx, ry, 200, 1, 356, 200)
        m = cam["rot"][0].reshape(3, 3).T.astype(np.float64)   # columns -> matrix
        assert np.allclose(m @ m.T, np.eye(3), atol=1e-6)
        assert np.isclose(np.linalg.det(m), 1, atol=1e-6)
    # zero pose plus the OFFSET tilt: very nearly the identity, but not exactly
    cam = orc.camera_init([0, 0, 0], 0, 0, 200, 1, 356, 200)
    m = cam["rot"][0].reshape(3, 3).T
    assert np.allclose(m, np.eye(3), atol=1e-4) and not np.array_equal(m, np.eye(3, dtype=np.float32))


def test_tie_break_lowest_index_wins(orc):
    """Two coincident triangles: the first in scene order is reported (strict <)."""
    surf = np.zeros(2, dtype=orc.SURFACE)
    for s in surf:
        s["kind"] = orc.TRIANGLE
        s["a"], s["b"], s["c"] = (-1, -1, 2), (1, -1, 2), (0, 1, 2)
    surf["material"] = [0, 1]
    mats = np.zeros(2, dtype=orc.MATERIAL)
    cam = orc.camera_init([0, 0, 0], 0, 0, 8, 0, 8, 8)
    sc = orc.Scene(surf, mats, [0, 0, 0], cam)
    idx, dist, _ = orc.intersect_rays(sc, [[0, 0, 0]], [[0, 0, 1]])
    assert idx[0] == 0 and dist[0] == 2.0


def test_empty_scene_misses_everything(orc):
    cam = orc.camera_init([0, 0, -3], 0, 0, 16, 1, 16, 16)
    sc = orc.Scene(np.zeros(0, dtype=orc.SURFACE), np.zeros(0, dtype=orc.MATERIAL), [0.25, 0.5, 1], cam)
    idx, dist = orc.primary_hits(sc)
    assert (idx == -1).all() and (dist == 0).all()
    acc, cnt = orc.render(sc, 3)
    assert np.array_equal(acc, np.broadcast_to(np.float32([0.75, 1.5, 3.0]), acc.shape))
    assert cnt.segments == cnt.samples == 16 * 16 * 3


def test_render_matches_golden_and_threads_agree(orc, golden):
    g = golden("render_64_4096spp.npz")
    sc = orc.builtin_scene("cornell_2", 64, 64)
    a1, c1 = orc.render(sc, 8, threads=1)
    a8, c8 = orc.render(sc, 8, threads=8)
    assert np.array_equal(a1, a8) and c1.segments == c8.segments
    # 8 spp is a noisy estimate of the 4096-spp golden mean: compare image averages (loosely: the
    # red channel, whose light bounces longest, scatters most), then with 72 spp more tightly
    assert np.allclose(a1.mean((0, 1)) / 8, g["cornell_2_mean"].mean((0, 1)), rtol=0.15)
    a64, _ = orc.render(sc, 64, first_sample=8, threads=8)
    assert np.allclose((a1 + a64).mean((0, 1)) / 72, g["cornell_2_mean"].mean((0, 1)), rtol=0.05)
    assert abs(c1.segments / c1.samples - float(g["cornell_2_segments_per_sample"])) < 0.15


def test_sample_ranges_add_up(orc):
    """Disjoint sample ranges rendered separately sum to the one-range image."""
    sc = orc.builtin_scene("orange_plane", 48, 32)
    whole, _ = orc.render(sc, 12, first_sample=0, seed=5)
    parts = np.zeros_like(whole)
    for first, n in [(0, 5), (5, 4), (9, 3)]:
        orc.render(sc, n, first_sample=first, seed=5, accum=parts)
    assert np.array_equal(whole, parts)


def test_libc_rand_variant_agrees_statistically(orc):
    """The serial libc-rand() render (the reference's generator) and the
    counter-based render estimate the same image."""
    sc = orc.builtin_scene("cornell_2", 32, 32)
    a, ca = orc.render(sc, 256, threads=8)
    b, cb = orc.render_libc_rand(sc, 256, seed=3)
    ma, mb = a.mean((0, 1)) / 256, b.mean((0, 1)) / 256
    assert np.allclose(ma, mb, rtol=0.03)
    assert abs(ca.segments / ca.samples - cb.segments / cb.samples) < 0.05


def test_tonemap(orc):
    x = np.float32([[0.0, 1.0, 4.0], [0.5, 0.25, -1.0], [1e-3, 0.999, 2.0]])
    out = orc.tonemap(x * 4, 4)
    want = np.floor(np.clip(x.astype(np.float64), 0, 1) ** (1 / 2.2) * 255 + 0.5)
    assert np.abs(out.astype(int) - want.astype(int)).max() <= 1
    assert out[0, 0] == 0 and out[0, 1] == 255 and out[0, 2] == 255 and out[1, 2] == 0


def test_neighbouring_pixel_keys_do_not_share_runs_of_paths(orc):
    """Among the pixels of an 800 x 800 image thousands of pairs have pixel keys less than 1024
    apart.  Were a path's key pcg(pixel_key + sample), such a pair (A, B, d) would trace the
    same path for samples (s + d, s), hundreds of times over; with pcg(pixel_key ^ sample * G)
    they do not, and neighbouring keys do not draw each other's numbers shifted either."""
    def pcg(v):
        v = np.asarray(v, dtype=np.uint64) & 0xFFFFFFFF
        s = (v * 747796405 + 2891336453) & 0xFFFFFFFF
        w = (((s >> ((s >> 28) + 4)) ^ s) * 277803737) & 0xFFFFFFFF
        return ((w >> 22) ^ w) & 0xFFFFFFFF

    seed = 1
    smix = int(pcg(int(pcg(seed)) ^ 0))
    pixels = np.arange(800 * 800, dtype=np.uint64)
    keys = pcg(smix ^ pixels)
    order = np.argsort(keys)
    gap = np.diff(keys[order].astype(np.int64))
    close = np.flatnonzero((gap > 0) & (gap < 1024))
    assert len(close) > 1000                      # it is the norm, not an accident
    for k in close[:40]:
        a, b, d = int(order[k]), int(order[k + 1]), int(gap[k])       # key[b] = key[a] + d
        for s in (0, 5, 77):
            da = [orc.rng_uniform(seed, a, s + d, i) for i in range(8)]
            db = [orc.rng_uniform(seed, b, s, i) for i in range(8)]
            assert sum(x == y for x, y in zip(da, db)) <= 1
    # draws of one path against the draws of paths with neighbouring sample numbers, at every shift
    base = [orc.rng_uniform(seed, 1234, 10, i) for i in range(24)]
    for other in (9, 11, 12):
        seq = [orc.rng_uniform(seed, 1234, other, i) for i in range(24)]
        assert len(set(base) & set(seq)) <= 1


def test_paths_whose_keys_meet_do_not_draw_the_same_numbers(orc):
    """A path's stream is named by 64 bits, the key and the counter its draws start from.  Among
    the 5 million paths of eight samples of an 800 x 800 image some three thousand pairs have the
    same 32-bit key (the birthday bound); with the counters starting at zero such a pair would
    draw identical numbers throughout.  They start at pixel * A + sample * B instead."""
    def pcg(v):
        v = np.asarray(v, dtype=np.uint64) & 0xFFFFFFFF
        s = (v * 747796405 + 2891336453) & 0xFFFFFFFF
        w = (((s >> ((s >> 28) + 4)) ^ s) * 277803737) & 0xFFFFFFFF
        return ((w >> 22) ^ w) & 0xFFFFFFFF

    seed = 1
    smix = int(pcg(int(pcg(seed)) ^ 0))
    pixels = np.arange(800 * 800, dtype=np.uint64)
    pixel_keys = pcg(smix ^ pixels)
    samples = np.arange(8, dtype=np.uint64)
    keys = pcg(pixel_keys[:, None] ^ ((samples[None, :] * 0x9E3779B9) & 0xFFFFFFFF)).ravel()
    order = np.argsort(keys, kind="stable")
    same = np.flatnonzero(np.diff(keys[order]) == 0)
    assert len(same) > 1000
    for k in same[:60]:
        a, b = int(order[k]), int(order[k + 1])
        pa, sa, pb, sb = a // 8, a % 8, b // 8, b % 8
        assert (pa, sa) != (pb, sb)
        da = [orc.rng_uniform(seed, pa, sa, i) for i in range(16)]
        db = [orc.rng_uniform(seed, pb, sb, i) for i in range(16)]
        assert len(set(da) & set(db)) <= 1, (pa, sa, pb, sb)


def test_russian_roulette_is_unbiased(orc):
    """src/main.cpp:97-104: from depth 5 on a path survives with probability p and a survivor's
    throughput is divided by p, so the mean image must equal that of the same loop without any
    roulette (here: paths followed for 48 segments, beyond which cornell_2 holds a millionth of
    its light).  Held to 4 standard errors of the difference of the image means, estimated from
    independent halves of each render - a long-path energy bug cannot hide behind a path cap."""
    sc = orc.builtin_scene("cornell_2", 24, 24)
    spp = 2048

    def halves(seed0):
        a, ca = orc.render(sc, spp, seed=seed0, threads=8)
        b, _ = orc.render(sc, spp, seed=seed0 + 1, threads=8)
        return a.mean((0, 1)) / spp, b.mean((0, 1)) / spp, ca

    r1, r2, c_rr = halves(11)
    orc.set_rr_start(1 << 20)
    orc.set_path_cap(48)
    try:
        n1, n2, c_no = halves(21)
    finally:
        orc.set_rr_start(-1)
        orc.set_path_cap(0)
    assert c_no.segments > 1.5 * c_rr.segments          # the roulette does end paths
    with_rr, without = (r1 + r2) / 2, (n1 + n2) / 2
    err = np.sqrt(((r1 - r2) ** 2 + (n1 - n2) ** 2) / 4)     # standard error of the difference of the two means
    print("with roulette", with_rr, "without", without, "standard error", err)
    assert np.all(np.abs(with_rr - without) <= 4 * err + 1e-4 * without)
