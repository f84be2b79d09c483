"""Generates the golden vectors under tests/golden/ from the CPU oracle.

The reference ships no fixtures for this path and cannot be built in this
image, so these vectors pin the ORACLE (oracle/pt_oracle.c), not the reference:
they guard the oracle against regressions and give the GPU tests fixed inputs.
The 14 known answers in reference_known_answers.json were recorded from the
reference's own intersection code when the reference was studied for this build; the
oracle reproduces them bit for bit (tests/test_oracle.py).

Run from the repository root:  python tests/golden/make_golden.py
"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import pt_oracle as orc  # noqa: E402

SIZE = 800
STRIDE = 8


def primary_hits():
    out = {}
    for name in orc.SCENE_NAMES:
        sc = orc.builtin_scene(name, SIZE, SIZE)
        idx, dist = orc.primary_hits(sc)
        assert idx.min() >= -1 and idx.max() < 127
        out[name + "_idx"] = idx.astype(np.int8)
        out[name + "_dist"] = dist[::STRIDE, ::STRIDE].copy()
    np.savez_compressed(os.path.join(HERE, "primary_hits_800.npz"), **out)


def bsdf_table():
    rng = np.random.default_rng(20240611)
    n = 12000
    mtype = rng.integers(0, 3, n).astype(np.int32)
    color = rng.uniform(0.05, 1.0, (n, 3)).astype(np.float32)
    normal = rng.normal(size=(n, 3))
    normal = (normal / np.linalg.norm(normal, axis=1, keepdims=True)).astype(np.float32)
    incident = rng.normal(size=(n, 3))
    incident = (incident / np.linalg.norm(incident, axis=1, keepdims=True)).astype(np.float32)
    # odd multiples of 2^-24, the values the counter-based generator produces
    xi = ((rng.integers(0, 1 << 23, (n, 2)) * 2 + 1) * 2.0 ** -24).astype(np.float32)
    # force both glass branches: rows 0..1999 are glass with xi0 at the extremes
    mtype[:2000] = orc.GLASS
    xi[:1000, 0] = 2.0 ** -24
    xi[1000:2000, 0] = 1 - 2.0 ** -24
    # grazing incidence from inside for total internal reflection
    k = 2000
    mtype[k:k + 500] = orc.GLASS
    t = rng.normal(size=(500, 3))
    t -= (t * normal[k:k + 500]).sum(1, keepdims=True) * normal[k:k + 500]
    t /= np.linalg.norm(t, axis=1, keepdims=True)
    g = t * 0.95 + normal[k:k + 500] * np.sqrt(1 - 0.95 ** 2)
    incident[k:k + 500] = g.astype(np.float32)
    out_dir, out_brdf = orc.eval_bsdf(mtype, color, normal, incident, xi)
    np.savez_compressed(os.path.join(HERE, "bsdf_table.npz"), type=mtype, color=color,
                        normal=normal, incident=incident, xi=xi, out_dir=out_dir,
                        out_brdf=out_brdf)


def camera_table():
    rng = np.random.default_rng(7)
    out = {}
    for name in ("orange_plane", "cornell_2", "red_corner", "sunrise", "dof_demo"):
        sc = orc.builtin_scene(name, 356, 200)
        n = 3000
        px = np.stack([rng.integers(0, 356, n), rng.integers(0, 200, n)], 1).astype(np.int32)
        xi = ((rng.integers(0, 1 << 23, (n, 4)) * 2 + 1) * 2.0 ** -24).astype(np.float32)
        o, d = orc.camera_rays(sc.camera, px, xi)
        out[name + "_px"] = px
        out[name + "_xi"] = xi
        out[name + "_origin"] = o
        out[name + "_dir"] = d
        out[name + "_camera"] = sc.camera
    np.savez_compressed(os.path.join(HERE, "camera_rays.npz"), **out)


def rng_table():
    seeds = [1, 0xDEADBEEFCAFE, 2 ** 64 - 1]
    rows = []
    for seed in seeds:
        for pixel in (0, 1, 639999, 8294399):
            for sample in (0, 1, 1023, 4095):
                for draw in (0, 1, 2, 17):
                    rows.append((seed, pixel, sample, draw,
                                 float(orc.rng_uniform(seed, pixel, sample, draw)).hex()))
    with open(os.path.join(HERE, "rng_table.json"), "w") as f:
        json.dump(rows, f)


def render_stats():
    """Low-resolution converged images for the statistical gate: 64x64, 4096 spp."""
    out = {}
    for name in ("cornell_2", "orange_plane", "cornell"):
        sc = orc.builtin_scene(name, 64, 64)
        acc, cnt = orc.render(sc, 4096, first_sample=0, seed=1, threads=8)
        out[name + "_mean"] = (acc / 4096).astype(np.float32)
        out[name + "_segments_per_sample"] = np.float64(cnt.segments / cnt.samples)
        acc2, _ = orc.render(sc, 4096, first_sample=0, seed=2, threads=8)
        out[name + "_mean_seed2"] = (acc2 / 4096).astype(np.float32)
    np.savez_compressed(os.path.join(HERE, "render_64_4096spp.npz"), **out)


if __name__ == "__main__":
    primary_hits()
    bsdf_table()
    camera_table()
    rng_table()
    render_stats()
    print("golden vectors written to", HERE)
