"""Soak test of the closest-hit parity: many random scenes (ordinary magnitudes, a mix of
triangles and spheres, clustered so that near-ties and grazing hits occur) against the oracle.
   python tools/soak_parity.py [seeds] [rays per seed]
Prints the number of mismatching rays; exits non-zero if there is any."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "path-tracer_amd"))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import torch  # noqa: E402
torch.cuda.init()
import numpy as np  # noqa: E402
import pt_amd as pt  # noqa: E402
import pt_oracle as orc  # noqa: E402

seeds = int(sys.argv[1]) if len(sys.argv) > 1 else 40
m = int(sys.argv[2]) if len(sys.argv) > 2 else 200000
ctx = pt.Context(0)
bad_total = redone = rays = 0
t0 = time.time()
for seed in range(seeds):
    rng = np.random.default_rng(1000 + seed)
    n = int(rng.integers(4, 400))
    surf = np.zeros(n, dtype=pt.SURFACE)
    surf["kind"] = rng.random(n) < 0.3
    spread = float(rng.choice([0.5, 2.0, 8.0]))
    size = float(rng.choice([0.02, 0.3, 2.0]))
    surf["a"] = rng.uniform(-spread, spread, (n, 3))
    surf["b"] = surf["a"] + rng.uniform(-size, size, (n, 3))
    surf["c"] = surf["a"] + rng.uniform(-size, size, (n, 3))
    sph = surf["kind"] == pt.SPHERE
    surf["b"][sph, 0] = rng.uniform(0.05, 1.0, sph.sum()) * size
    dup = rng.integers(0, n, n // 8)                     # duplicates: exact ties
    surf[dup] = surf[rng.integers(0, n, n // 8)]
    mats = np.zeros(1, dtype=pt.MATERIAL)
    osc = orc.Scene(surf.view(orc.SURFACE), mats.view(orc.MATERIAL), [0, 0, 0],
                    orc.camera_init([0, 0, 0], 0, 0, 8, 0, 8, 8))
    o = rng.uniform(-1.5 * spread, 1.5 * spread, (m, 3)).astype(np.float32)
    d = rng.normal(size=(m, 3))
    d = (d / np.linalg.norm(d, axis=1, keepdims=True)).astype(np.float32)
    # a third of the rays aim at points near surface edges and vertices (grazing the thresholds)
    k = rng.integers(0, n, m // 3)
    w = rng.choice([0.0, 1e-6, 1e-5, 0.5, 1.0 - 1e-6, 1.0], (m // 3, 2))
    target = surf["a"][k] + w[:, :1] * (surf["b"][k] - surf["a"][k]) + (1 - w[:, :1]) * w[:, 1:] * (surf["c"][k] - surf["a"][k])
    d[:m // 3] = (target - o[:m // 3]).astype(np.float32)
    # secondary rays: origins on surfaces
    i0, _, p0 = orc.intersect_rays(osc, o[m // 3:2 * m // 3], d[m // 3:2 * m // 3])
    hit = i0 >= 0
    o[m // 3:2 * m // 3][hit] = p0[hit]
    want = orc.intersect_rays(osc, o, d)
    for accel in (pt.ACCEL_NONE, pt.ACCEL_BVH, pt.ACCEL_LDS_TILES):
        ctx.set_accel(accel)
        ctx.set_scene(surf, mats, [0, 0, 0])
        got = ctx.intersect_rays(o, d)
        bad = int((got[0] != want[0]).sum()) + int((~((got[1] == want[1]) | (np.isnan(got[1]) & np.isnan(want[1])))).sum())
        if bad:
            b = np.flatnonzero(got[0] != want[0])[:3]
            print("seed %d accel %d: %d mismatches, e.g. rays %s got %s want %s" % (
                seed, accel, bad, b.tolist(), got[0][b].tolist(), want[0][b].tolist()), flush=True)
        bad_total += bad
    rays += m
    if seed % 10 == 9:
        print("%d seeds, %d rays x 3 modes, %d mismatches, %.0f s" % (seed + 1, rays, bad_total, time.time() - t0), flush=True)
print("total: %d rays x 3 modes, %d mismatches" % (rays, bad_total))
sys.exit(1 if bad_total else 0)
