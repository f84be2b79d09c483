"""Soak test of the hierarchy under real paths: random scenes (triangles and spheres, clustered,
with duplicates; diffuse, mirror and glass), rendered through the scan of every surface and
through the hierarchy.  The two images must be equal bit for bit: the paths' rays - origins on
surfaces, directions from the samplers, unit vectors only to rounding - must find the same
surface either way, every segment of every path.
   python tools/soak_render.py [seeds] [first seed]
Prints the scenes whose images differ; exits non-zero if there is any."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "path-tracer_amd"))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np  # noqa: E402
import pt_amd as pt  # noqa: E402
import pt_oracle as orc  # noqa: E402

seeds = int(sys.argv[1]) if len(sys.argv) > 1 else 50
first = int(sys.argv[2]) if len(sys.argv) > 2 else 0
ctx = pt.Context(0)
bad = 0
segments = 0
t0 = time.time()
for seed in range(first, first + seeds):
    rng = np.random.default_rng(5000 + seed)
    n = int(rng.integers(40, 600))
    surf = np.zeros(n, dtype=pt.SURFACE)
    surf["kind"] = rng.random(n) < rng.choice([0.0, 0.3, 0.7, 1.0])
    spread = float(rng.choice([0.5, 2.0, 8.0]))
    size = float(rng.choice([0.05, 0.15, 0.4])) * spread
    surf["a"] = rng.uniform(-spread, spread, (n, 3))
    surf["b"] = surf["a"] + rng.uniform(-size, size, (n, 3))
    surf["c"] = surf["a"] + rng.uniform(-size, size, (n, 3))
    sph = surf["kind"] == pt.SPHERE
    surf["b"][sph, 0] = rng.uniform(0.2, 1.0, sph.sum()) * size
    dup = rng.integers(0, n, n // 10)
    surf[dup] = surf[rng.integers(0, n, n // 10)]
    # a floor and a back wall far larger than the rest (they stay outside the hierarchy)
    surf[0]["kind"] = pt.TRIANGLE
    surf[0]["a"], surf[0]["b"], surf[0]["c"] = (-9 * spread, 1.2 * spread, -9 * spread), (0, 1.2 * spread, 18 * spread), (9 * spread, 1.2 * spread, -9 * spread)
    surf[1]["kind"] = pt.SPHERE
    surf[1]["a"], surf[1]["b"] = (0, 0, 6 * spread), (4 * spread, 0, 0)
    mats = np.zeros(4, dtype=pt.MATERIAL)
    mats["type"] = [0, 0, 1, 2]
    mats["color"] = [(0.8, 0.8, 0.8), (0.9, 0.3, 0.2), (1, 1, 1), (1, 1, 1)]
    mats["emittance"][0] = (0.4, 0.4, 0.4)
    surf["material"] = rng.integers(0, 4, n)
    cam = orc.camera_init([0, 0, -3.0 * spread], 0, 0, 96, 1, 96, 96).view(pt.CAMERA)
    images = []
    for accel in (pt.ACCEL_NONE, pt.ACCEL_BVH):
        ctx.set_accel(accel)
        ctx.set_scene(surf, mats, [0.2, 0.2, 0.3])
        ctx.set_camera(cam)
        ctx.render(8, 0, seed, 0)
        images.append(ctx.read_accum()[0])
        c = ctx.counters()
    segments += c.segments
    if c.accel != pt.ACCEL_BVH:
        print("seed %d: no hierarchy was built (%d surfaces)" % (seed, n), flush=True)
    differ = int((images[0] != images[1]).any(axis=2).sum())
    if differ:
        bad += 1
        print("seed %d: %d of %d pixels differ (%d surfaces, %d spheres)" % (seed, differ, 96 * 96, n, int(sph.sum())), flush=True)
    if (seed - first) % 10 == 9:
        print("%d scenes, %.3g segments, %d scenes differ, %.0f s" % (seed - first + 1, segments, bad, time.time() - t0), flush=True)
ctx.set_accel(pt.ACCEL_AUTO)
print("total: %d scenes, %.3g segments through the hierarchy, %d scenes differ" % (seeds, segments, bad))
sys.exit(1 if bad else 0)
