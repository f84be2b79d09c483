"""Times the closest-hit search over the million-triangle height field with the
linear scan (through the scalar cache, and staged through LDS tiles) and with the hierarchy, on a full-size launch (800x800 camera rays).
   python tools/brute_vs_bvh.py [quads]
"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "path-tracer_amd"))
import torch  # noqa: E402  (first: its HIP runtime must initialise before the library's)
torch.cuda.init()
import numpy as np  # noqa: E402
import pt_amd  # noqa: E402

quads = int(sys.argv[1]) if len(sys.argv) > 1 else 707
scene = pt_amd.builtin_scene("heightfield:%d" % quads, 800, 800)
n = len(scene.surfaces)
ctx = pt_amd.Context(0)
out = {}
for name, accel in (("hierarchy", pt_amd.ACCEL_BVH), ("scan", pt_amd.ACCEL_NONE), ("lds tiles", pt_amd.ACCEL_LDS_TILES)):
    ctx.set_accel(accel)
    t0 = time.perf_counter()
    ctx.load(scene)
    t1 = time.perf_counter()
    ctx.load(scene)
    t2 = time.perf_counter()
    print("%-10s pt_set_scene + pt_set_camera: first %.1f ms, again %.1f ms" % (name, (t1 - t0) * 1e3, (t2 - t1) * 1e3),
          flush=True)
    ctx.primary_hits()                       # warm-up
    best = 1e9
    for _ in range(3):
        t0 = time.perf_counter()
        idx, dist = ctx.primary_hits()
        best = min(best, time.perf_counter() - t0)
    best = min(best, ctx.counters().kernel_ms * 1e-3)          # the kernel alone, without the copies back
    out[name] = (best, idx, dist)
    print("%-10s %d surfaces, 640000 rays: %.3f ms  (%.3g ray-surface pairs/s)" % (name, n, best * 1e3, 640000 * n / best),
          flush=True)
same = all(np.array_equal(out["scan"][k], out[m][k]) for k in (1, 2) for m in ("hierarchy", "lds tiles"))
print("identical results:", same)
sys.exit(0 if same else 1)
