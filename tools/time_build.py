"""Times pt_set_scene (flatten, upload, hierarchy build) on the synthetic height fields."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "path-tracer_amd"))
import pt_amd  # noqa: E402

ctx = pt_amd.Context(0)
for name in ("heightfield:40", "heightfield:200", "heightfield"):
    sc = pt_amd.builtin_scene(name, 800, 800)
    for accel, label in ((pt_amd.ACCEL_NONE, "no hierarchy"), (pt_amd.ACCEL_AUTO, "hierarchy")):
        ctx.set_accel(accel)
        best = 1e9
        for _ in range(3):
            t0 = time.perf_counter()
            ctx.set_scene(sc.surfaces, sc.materials, sc.bg)
            best = min(best, time.perf_counter() - t0)
        print("%-16s %8d surfaces, %-13s pt_set_scene %.2f ms" % (name, len(sc.surfaces), label, best * 1e3), flush=True)
ctx.close()
