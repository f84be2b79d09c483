"""Runs the closest-hit search on its own (k_primary_hits over the camera rays of a
4096 x 4096 image of a built-in scene), a few launches per numeric policy, and prints the
best kernel time of each: the program tools/profile_primary.sh puts under rocprofv3."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "path-tracer_amd"))
import pt_amd  # noqa: E402

scene = sys.argv[1] if len(sys.argv) > 1 else "cornell_2"
side = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
ctx = pt_amd.Context(0)
ctx.load(pt_amd.builtin_scene(scene, side, side))
for name, numeric in (("exact", pt_amd.NUM_EXACT), ("fast", pt_amd.NUM_FAST)):
    best = None
    for _ in range(3):
        ctx.primary_hits(numeric)
        c = ctx.counters()
        best = c.kernel_ms if best is None else min(best, c.kernel_ms)
    print("%s %s %dx%d: k_primary_hits %.4f ms, %.3e tests/s, %d VGPRs" % (
        scene, name, side, side, best, c.tests / (best * 1e-3), c.vgprs))
ctx.close()
