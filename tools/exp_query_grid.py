"""Closest-hit query kernels: rays per lane x workgroups per CU against kernel time.
    python tools/exp_query_grid.py [side] [scene]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "path-tracer_amd"))
import pt_amd  # noqa: E402

PEAK = 157.3e12
side = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
scene = sys.argv[2] if len(sys.argv) > 2 else "cornell_2"
ctx = pt_amd.Context(0)
ctx.load(pt_amd.builtin_scene(scene, side, side))
for name, numeric in (("exact", pt_amd.NUM_EXACT), ("fast", pt_amd.NUM_FAST)):
    for n in (1, 2, 4, 8):
        for grid in (0, 2, 4, 6, 8, 16):
            os.environ["PT_QUERY_RAYS"] = str(n)
            os.environ["PT_QUERY_GRID"] = str(grid)
            best = None
            for _ in range(5):
                ctx.primary_hits(numeric)
                c = ctx.counters()
                best = c.kernel_ms if best is None else min(best, c.kernel_ms)
            print("%-10s %-5s N=%d grid/CU=%2d: %.4f ms  frac %.3f  vgprs %3d  blocks %6d" % (
                scene, name, n, grid, best, c.flops / (best * 1e-3) / PEAK, c.vgprs, c.grid_blocks), flush=True)
ctx.close()
