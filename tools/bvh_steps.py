"""Traversal work per segment on the synthetic height fields (PT_MODE_COUNT_STEPS): node
records opened (two box tests each) and leaf triangles tested, beside the 'effective' tests the
reference's scan of every surface would make, and the kernel time with and without counting."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "path-tracer_amd"))
import pt_amd  # noqa: E402

ctx = pt_amd.Context(0)
for name, spp in (("heightfield:40", 64), ("heightfield:200", 64), ("heightfield", 64)):
    sc = pt_amd.builtin_scene(name, 800, 800)
    ctx.load(sc)
    for numeric, label in ((pt_amd.NUM_EXACT, "exact"), (pt_amd.NUM_FAST, "fast")):
        ctx.render(spp, 0, 1, numeric)
        plain = ctx.counters()
        ctx.render(spp, spp, 1, numeric | pt_amd.MODE_COUNT_STEPS)
        c = ctx.counters()
        print("%-16s %7d surfaces %s: %.2f ms (%.2f ms counting); per segment: %.1f nodes opened, %.2f leaves tested, "
              "%d surfaces scanned outside the hierarchy; effective tests %d" % (
                  name, len(sc.surfaces), label, plain.kernel_ms, c.kernel_ms, c.node_steps / c.segments,
                  c.leaf_steps / c.segments, len(sc.surfaces) - sc.n_triangles + 5, len(sc.surfaces)), flush=True)
ctx.close()
