"""Scenes of many small spheres: the scan of every surface (with its wave-level early-outs) against
the hierarchy with spheres as leaves, kernel time at 800 x 800.  A lattice of n^3 glass and
diffuse spheres inside the Cornell walls of glass_stress, under its sphere light."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "path-tracer_amd"))
import pt_amd as pt  # noqa: E402

base = pt.builtin_scene("glass_stress", 800, 800)
keep = [i for i in range(len(base.surfaces)) if base.surfaces["kind"][i] == pt.TRIANGLE or base.surfaces["b"][i][0] >= 0.5]
ctx = pt.Context(0)
for side, spp in ((4, 32), (6, 32), (8, 16), (12, 16), (16, 8)):
    rng = np.random.default_rng(side)
    g = np.stack(np.meshgrid(*[np.arange(side)] * 3, indexing="ij"), -1).reshape(-1, 3)
    pitch = 1.7 / side
    centres = np.array([-0.85, -0.7, -0.8]) + pitch * (g + 0.5) + rng.uniform(-0.1, 0.1, g.shape) * pitch
    n = len(centres)
    surf = np.zeros(len(keep) + n, dtype=pt.SURFACE)
    surf[:len(keep)] = base.surfaces[keep]
    surf["kind"][len(keep):] = pt.SPHERE
    surf["a"][len(keep):] = centres
    surf["b"][len(keep):, 0] = 0.3 * pitch
    mats = np.zeros(len(base.materials) + 2, dtype=pt.MATERIAL)
    mats[:len(base.materials)] = base.materials
    mats[-2]["type"] = 2
    mats[-2]["color"] = (1, 1, 1)
    mats[-1]["type"] = 0
    mats[-1]["color"] = (0.9, 0.6, 0.3)
    surf["material"][len(keep):] = len(base.materials) + (rng.random(n) < 0.5)
    out = []
    for accel in (pt.ACCEL_NONE, pt.ACCEL_BVH, pt.ACCEL_AUTO):
        ctx.set_accel(accel)
        ctx.set_scene(surf, mats, base.bg)
        ctx.set_camera(base.camera)
        for numeric in (pt.NUM_EXACT, pt.NUM_FAST):
            ctx.render(spp, 0, 1, numeric)
            ctx.render(spp, spp, 1, numeric)
            c = ctx.counters()
            out.append("%.2f ms%s" % (c.kernel_ms, " (hierarchy)" if c.accel == pt.ACCEL_BVH else ""))
    print("%5d spheres + %d others, %2d spp, exact / fast: scan %s / %s   hierarchy %s / %s   auto %s / %s" % (
        n, len(keep), spp, *out), flush=True)
ctx.set_accel(pt.ACCEL_AUTO)
ctx.close()
