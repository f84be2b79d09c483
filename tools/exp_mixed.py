"""Scan against hierarchy when most surfaces are spheres (scanned either way) and only some are
small triangles: glass_stress's 85 surfaces plus heightfield:N's, 800x800, 64 spp, kernel ms."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "path-tracer_amd"))
import torch; torch.cuda.init()
import numpy as np
import pt_amd as pt
ctx = pt.Context(0)
a = pt.builtin_scene("glass_stress", 800, 800)
for q in (3, 5, 8, 16):
    b = pt.builtin_scene("heightfield:%d" % q, 800, 800)
    small = b.surfaces[:2 * q * q] if False else b.surfaces
    surf = np.concatenate([a.surfaces, small])
    surf["material"][len(a.surfaces):] += len(a.materials)
    mats = np.concatenate([a.materials, b.materials])
    sc = pt.SceneData(surf, mats, a.bg, a.camera)
    row = []
    for accel in (pt.ACCEL_NONE, pt.ACCEL_BVH):
        ctx.set_accel(accel); ctx.load(sc)
        ctx.render(64, 0, 1, 0); ctx.render(64, 0, 1, 0)
        c = ctx.counters(); row.append((c.kernel_ms, c.accel))
    print("85 + %d surfaces (%d small triangles): scan %.2f ms, hierarchy %.2f ms" % (len(b.surfaces), 2 * q * q, row[0][0], row[1][0]), flush=True)
ctx.set_accel(pt.ACCEL_AUTO)
