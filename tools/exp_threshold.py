"""Scan against hierarchy for scenes of a few dozen to a few hundred small triangles
(heightfield:N has 2*N*N of them plus walls and spheres): kernel ms at 800x800, 64 spp."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "path-tracer_amd"))
import torch; torch.cuda.init()
import pt_amd
ctx = pt_amd.Context(0)
spp = int(sys.argv[1]) if len(sys.argv) > 1 else 64
for q in (2, 3, 4, 6, 11):
    sc = pt_amd.builtin_scene("heightfield:%d" % q, 800, 800)
    row = []
    for accel in (pt_amd.ACCEL_NONE, pt_amd.ACCEL_BVH):
        ctx.set_accel(accel)
        ctx.load(sc)
        for i in range(2): ctx.render(spp, 0, 1, 0)
        ks = []
        for i in range(3):
            ctx.render(spp, 0, 1, 0); ks.append(ctx.counters().kernel_ms)
        row.append(min(ks))
    print("%4d small triangles (%d surfaces): scan %7.2f ms  hierarchy %7.2f ms" % (2 * q * q, len(sc.surfaces), row[0], row[1]), flush=True)
ctx.set_accel(pt_amd.ACCEL_AUTO)
