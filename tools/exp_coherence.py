"""How much the mix of materials costs: the headline scene as it is, and with every material
made diffuse (one shading branch), in kernel time per path segment."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "path-tracer_amd"))
import torch; torch.cuda.init()
import numpy as np
import pt_amd
base = pt_amd.builtin_scene("cornell_2", 800, 800)
ctx = pt_amd.Context(0)
for label, types in (("as it is", None), ("all diffuse", 0), ("all mirror but the light", 1)):
    mats = base.materials.copy()
    if types is not None:
        for m in mats:
            if types == 0 or m["emittance"].sum() == 0: m["type"] = types
    sc = pt_amd.SceneData(base.surfaces, mats, base.bg, base.camera)
    ctx.load(sc)
    for i in range(2): ctx.render(256, 0, 1, 0)
    ctx.render(256, 0, 1, 0)
    c = ctx.counters()
    print("%-26s kernel %7.2f ms  %5.2f segments/sample  %6.3f ns per segment" % (
        label, c.kernel_ms, c.segments / c.samples, c.kernel_ms * 1e6 / c.segments), flush=True)
