"""Closest-hit query kernels, rays per lane against kernel time (PT_QUERY_RAYS = 0: the one-ray
kernels that read records through the scalar cache; 1, 2, 4, 8: N rays per lane, records
broadcast from LDS into vector registers).  EXACT results must equal the one-ray kernel's bit
for bit; FAST results are compared and the differences counted.

    python tools/exp_query.py [side] [scene ...]
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "path-tracer_amd"))
import pt_amd  # noqa: E402

PEAK = 157.3e12
side = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
scenes = sys.argv[2:] or ["cornell_2", "cornell", "orange_plane", "glass_stress"]
for scene in scenes:
    ctx = pt_amd.Context(0)
    ctx.load(pt_amd.builtin_scene(scene, side, side))
    for name, numeric in (("exact", pt_amd.NUM_EXACT), ("fast", pt_amd.NUM_FAST)):
        ref = None
        for n in (0, 1, 2, 4, 8):
            os.environ["PT_QUERY_RAYS"] = str(n)
            best = None
            for _ in range(4):
                idx, dist = ctx.primary_hits(numeric)
                c = ctx.counters()
                best = c.kernel_ms if best is None else min(best, c.kernel_ms)
            if ref is None:
                ref = (idx, dist)
            same_idx = int((idx != ref[0]).sum())
            same_dist = int((dist.view(np.uint32) != ref[1].view(np.uint32)).sum())
            print("%-13s %-5s N=%d: %.4f ms  frac %.3f  vgprs %3d  grid %6d  idx diffs %d  dist bit diffs %d" % (
                scene, name, n, best, c.flops / (best * 1e-3) / PEAK, c.vgprs, c.grid_blocks, same_idx, same_dist),
                flush=True)
    ctx.close()
os.environ.pop("PT_QUERY_RAYS", None)
