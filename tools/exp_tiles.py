"""The linear scan in the render kernel: records through the scalar cache (PT_ACCEL_NONE)
against records staged through LDS tiles shared by the workgroup (PT_ACCEL_LDS_TILES), on
scenes of 8 to 1360 surfaces at 800 x 800.  Prints the render kernel's time for both."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "path-tracer_amd"))
import pt_amd  # noqa: E402

ctx = pt_amd.Context(0)
for name, spp in (("cornell_2", 64), ("cornell", 64), ("glass_stress", 16), ("heightfield:7", 16), ("heightfield:11", 8),
                  ("heightfield:17", 4), ("heightfield:26", 2)):
    sc = pt_amd.builtin_scene(name, 800, 800)
    row = []
    for accel in (pt_amd.ACCEL_NONE, pt_amd.ACCEL_LDS_TILES):
        for numeric in (pt_amd.NUM_EXACT, pt_amd.NUM_FAST):
            ctx.set_accel(accel)
            ctx.load(sc)
            best = None
            for k in range(3):
                ctx.render(spp, k * spp, 1, numeric)
                c = ctx.counters()
                best = c.kernel_ms if best is None else min(best, c.kernel_ms)
            row.append(best)
    print("%-16s %5d surfaces, %3d spp: scalar cache %8.2f ms exact %8.2f fast | LDS tiles %8.2f exact %8.2f fast (%d VGPRs)" % (
        name, len(sc.surfaces), spp, row[0], row[1], row[2], row[3], c.vgprs), flush=True)
ctx.close()
