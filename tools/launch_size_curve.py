"""Kernel time of pt_render against the launch size (samples per pixel at 800 x 800).
   python tools/launch_size_curve.py [scene]
"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "path-tracer_amd"))
import torch; torch.cuda.init()
import pt_amd
scene = pt_amd.builtin_scene(sys.argv[1] if len(sys.argv) > 1 else "cornell_2", 800, 800)
ctx = pt_amd.Context(0); ctx.load(scene)
acc = torch.zeros(800, 800, 3, device="cuda")
for spp in (1, 2, 4, 8, 12, 16, 24, 32, 48, 64, 96, 128, 256) + ((1024,) if len(scene.surfaces) < 100 else ()):
    for i in range(3): ctx.render_into(acc.data_ptr(), spp, i * spp, 1, 0)
    ks = []
    for i in range(9):
        ctx.render_into(acc.data_ptr(), spp, (5 + i) * spp, 1, 0)
        c = ctx.counters(); ks.append((c.kernel_ms, c.total_ms))
    ks.sort()
    k, t = ks[len(ks) // 2]
    print("%4d spp: kernel %8.1f us  with reduce %8.1f us  blocks %4d  %6.2f G samples/s" % (spp, k * 1e3, t * 1e3, c.grid_blocks, 640000 * spp / k / 1e6), flush=True)
