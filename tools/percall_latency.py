"""Device time of small render calls (the reference's usage: one sample per pixel per call).
   python tools/percall_latency.py
"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "path-tracer_amd"))
import torch; torch.cuda.init()
import pt_amd
for w, h, spp in ((356, 200, 1), (800, 800, 1), (800, 800, 4), (800, 800, 16), (800, 800, 64)):
    scene = pt_amd.builtin_scene("cornell_2", w, h)
    ctx = pt_amd.Context(0); ctx.load(scene)
    acc = torch.zeros(h, w, 3, device="cuda")
    for i in range(5): ctx.render_into(acc.data_ptr(), spp, i * spp, 1, 0)
    torch.cuda.synchronize()
    ks = []
    for i in range(20):
        ctx.render_into(acc.data_ptr(), spp, (5 + i) * spp, 1, 0)
        ks.append(ctx.counters().kernel_ms * 1e3)
    ks.sort()
    print("%dx%d %d spp: kernel median %.1f us, blocks %d" % (w, h, spp, ks[len(ks) // 2], ctx.counters().grid_blocks), flush=True)
    ctx.close()
