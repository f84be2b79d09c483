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

# Wall time per call when calls are queued back to back (what an interactive loop pays per frame):
# the render call alone, then with the tone map and the image read back every frame.
for w, h in ((356, 200), (800, 800)):
    scene = pt_amd.builtin_scene("cornell_2", w, h)
    ctx = pt_amd.Context(0); ctx.load(scene)
    for i in range(10): ctx.render(1, i, 1, 0)
    ctx.read_accum()
    n = 200
    t0 = time.perf_counter()
    for i in range(n): ctx.render(1, 10 + i, 1, 0)
    ctx.read_accum()
    t1 = time.perf_counter()
    for i in range(n):
        ctx.render(1, 10 + n + i, 1, 0)
        ctx.tonemap_rgb8()
    t2 = time.perf_counter()
    print("%dx%d 1 spp per call: %.0f us per pt_render queued back to back, %.0f us per frame with tone map and read-back" % (
        w, h, (t1 - t0) / n * 1e6, (t2 - t1) / n * 1e6), flush=True)
    ctx.close()
