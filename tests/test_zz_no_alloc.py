"""pt_render_into_static, the entry point that allocates nothing (its work queue and partial
sums live in a static device array), against pt_render_into through the C ABI.

Bar: the two entry points produce the same image bit for bit whenever they cut the samples
into the same chunks (images small enough that the static area does not bind); where the
static area forces fewer, longer chunks the sums differ by the order of fp32 additions
only, held to 2e-6 relative per pixel sum.  Two calls in a row, the second with a smaller
input, so that whatever the first call left in the shared area cannot leak into the second.
"""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _both(ctx, pt, torch, scene, spp, seed, mode=0):
    ctx.load(scene)
    h, w = scene.height, scene.width
    ctx.set_stream(torch.cuda.current_stream().cuda_stream)
    try:
        a = torch.zeros(h, w, 3, dtype=torch.float32, device="cuda:0")
        b = torch.zeros(h, w, 3, dtype=torch.float32, device="cuda:0")
        ctx.render_into(a.data_ptr(), spp, 3, seed, mode)
        ca = ctx.counters()
        ctx.render_into_static(b.data_ptr(), spp, 3, seed, mode)
        cb = ctx.counters()
        torch.cuda.synchronize()
    finally:
        ctx.set_stream(0)
    assert ca.samples == cb.samples == w * h * spp
    assert ca.segments == cb.segments
    return a.cpu().numpy(), b.cpu().numpy()


@pytest.mark.parametrize("mode_name", ["exact", "fast"])
def test_static_scratch_render_equals_render_into_twice_in_a_row(ctx, pt, mode_name):
    import torch
    mode = pt.NUM_FAST if mode_name == "fast" else pt.NUM_EXACT
    # first call: 800 x 800 at 64 spp; the static area holds 4 chunks of it, pt_render_into
    # takes more, so the additions are ordered differently
    a, b = _both(ctx, pt, torch, pt.builtin_scene("cornell_2", 800, 800), 64, seed=7, mode=mode)
    assert a.sum() > 0
    assert np.allclose(a, b, rtol=2e-6, atol=1e-6)
    # second call, smaller (and another scene): both entry points take one sample per work
    # item here, so the images agree bit for bit; nothing of the first call may show
    a, b = _both(ctx, pt, torch, pt.builtin_scene("orange_plane", 96, 64), 9, seed=8, mode=mode)
    assert a.sum() > 0
    assert np.array_equal(a, b)


def test_static_scratch_through_the_hierarchy(ctx, pt):
    import torch
    sc = pt.builtin_scene("heightfield:12", 120, 80)
    a, b = _both(ctx, pt, torch, sc, 5, seed=3)
    assert ctx.counters().accel == pt.ACCEL_BVH
    assert np.array_equal(a, b)
    a, b = _both(ctx, pt, torch, pt.builtin_scene("cornell", 40, 40), 3, seed=3)
    assert np.array_equal(a, b)


def test_static_scratch_limits(ctx, pt):
    import torch
    assert pt.STATIC_SCRATCH_BYTES == 32 << 20
    sc = pt.builtin_scene("cornell_2", 2048, 1366)          # 2 797 568 pixels: one too many rows
    ctx.load(sc)
    buf = torch.zeros(1366, 2048, 3, dtype=torch.float32, device="cuda:0")
    with pytest.raises(pt.PtError) as e:
        ctx.render_into_static(buf.data_ptr(), 1, 0, 1, 0)
    assert e.value.code == -4 and "static scratch" in str(e.value)
    sc = pt.builtin_scene("cornell_2", 2048, 1365)          # 2 795 520 pixels fit
    ctx.load(sc)
    ctx.render_into_static(buf.data_ptr(), 1, 0, 1, 0)
    torch.cuda.synchronize()
    sc = pt.builtin_scene("cornell_2", 32, 32)
    ctx.load(sc)
    with pytest.raises(pt.PtError) as e:                    # the wavefront layout allocates
        ctx.render_into_static(buf.data_ptr(), 1, 0, 1, pt.MODE_WAVEFRONT)
    assert e.value.code == -1
    with pytest.raises(pt.PtError):                         # not device memory
        ctx.render_into_static(0, 1, 0, 1, 0)
    host = np.zeros((32, 32, 3), np.float32)
    with pytest.raises(pt.PtError):
        ctx.render_into_static(host.ctypes.data, 1, 0, 1, 0)


def test_static_scratch_call_leaves_device_memory_as_it_found_it(ctx, pt):
    """What the entry point's name promises: the device's free memory (hipMemGetInfo) is the same
    before and after the call - scan, hierarchy and both numeric policies - once a first call has
    loaded the kernels (the runtime sets up code objects and the private segment the first time a
    kernel runs; that is the runtime's, once per process, not the call's)."""
    import torch
    for name, w, h, spp in (("cornell_2", 320, 200, 6), ("heightfield:12", 160, 120, 4)):
        sc = pt.builtin_scene(name, w, h)
        ctx.load(sc)
        buf = torch.zeros(h, w, 3, dtype=torch.float32, device="cuda:0")
        ctx.set_stream(torch.cuda.current_stream().cuda_stream)
        try:
            for mode in (pt.NUM_EXACT, pt.NUM_FAST):
                ctx.render_into_static(buf.data_ptr(), spp, 0, 1, mode)       # warm-up
                torch.cuda.synchronize()
                free_before, _ = torch.cuda.mem_get_info(0)
                for k in range(3):
                    ctx.render_into_static(buf.data_ptr(), spp, spp * (k + 1), 1, mode)
                torch.cuda.synchronize()
                free_after, _ = torch.cuda.mem_get_info(0)
                assert free_after == free_before, (name, mode, free_before - free_after)
        finally:
            ctx.set_stream(0)


def test_static_scratch_call_under_stream_capture(ctx, pt):
    """The callers the entry point is for: captured into a HIP graph (where an allocation would
    fail the capture), then replayed twice - each replay renders the image pt_render_into renders."""
    import torch
    sc = pt.builtin_scene("cornell_2", 200, 120)
    ctx.load(sc)
    spp, seed = 7, 5
    want = torch.zeros(120, 200, 3, dtype=torch.float32, device="cuda:0")
    got = torch.zeros_like(want)
    stream = torch.cuda.Stream()
    ctx.set_stream(stream.cuda_stream)
    try:
        ctx.render_into(want.data_ptr(), spp, 0, seed, 0)
        ctx.render_into_static(got.data_ptr(), spp, 0, seed, 0)               # warm-up, not captured
        stream.synchronize()
        assert torch.equal(got, want)
        got.zero_()
        torch.cuda.synchronize()
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph, stream=stream):
            ctx.render_into_static(got.data_ptr(), spp, 0, seed, 0)
        torch.cuda.synchronize()
        assert not got.any()                       # captured, not run
        with pytest.raises(pt.PtError):            # a captured call has no counters
            ctx.counters()
        for _ in range(2):
            graph.replay()
            torch.cuda.synchronize()
            assert torch.equal(got, want)
            got.zero_()
        # and the context works as before once the capture is over
        ctx.render_into_static(got.data_ptr(), spp, 0, seed, 0)
        stream.synchronize()
        assert torch.equal(got, want) and ctx.counters().samples == 200 * 120 * spp
    finally:
        ctx.set_stream(0)


def test_static_scratch_symbol_is_exported(pt):
    lib = pt.load()
    assert lib.pt_render_into_static.restype is C.c_int
