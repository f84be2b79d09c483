"""pt_group_*: several GPUs behind the C ABI (one context per device, samples partitioned by
global sample number, ONE ncclReduce onto the first device).  The GPU test uses as many
devices as the box shows (one is legal for ncclCommInitAll); the partition itself is a pure
function and is tested on the CPU against the Python rule of pt_dist."""
import numpy as np
import pytest


def test_group_share_tiles_the_range(pt):
    import pt_dist
    for spp in (0, 1, 7, 64, 1000, 4096):
        for n in (1, 2, 3, 8):
            got = [pt.group_share(spp, n, r, 5) for r in range(n)]
            assert got == [pt_dist.partition_samples(spp, n, r, 5) for r in range(n)]
            assert sum(c for _, c in got) == spp
            pos = 5
            for first, count in got:
                assert first == pos
                pos += count


def test_group_share_of_a_member_that_does_not_exist_is_empty(pt):
    # no division by zero for n < 1, and no samples beyond the request for a rank outside [0, n)
    for n, rank in ((0, 0), (-1, 0), (-5, -5), (4, -1), (4, 4), (4, 9), (1, 1)):
        assert pt.group_share(1000, n, rank, 7) == (7, 0)


@pytest.mark.gpu
def test_group_image_equals_one_context(ctx, pt):
    import torch
    n = torch.cuda.device_count()
    assert n >= 1
    sc = pt.builtin_scene("cornell_2", 160, 120)
    ctx.load(sc)
    ctx.render(24, 0, 5, 0)
    ctx.render(8, 24, 5, 0)
    want, _ = ctx.read_accum()
    with pt.Group(list(range(n))) as g:
        assert g.size == n
        g.load(sc)
        g.render(24, 0, 5, 0)            # two render calls, one reduce
        g.render(8, 24, 5, 0)
        g.reduce()
        got, total = g.read_accum()
        assert total == 32
        if n == 1:
            assert np.array_equal(got, want)
        else:
            assert np.allclose(got, want, rtol=1e-5, atol=1e-5)      # the order of fp32 additions
        seg = sum(g.counters(i).segments for i in range(n))
        assert seg > 0
        rgb = g.tonemap_rgb8()
        assert rgb.shape == (120, 160, 3) and rgb.max() > 0
        # more samples after a reduce: the image keeps accumulating; read_accum reduces by itself
        g.render(4, 32, 5, 0)
        more, total = g.read_accum()
        assert total == 36
        ctx.render(4, 32, 5, 0)
        want2, _ = ctx.read_accum()
        assert np.allclose(more, want2, rtol=1e-5, atol=1e-5)
        g.clear()
        zero, total = g.read_accum()
        assert total == 0 and not zero.any()
        # another image size, through the hierarchy
        sc2 = pt.builtin_scene("heightfield:12", 96, 64)
        g.load(sc2)
        g.render(6, 0, 2, pt.NUM_FAST)
        a, _ = g.read_accum()
        ctx.load(sc2)
        ctx.render(6, 0, 2, pt.NUM_FAST)
        b, _ = ctx.read_accum()
        assert np.allclose(a, b, rtol=1e-5, atol=1e-5)


@pytest.mark.gpu
def test_group_errors(pt):
    with pytest.raises(pt.PtError):
        pt.Group([])
    with pytest.raises(pt.PtError):
        pt.Group([0, 0])
    with pytest.raises(pt.PtError):
        pt.Group([99])
    with pt.Group([0]) as g:
        with pytest.raises(pt.PtError):
            g.render(1)                   # no camera yet


@pytest.mark.gpu
def test_group_can_be_made_and_destroyed_repeatedly(ctx, pt):
    """Communicator, stream and image lifetimes: build, render, reduce, destroy - three times, and
    the third group's image equals the first's bit for bit (the same devices, the same shares)."""
    import torch
    n = torch.cuda.device_count()
    sc = pt.builtin_scene("orange_plane", 96, 64)
    images = []
    for _ in range(3):
        with pt.Group(list(range(n))) as g:
            g.load(sc)
            g.render(12, 0, 3, 0)
            g.reduce()
            g.render(4, 12, 3, 0)          # left unreduced: read_accum reduces
            got, total = g.read_accum()
            assert total == 16
            images.append(got)
    assert np.array_equal(images[0], images[1]) and np.array_equal(images[0], images[2])
    ctx.load(sc)
    ctx.render(16, 0, 3, 0)
    want, _ = ctx.read_accum()
    assert np.allclose(images[0], want, rtol=1e-5, atol=1e-5)


@pytest.mark.gpu
def test_two_contexts_on_one_device_interleaved_from_one_thread(pt):
    """What pt_group_render relies on: render calls on different contexts, each with a stream of
    its own, enqueued from one thread without waiting, do not disturb each other."""
    import torch
    sc_a = pt.builtin_scene("cornell_2", 120, 80)
    sc_b = pt.builtin_scene("red_corner", 64, 96)
    sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
    with pt.Context(0) as a, pt.Context(0) as b, pt.Context(0) as alone:
        a.load(sc_a)
        b.load(sc_b)
        a.set_stream(sa.cuda_stream)
        b.set_stream(sb.cuda_stream)
        img_a = torch.zeros(80, 120, 3, dtype=torch.float32, device="cuda")
        img_b = torch.zeros(96, 64, 3, dtype=torch.float32, device="cuda")
        torch.cuda.synchronize()
        for k in range(6):                  # enqueued alternately, nothing waits in between
            a.render_into(img_a.data_ptr(), 5, 5 * k, 9, 0)
            b.render_into(img_b.data_ptr(), 3, 3 * k, 4, pt.NUM_FAST)
        sa.synchronize()
        sb.synchronize()
        alone.load(sc_a)
        alone.render(30, 0, 9, 0)
        want_a, _ = alone.read_accum()
        alone.load(sc_b)
        alone.render(18, 0, 4, pt.NUM_FAST)
        want_b, _ = alone.read_accum()
        # the same work items in the same order per context: only the split into launches differs
        assert np.allclose(img_a.cpu().numpy(), want_a, rtol=1e-5, atol=1e-5)
        assert np.allclose(img_b.cpu().numpy(), want_b, rtol=1e-5, atol=1e-5)
        a.set_stream(0)
        b.set_stream(0)
