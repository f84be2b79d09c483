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
