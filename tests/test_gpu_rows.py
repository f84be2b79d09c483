"""pt_render_rows: a band of the image's rows rendered on its own (the unit of the row-band way
of sharing an image among GPUs), through the C ABI, against pt_render_into of the whole image."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("name,w,h,spp,mode_name", [("cornell_2", 160, 90, 24, "exact"), ("orange_plane", 97, 61, 9, "fast"),
                                                    ("heightfield:12", 120, 80, 6, "exact")])
def test_bands_make_up_the_whole_image(ctx, pt, name, w, h, spp, mode_name):
    import torch
    mode = pt.NUM_FAST if mode_name == "fast" else pt.NUM_EXACT
    sc = pt.builtin_scene(name, w, h)
    ctx.load(sc)
    ctx.set_stream(torch.cuda.current_stream().cuda_stream)
    try:
        whole = torch.zeros(h, w, 3, dtype=torch.float32, device="cuda:0")
        ctx.render_into(whole.data_ptr(), spp, 3, 7, mode)
        seg_whole = ctx.counters().segments
        bands = torch.zeros_like(whole)
        cuts = [0, h // 3, h // 3 + 1, h]                      # three bands, one of a single row
        seg = samples = 0
        for lo, hi in zip(cuts, cuts[1:]):
            before = bands.clone()
            ctx.render_rows(bands.data_ptr(), lo, hi - lo, spp, 3, 7, mode)
            c = ctx.counters()
            seg += c.segments
            samples += c.samples
            torch.cuda.synchronize()
            # only the band's rows changed
            assert torch.equal(bands[:lo], before[:lo]) and torch.equal(bands[hi:], before[hi:])
            assert bands[lo:hi].sum() > 0
        # the same paths pixel by pixel (a pixel's streams are named by its place in the whole image):
        # the same segment count, and sums that differ only by how the samples were cut into chunks
        assert samples == w * h * spp and seg == seg_whole
        a, b = whole.cpu().numpy(), bands.cpu().numpy()
        assert np.allclose(a, b, rtol=3e-6, atol=1e-6)
        # a second helping adds to what is there
        ctx.render_rows(bands.data_ptr(), 0, h, spp, 3 + spp, 7, mode)
        ctx.render_into(whole.data_ptr(), spp, 3 + spp, 7, mode)
        torch.cuda.synchronize()
        assert np.allclose(whole.cpu().numpy(), bands.cpu().numpy(), rtol=3e-6, atol=1e-6)
    finally:
        ctx.set_stream(0)


def test_render_rows_arguments(ctx, pt):
    import torch
    sc = pt.builtin_scene("cornell_2", 40, 30)
    ctx.load(sc)
    buf = torch.zeros(30, 40, 3, dtype=torch.float32, device="cuda:0")
    ctx.render_rows(buf.data_ptr(), 10, 0, 4)                  # an empty band: nothing happens
    torch.cuda.synchronize()
    assert not buf.any()
    for first, n in ((-1, 5), (0, 31), (29, 2), (5, -1)):
        with pytest.raises(pt.PtError):
            ctx.render_rows(buf.data_ptr(), first, n, 4)
    with pytest.raises(pt.PtError):                            # the wavefront layout renders whole images
        ctx.render_rows(buf.data_ptr(), 0, 10, 4, 0, 1, pt.MODE_WAVEFRONT)
    with pytest.raises(pt.PtError):
        ctx.render_rows(0, 0, 10, 4)
