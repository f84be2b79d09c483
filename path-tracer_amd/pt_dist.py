"""Multi-GPU render step: one process per GPU, samples-per-pixel partitioned by
global sample number (or, on request, the image partitioned into bands of rows),
images combined with ONE reduce (RCCL over xGMI when the process group's backend
is "nccl").

The random stream of a path depends only on (seed, pixel, global sample
number), so the union of the ranks' samples is the same set of paths whatever
the number of GPUs: an N-GPU image differs from the 1-GPU image only by the
order of fp32 additions.  The reference has no counterpart (it is a single
thread); its running sum `buffer[u][v] += color` (src/main.cpp:139) is what
makes the partition exact.
"""


def partition_samples(total_spp, world, rank, first_sample=0):
    """Contiguous share of global samples [first_sample, first_sample+total_spp)
    for `rank` of `world`: returns (first, count).  Shares differ by at most one
    sample and tile the range exactly; a share may be empty when world > total_spp."""
    if world < 1 or not 0 <= rank < world:
        raise ValueError("rank %d of world %d" % (rank, world))
    if total_spp < 0:
        raise ValueError("total_spp must be non-negative")
    lo = rank * total_spp // world
    hi = (rank + 1) * total_spp // world
    return first_sample + lo, hi - lo


def render_share(renderer, accum, total_spp, world, rank, first_sample=0, seed=1, mode=0):
    """Adds this rank's share of the samples into `accum` (a torch tensor of
    height x width x 3 float32 on the renderer's device).  `renderer` offers
    render_into(device_ptr, spp, first_sample, seed, mode), as pt_amd.Context does.
    The launch is put on torch's current stream (renderer.set_stream, when the renderer
    has one and accum is a device tensor), the stream the reduce that follows runs on:
    the library does not wait for its kernels, so on any other stream the reduce could
    read an image still being written."""
    first, count = partition_samples(total_spp, world, rank, first_sample)
    if count > 0:
        if getattr(accum, "is_cuda", False) and hasattr(renderer, "set_stream"):
            import torch
            renderer.set_stream(torch.cuda.current_stream(accum.device).cuda_stream)
        renderer.render_into(accum.data_ptr(), count, first, seed, mode)
    return count


def partition_rows(height, world, rank):
    """Contiguous band of image rows for `rank` of `world`: returns (first_row, n_rows).  The other
    way to share an image among GPUs: every rank renders all samples of its rows.  Bands differ by
    at most one row and tile the image; a band may be empty when world > height."""
    if world < 1 or not 0 <= rank < world:
        raise ValueError("rank %d of world %d" % (rank, world))
    if height < 0:
        raise ValueError("height must be non-negative")
    lo = rank * height // world
    hi = (rank + 1) * height // world
    return lo, hi - lo


def render_band(renderer, accum, spp, world, rank, first_sample=0, seed=1, mode=0):
    """Adds all `spp` samples of this rank's band of rows into `accum` (the whole image, height x
    width x 3 float32 on the renderer's device; rows outside the band are left alone, so with
    buffers that start at zero the same reduce_image combines the bands - adding zeros is exact,
    and the combined image does not depend on the number of ranks at all).  `renderer` offers
    render_rows(device_ptr, first_row, n_rows, spp, first_sample, seed, mode), as pt_amd.Context
    does.  Rows cost what their paths cost, so bands are not as evenly loaded as sample shares:
    the sample partition (render_share) is the default."""
    first_row, n_rows = partition_rows(int(accum.shape[0]), world, rank)
    if n_rows > 0:
        if getattr(accum, "is_cuda", False) and hasattr(renderer, "set_stream"):
            import torch
            renderer.set_stream(torch.cuda.current_stream(accum.device).cuda_stream)
        renderer.render_rows(accum.data_ptr(), first_row, n_rows, spp, first_sample, seed, mode)
    return first_row, n_rows


def reduce_image(accum, dst=0, group=None, always=False):
    """Sums the ranks' accumulation buffers onto rank `dst` with one collective.  A group of one
    rank has nothing to add and skips the call, unless `always` asks for it (to exercise the
    backend on a one-GPU box)."""
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized() and (dist.get_world_size(group) > 1 or always):
        if accum.is_cuda and dist.get_backend(group) == "gloo":
            # gloo has no reduce for device tensors: stage through the host.  Only
            # rehearsals of the multi-rank path on a box with fewer GPUs than ranks
            # come here; with RCCL ("nccl") the reduce runs on the device buffers.
            host = accum.cpu()
            dist.reduce(host, dst=dst, op=dist.ReduceOp.SUM, group=group)
            if dist.get_rank(group) == dst:
                accum.copy_(host)
        else:
            dist.reduce(accum, dst=dst, op=dist.ReduceOp.SUM, group=group)
    return accum


def adopt_image(renderer, accum, total_spp):
    """On the destination rank after reduce_image: makes the combined image (and the sample
    count of all ranks together) the renderer's own, so that its tone map and read-back see it.
    `renderer` offers write_accum_device(device_ptr, total_samples), as pt_amd.Context does."""
    renderer.write_accum_device(accum.data_ptr(), total_spp)
