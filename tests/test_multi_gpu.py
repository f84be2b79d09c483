"""The sample-partition + single-reduce path on CPU: two processes over gloo.
Each rank renders its share with the CPU oracle standing in for the GPU (same
render_into contract), the images are reduced to rank 0 and compared with the
one-process render of all samples."""
import os
import socket
import sys

import numpy as np
import pytest

from conftest import ROOT


def test_partition_tiles_the_range(pt):
    sys.path.insert(0, os.path.join(ROOT, "path-tracer_amd"))
    import pt_dist
    for total in (0, 1, 7, 64, 1024, 4097):
        for world in (1, 2, 3, 4, 8):
            parts = [pt_dist.partition_samples(total, world, r, first_sample=5) for r in range(world)]
            assert parts[0][0] == 5
            assert sum(c for _, c in parts) == total
            for (f0, c0), (f1, _) in zip(parts, parts[1:]):
                assert f0 + c0 == f1
            counts = [c for _, c in parts]
            assert max(counts) - min(counts) <= 1
    with pytest.raises(ValueError):
        pt_dist.partition_samples(8, 2, 2)


def test_row_bands_tile_the_image(pt):
    sys.path.insert(0, os.path.join(ROOT, "path-tracer_amd"))
    import pt_dist
    for height in (0, 1, 7, 200, 2160):
        for world in (1, 2, 3, 8):
            bands = [pt_dist.partition_rows(height, world, r) for r in range(world)]
            assert bands[0][0] == 0 and sum(n for _, n in bands) == height
            for (f0, n0), (f1, _) in zip(bands, bands[1:]):
                assert f0 + n0 == f1
            assert max(n for _, n in bands) - min(n for _, n in bands) <= 1
    with pytest.raises(ValueError):
        pt_dist.partition_rows(8, 2, 2)


class OracleRenderer:
    """render_into on a CPU tensor, backed by the oracle (test stand-in for a GPU context)."""

    def __init__(self, orc, scene, tensors):
        self.orc, self.scene, self.tensors = orc, scene, tensors

    def render_into(self, ptr, spp, first_sample, seed, mode):
        t = self.tensors[ptr]
        acc = t.numpy()
        self.orc.render(self.scene, spp, first_sample=first_sample, seed=seed, threads=2, accum=acc)

    def render_rows(self, ptr, first_row, n_rows, spp, first_sample, seed, mode):
        # the oracle renders whole images: the band is cut out of one (a stand-in, as above)
        acc = self.tensors[ptr].numpy()
        whole, _ = self.orc.render(self.scene, spp, first_sample=first_sample, seed=seed, threads=2)
        acc[first_row:first_row + n_rows] += whole[first_row:first_row + n_rows]

    def write_accum_device(self, ptr, total_samples):
        self.adopted = (self.tensors[ptr].numpy().copy(), total_samples)


def _worker(rank, world, port, out_path):
    import torch
    import torch.distributed as dist
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    sys.path.insert(0, os.path.join(ROOT, "path-tracer_amd"))
    import pt_dist
    import pt_oracle
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    scene = pt_oracle.builtin_scene("orange_plane", 40, 24)
    accum = torch.zeros(24, 40, 3, dtype=torch.float32)
    r = OracleRenderer(pt_oracle, scene, {accum.data_ptr(): accum})
    n = pt_dist.render_share(r, accum, total_spp=9, world=world, rank=rank, first_sample=3, seed=11)
    assert n == (4 if rank == 0 else 5)
    pt_dist.reduce_image(accum, dst=0)
    if rank == 0:
        pt_dist.adopt_image(r, accum, 9)
        image, total = r.adopted
        assert total == 9 and np.array_equal(image, accum.numpy())
        np.save(out_path, accum.numpy())
    dist.barrier()
    dist.destroy_process_group()


def _band_worker(rank, world, port, out_path):
    import torch
    import torch.distributed as dist
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    sys.path.insert(0, os.path.join(ROOT, "path-tracer_amd"))
    import pt_dist
    import pt_oracle
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    scene = pt_oracle.builtin_scene("cornell_2", 36, 25)
    accum = torch.zeros(25, 36, 3, dtype=torch.float32)
    r = OracleRenderer(pt_oracle, scene, {accum.data_ptr(): accum})
    band = pt_dist.render_band(r, accum, spp=5, world=world, rank=rank, first_sample=2, seed=4)
    assert band == ((0, 12) if rank == 0 else (12, 13))
    outside = accum.numpy()[12:] if rank == 0 else accum.numpy()[:12]
    assert not outside.any()
    pt_dist.reduce_image(accum, dst=0)
    if rank == 0:
        np.save(out_path, accum.numpy())
    dist.barrier()
    dist.destroy_process_group()


def test_two_ranks_with_row_bands_make_the_single_rank_image_exactly(orc, tmp_path):
    """The other partition: every rank renders all samples of its own band of rows.  The bands
    are disjoint and the rest of each rank's buffer is zero, so the reduce adds zeros: the
    combined image is the one-rank image bit for bit, whatever the number of ranks."""
    import torch.multiprocessing as mp
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    out = str(tmp_path / "bands.npy")
    mp.spawn(_band_worker, args=(2, port, out), nprocs=2, join=True)
    got = np.load(out)
    want, _ = orc.render(orc.builtin_scene("cornell_2", 36, 25), 5, first_sample=2, seed=4, threads=2)
    assert np.array_equal(got, want) and got.sum() > 0


def test_two_ranks_reduce_to_the_single_rank_image(orc, tmp_path):
    import torch.multiprocessing as mp
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    out = str(tmp_path / "reduced.npy")
    mp.spawn(_worker, args=(2, port, out), nprocs=2, join=True)
    got = np.load(out)
    scene = orc.builtin_scene("orange_plane", 40, 24)
    want, _ = orc.render(scene, 9, first_sample=3, seed=11, threads=2)
    # same paths; only the order of the fp32 additions differs
    assert np.allclose(got, want, rtol=1e-5, atol=1e-6)
    assert got.sum() > 0
