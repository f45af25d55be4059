"""CPU, world_size 2, gloo: the N>1 path's sharding and DOA gather (mcarray_amd/dist.py) reproduce
the unsharded result.  The per-array compute is replaced by a deterministic stand-in (the real one
needs a GPU; it is per-array independent, which is exactly what this test relies on)."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from mcarray_amd import dist as mdist


def _fake_localise(global_index, frames):
    g = torch.Generator().manual_seed(mdist.array_seed(1000, global_index))
    e = torch.rand(frames, 37, generator=g)
    return e.argmax(dim=1).to(torch.int32).unsqueeze(-1), e.max(dim=1).values.unsqueeze(-1)


def _worker(rank, world, port, n_arrays, frames, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    mine = mdist.local_range(n_arrays, rank, world)
    bins = torch.stack([_fake_localise(g, frames)[0] for g in mine]) if len(mine) else torch.empty(0, frames, 1, dtype=torch.int32)
    prob = torch.stack([_fake_localise(g, frames)[1] for g in mine]) if len(mine) else torch.empty(0, frames, 1)
    all_bins = mdist.gather_arrays(bins, n_arrays)
    all_prob = mdist.gather_arrays(prob, n_arrays)
    torch.save((all_bins, all_prob), os.path.join(out_dir, "r%d.pt" % rank))
    dist.barrier()
    dist.destroy_process_group()


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def test_partition_covers_every_array_once():
    for n in (1, 7, 8, 1024, 1025):
        for w in (1, 2, 3, 8):
            idx = [i for r in range(w) for i in mdist.local_range(n, r, w)]
            assert idx == list(range(n))
    assert mdist.partition(1024, 8) == [128] * 8          # BASELINE configs[4]: 128 arrays per GPU


@pytest.mark.parametrize("n_arrays", [6, 5])
def test_two_rank_gather_matches_unsharded(tmp_path, n_arrays):
    frames, world = 9, 2
    mp.spawn(_worker, args=(world, _free_port(), n_arrays, frames, str(tmp_path)), nprocs=world, join=True)
    ref_bins = torch.stack([_fake_localise(g, frames)[0] for g in range(n_arrays)])
    ref_prob = torch.stack([_fake_localise(g, frames)[1] for g in range(n_arrays)])
    for r in range(world):
        b, p = torch.load(os.path.join(str(tmp_path), "r%d.pt" % r))
        assert torch.equal(b, ref_bins) and torch.equal(p, ref_prob)
