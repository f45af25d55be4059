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


def _step_worker(rank, world, port, n_local, frames, steps, out_dir):
    """drives mdist.StepGather -- the exchange bench.py runs per step (packed DOA all_gather, audio gather to rank 0)"""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    g = mdist.StepGather(n_local, frames, 1, torch.device("cpu"), audio_samples=frames * 4)
    seen = []
    for i in range(steps):
        b = g.begin_step()
        bins, prob = g.doa_buffers(b)
        assert bins.is_contiguous() and prob.is_contiguous()          # the C ABI writes [arrays][frames][S] blocks
        for a in range(n_local):
            gi = rank * n_local + a
            fb, fp = _fake_localise(gi + 100 * i, frames)
            bins[a].copy_(fb)
            prob[a].copy_(fp)
            g.audio_buffer(b)[a].fill_(float(gi + 100 * i))
        g.end_step(b)
        if i >= 1:          # the previous step's gather may still be in flight while this one was "computed": now read it
            pb = b ^ 1
            g._wait(pb)
            ab, ap = g.gathered_doa(pb)
            au = g.gathered_audio(pb)
            seen.append((i - 1, ab.clone(), ap.clone(), None if au is None else au[:, 0, 0].clone()))
    g.drain()
    b = (steps - 1) & 1
    ab, ap = g.gathered_doa(b)
    au = g.gathered_audio(b)
    seen.append((steps - 1, ab.clone(), ap.clone(), None if au is None else au[:, 0, 0].clone()))
    torch.save(seen, os.path.join(out_dir, "s%d.pt" % rank))
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


def test_two_rank_step_gather_double_buffered(tmp_path):
    """mdist.StepGather with 2 ranks: every step's packed DOA buffers reach every rank in global array order, the audio
    reaches rank 0, and the two buffer pairs do not get mixed up over 5 steps."""
    n_local, frames, steps, world = 3, 7, 5, 2
    mp.spawn(_step_worker, args=(world, _free_port(), n_local, frames, steps, str(tmp_path)), nprocs=world, join=True)
    for r in range(world):
        seen = torch.load(os.path.join(str(tmp_path), "s%d.pt" % r))
        assert [s[0] for s in seen] == list(range(steps))
        for i, ab, ap, au in seen:
            ref_b = torch.stack([_fake_localise(g + 100 * i, frames)[0] for g in range(world * n_local)])
            ref_p = torch.stack([_fake_localise(g + 100 * i, frames)[1] for g in range(world * n_local)])
            assert torch.equal(ab, ref_b) and torch.equal(ap, ref_p)
            if r == 0:
                assert au.tolist() == [float(g + 100 * i) for g in range(world * n_local)]
            else:
                assert au is None


def test_step_gather_without_process_group():
    g = mdist.StepGather(2, 4, 1, torch.device("cpu"), audio_samples=8)
    b = g.begin_step()
    bins, prob = g.doa_buffers(b)
    bins.fill_(3)
    prob.fill_(0.5)
    g.end_step(b)
    g.drain()
    ab, ap = g.gathered_doa(b)
    assert ab.shape == (2, 4, 1) and int(ab.sum()) == 24 and float(ap.sum()) == 4.0
    assert g.gathered_audio(b).shape == (2, 1, 8)
