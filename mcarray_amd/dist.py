"""Sharding of independent arrays over the GPUs of one node (one process per GPU).

Arrays / streams never interact (no state is shared between module objects), so the path shards
embarrassingly: rank r owns a contiguous block of arrays, runs the whole path on them with no
collective inside the compute, and the only exchange is a gather of the per-frame output buffers
-- the DOA bin + probability of every frame (8 bytes per frame and source) to every rank, and
optionally the beamformed audio (2 KB per frame) to rank 0, which is where a consumer like the
reference's WAV writer (src/programs/mcabeamf.cpp:112-118) sits.  `torch.distributed`: RCCL over
xGMI with backend "nccl" on the GPUs, gloo on CPU in the tests; the same code runs on both.
"""
import os
import subprocess
import sys

import torch
import torch.distributed as dist


def partition(n_units, world):
    """Contiguous block sizes, first n_units % world ranks get one more."""
    base, rem = divmod(n_units, world)
    return [base + (1 if r < rem else 0) for r in range(world)]


def local_range(n_units, rank, world):
    sizes = partition(n_units, world)
    start = sum(sizes[:rank])
    return range(start, start + sizes[rank])


def array_seed(base_seed, global_array_index):
    """Synthetic-input seed of an array depends on its GLOBAL index only, so the data a given array
    sees does not depend on how many ranks the job has."""
    return int(base_seed) + int(global_array_index)


def gather_arrays(local, n_units, group=None):
    """local: [A_local, ...] result block of this rank -> [n_units, ...] on every rank, in array order.
    Handles ragged blocks by padding to the largest block."""
    if not dist.is_available() or not dist.is_initialized() or dist.get_world_size(group) == 1:
        return local
    world = dist.get_world_size(group)
    sizes = partition(n_units, world)
    amax = max(sizes)
    if local.shape[0] != sizes[dist.get_rank(group)]:
        raise ValueError("local block has %d arrays, partition says %d" % (local.shape[0], sizes[dist.get_rank(group)]))
    if local.shape[0] < amax:
        pad = torch.zeros((amax - local.shape[0],) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
        local = torch.cat([local, pad], dim=0)
    local = local.contiguous()
    parts = [torch.empty_like(local) for _ in range(world)]
    dist.all_gather(parts, local, group=group)
    return torch.cat([p[:s] for p, s in zip(parts, sizes)], dim=0)


class StepGather:
    """The per-step exchange of a sharded job, double buffered so that step i + 1 computes while the gather of step i
    is in flight.

    Every rank owns `n_local` arrays (equal blocks: weak scaling).  The DOA bins (int32) and the probabilities (fp32) of
    a step share one 4-byte-word buffer [2][n_local][F][S] (two contiguous halves: what the C ABI writes), so the
    exchange is ONE all_gather_into_tensor per step, into [world][2][n_local][F][S] on every rank.  With
    `audio_samples` > 0 the beamformed audio [n_local][S][audio_samples] is gathered too, to rank 0 only.

        b = g.begin_step()                 # waits for the gather that last used this buffer pair (two steps ago)
        bins, prob = g.doa_buffers(b)      # views the compute writes into (and g.audio_buffer(b))
        ...enqueue the compute...
        g.end_step(b)                      # asynchronous gather of buffer pair b
        g.drain()                          # before reading g.gathered_doa(b) / g.gathered_audio(b)
    """

    def __init__(self, n_local, n_frames, n_sources, device, group=None, audio_samples=0):
        self.group = group
        self.on = dist.is_available() and dist.is_initialized()
        self.world = dist.get_world_size(group) if self.on else 1
        self.rank = dist.get_rank(group) if self.on else 0
        self.n_local, self.F, self.S = n_local, n_frames, n_sources
        shape = (2, n_local, n_frames, n_sources)
        self.packed = [torch.empty(shape, dtype=torch.int32, device=device) for _ in range(2)]
        self.all_doa = [torch.empty((self.world * 2,) + shape[1:], dtype=torch.int32, device=device) for _ in range(2)] if self.on else None
        self.audio = self.all_audio = None
        if audio_samples > 0:
            self.audio = [torch.empty(n_local, n_sources, audio_samples, dtype=torch.float32, device=device) for _ in range(2)]
            if self.on and self.rank == 0:
                self.all_audio = [[torch.empty_like(self.audio[0]) for _ in range(self.world)] for _ in range(2)]
        self.pending = [[], []]
        self.step = 0

    def begin_step(self):
        b = self.step & 1
        self.step += 1
        self._wait(b)
        return b

    def doa_buffers(self, b):
        """(doa_bin int32 [n_local][F][S], prob fp32 [n_local][F][S]): the two contiguous halves of the packed buffer."""
        return self.packed[b][0], self.packed[b][1].view(torch.float32)

    def audio_buffer(self, b):
        return self.audio[b] if self.audio is not None else None

    def end_step(self, b):
        if not self.on:
            return
        self.pending[b].append(dist.all_gather_into_tensor(self.all_doa[b], self.packed[b], group=self.group, async_op=True))
        if self.audio is not None:
            dst = dist.get_global_rank(self.group, 0) if self.group is not None else 0
            self.pending[b].append(dist.gather(self.audio[b], self.all_audio[b] if self.rank == 0 else None, dst=dst,
                                               group=self.group, async_op=True))

    def _wait(self, b):
        for w in self.pending[b]:
            w.wait()
        self.pending[b] = []

    def drain(self):
        self._wait(0)
        self._wait(1)

    def gathered_doa(self, b):
        """(doa_bin, prob) of ALL arrays of the job, [world * n_local][F][S] each, in global array order."""
        src = (self.all_doa[b] if self.on else self.packed[b]).view(self.world, 2, self.n_local, self.F, self.S)
        n = self.world * self.n_local
        return src[:, 0].reshape(n, self.F, self.S), src[:, 1].reshape(n, self.F, self.S).view(torch.float32)

    def gathered_audio(self, b):
        """rank 0: [world * n_local][S][samples]; other ranks: None."""
        if self.audio is None:
            return None
        if not self.on:
            return self.audio[b]
        return torch.cat(self.all_audio[b], dim=0) if self.rank == 0 else None


def launch_ranks(n, argv, master_port=None):
    """Runs `python argv...` as n rank processes on this node (one per GPU: RANK = LOCAL_RANK = 0..n-1, rendezvous on
    127.0.0.1) and returns the worst exit code.  The caller must not have touched the GPU: the children are plain
    subprocesses of a parent that only waits (never an exec from a process that initialised HIP)."""
    if master_port is None:
        import socket
        s = socket.socket()
        s.bind(("127.0.0.1", 0))
        master_port = s.getsockname()[1]
        s.close()
    procs = []
    for r in range(n):
        env = dict(os.environ)
        env.update(RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(master_port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable] + list(argv), env=env))
    # poll: a rank that dies before the rendezvous would leave the others waiting for the collective's timeout
    import time
    codes = [None] * n
    while any(c is None for c in codes):
        for i, pr in enumerate(procs):
            if codes[i] is None:
                codes[i] = pr.poll()
        failed = [c for c in codes if c not in (None, 0)]
        if failed:
            for i, pr in enumerate(procs):
                if codes[i] is None:
                    pr.terminate()
            for i, pr in enumerate(procs):
                if codes[i] is None:
                    try:
                        codes[i] = pr.wait(timeout=10)
                    except subprocess.TimeoutExpired:
                        pr.kill()
                        codes[i] = pr.wait()
            break
        time.sleep(0.05)
    return max((abs(c) for c in codes), default=0)
