"""Data-parallel sharding of utterance mini-batches and the gradient exchange.

The reference's multi-GPU mode is asynchronous model averaging through files
(/root/reference/src/net/communicator.h:39-119).  Here every rank runs the same Net on its own shard of
utterances and the FRESH gradients (sums over frames, /root/reference/src/net/bilstm-parallel-layer.h:504-510)
are summed with ONE all-reduce over RCCL/xGMI between backprop and update, so that N ranks x S utterances
equal the reference run with --num-sequence = N*S: corr = momentum*corr + sum_ranks(g), clip, update
(momentum and clipping are applied AFTER the sum).

Two transports for that exchange:
  * the library's own RCCL communicator (eesen_amd.api.Comm + Net.SetComm: per-layer buckets on a communication
    stream, overlapped with the lower layers' backward pass; no torch in the process) -- what bench.py and the
    trainers use;
  * `GradAllReducer`, the same sum as a `net.grad_hook` over `torch.distributed` (backend "nccl" = RCCL on ROCm,
    "gloo" in the CPU tests): one bulk all-reduce of the contiguous gradient buffer, for hosts that already live
    inside a torch process group.  torch is plumbing only: a zero-copy tensor view of the library's buffer.
"""
from __future__ import annotations

from typing import List

import numpy as np


def deal_shards(n_utts: int, world: int) -> List[List[int]]:
    """Interleaved deal of a length-sorted minibatch: utterance s goes to rank s mod N, which keeps every
    rank's T_max balanced (the reference's prep_scps.sh deals whole length-sorted batches round-robin,
    /root/reference/asr_egs/wsj/utils/prep_scps.sh:37-76)."""
    return [list(range(r, n_utts, world)) for r in range(world)]


def shard_batch(batch, rank: int, world: int):
    """The sub-batch of `rank`: its utterances re-padded to the shard's own T_max (padded frames add exactly
    zero gradient), in the trainer's time-major interleaved layout."""
    from .synth import Batch
    idx = deal_shards(batch.S, world)[rank]
    D = batch.feats.shape[1]
    f3 = batch.feats.reshape(batch.T, batch.S, D)
    lens = batch.lens[idx]
    T = int(lens.max()) if len(idx) else 0
    feats = np.ascontiguousarray(f3[:T, idx, :]).reshape(T * len(idx), D)
    return Batch(feats=feats, lens=np.ascontiguousarray(lens, np.int32), labels=[batch.labels[i] for i in idx], T=T, S=len(idx))


class _DeviceView:
    """A raw device pointer presented through __cuda_array_interface__ (zero-copy into a torch tensor)."""

    def __init__(self, ptr: int, n: int):
        self.__cuda_array_interface__ = {"shape": (n,), "typestr": "<f4", "data": (ptr, False), "version": 2, "strides": None}


def grad_tensor(net):
    """torch view of the Net's contiguous fresh-gradient buffer (no copy; memory stays owned by the Net)."""
    import torch
    ptr, n = net.grad_buffer()
    t = torch.as_tensor(_DeviceView(ptr, n), device=f"cuda:{net.device}")
    assert t.data_ptr() == ptr and t.numel() == n and t.dtype == torch.float32
    return t


class GradAllReducer:
    """`net.grad_hook`: sums the fresh gradients over all ranks, in place, between backprop and update.

    The Net must enqueue on torch's current stream (pass torch.cuda.current_stream().cuda_stream at
    creation, or leave both on the default stream): ProcessGroupNCCL orders the collective after the work
    already enqueued on the current stream and makes the current stream wait for it, so no host
    synchronisation is needed and the update kernels simply queue up behind the all-reduce."""

    def __init__(self, net, group=None):
        import torch.distributed as dist
        self.dist = dist
        self.group = group
        # a Net over the C-ABI exposes a device pointer; any other host of the recipe (the CPU tests' oracle-backed
        # stand-in) may expose its contiguous gradient buffer as a torch tensor directly
        self.t = net.grad_tensor() if hasattr(net, "grad_tensor") else grad_tensor(net)

    def __call__(self, net):
        self.dist.all_reduce(self.t, op=self.dist.ReduceOp.SUM, group=self.group)


def minibatch_owner(index: int, world: int) -> int:
    """Which rank trains minibatch `index` when all ranks read the SAME feature list: consecutive minibatches go to
    consecutive ranks, so that the N minibatches of one synchronous step are N neighbouring groups of the length-sorted
    list (similar T_max on every rank) -- the counterpart of the reference's round-robin deal of length-sorted batches to
    jobs (/root/reference/asr_egs/wsj/utils/prep_scps.sh:37-76)."""
    return index % world


def shard_minibatches(batches, rank: int, world: int):
    """Filter an iterator of minibatches down to the ones `rank` owns (see minibatch_owner)."""
    for i, b in enumerate(batches):
        if minibatch_owner(i, world) == rank:
            yield b


def job_rspecifier(rspecifier: str, rank: int, world: int = 2) -> str:
    """Kaldi's queue scripts substitute the literal JOB in per-job arguments BEFORE the process starts (`JOB=1:$nj ...
    feats_tr.JOB.scp`, /root/reference/asr_egs/wsj/steps/train_ctc_parallel_h.sh:96,141-143), so a trainer normally never
    sees it.  For launchers that hand every rank the same command line the trainers substitute it themselves -- only with
    several jobs, and only a JOB that stands alone (`feats.JOB.scp`; not `exp/JOBS/x` or `$JOBNAME`); job id = rank + 1."""
    import re
    if world <= 1:
        return rspecifier
    return re.sub(r"(?<![A-Za-z0-9_])JOB(?![A-Za-z0-9_])", str(rank + 1), rspecifier)


def allreduce_stats(values, group=None, device=None):
    """Sum of (sum ln p, #errors, #refs, #frames) over ranks: replaces the done-files of
    /root/reference/src/net/communicator.h:121-170."""
    import torch
    import torch.distributed as dist
    t = torch.tensor(list(values), dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
    return t.tolist()
