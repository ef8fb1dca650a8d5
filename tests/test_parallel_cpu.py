"""CPU, world_size 2 over gloo: the data-parallel recipe (shard deal, all-reduce(SUM) of FRESH gradients, then
momentum / clip / update on every rank) equals one process on the whole minibatch — the parity statement of
SURVEY.md section 8(e): N ranks x S == --num-sequence = N*S.  The per-rank arithmetic here is the oracle's; the
host logic under test (eesen_amd/parallel.py) is what bench.py runs over RCCL."""
import os
import socket

import numpy as np
import pytest

from eesen_amd import synth
from eesen_amd.parallel import deal_shards, shard_batch
from tests.util import rel_err


def test_deal_is_a_balanced_partition():
    for n, w in [(32, 8), (33, 4), (5, 8), (256, 8)]:
        sh = deal_shards(n, w)
        assert sorted(i for s in sh for i in s) == list(range(n))
        assert max(map(len, sh)) - min(map(len, sh)) <= 1


def test_shard_repads_to_its_own_tmax():
    cfg = synth.config("small_bi")
    b = synth.make_batch(**cfg)
    parts = [shard_batch(b, r, 3) for r in range(3)]
    assert sum(p.S for p in parts) == b.S and sum(p.real_frames for p in parts) == b.real_frames
    for r, p in enumerate(parts):
        idx = deal_shards(b.S, 3)[r]
        assert p.T == b.lens[idx].max() and p.feats.shape == (p.T * p.S, cfg["D"])
        f_all = b.feats.reshape(b.T, b.S, -1)
        assert np.array_equal(p.feats.reshape(p.T, p.S, -1), f_all[:p.T, idx])


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _worker(rank, world, port, cfg, steps, q):
    import torch
    import torch.distributed as dist
    from oracle import net as onet
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    layers = synth.make_model(max_grad=0.05, **cfg)
    full = synth.make_batch(**cfg)
    mine = shard_batch(full, rank, world)
    net = onet.OracleNet(layers, "f32")
    net.set_train_options(1e-3, 0.9)
    lnp = 0.0
    for _ in range(steps):
        net.set_seq_lengths(mine.lens)
        out = net.propagate(mine.feats)
        c = onet.ctc_eval_parallel(out, mine.T, mine.S, mine.lens, mine.label_ids, mine.label_off, "f32")
        # backprop WITHOUT update and with the momentum term kept out: fresh gradients only
        mmt, net.momentum = net.momentum, 0.0
        saved = [[x.copy() for x in L["corr"]] for L in net.layers]
        for L in net.layers:
            for x in L["corr"]: x[...] = 0
        net.backpropagate(c["diff"], update=False)
        fresh = [[x.copy() for x in L["corr"]] for L in net.layers]
        net.momentum = mmt
        flat = torch.from_numpy(np.concatenate([g.ravel() for f in fresh for g in f]))
        dist.all_reduce(flat, op=dist.ReduceOp.SUM)            # the one exchange step
        g = flat.numpy(); i = 0
        for L, sv in zip(net.layers, saved):
            for x, s0 in zip(L["corr"], sv):
                x[...] = mmt * s0 + g[i:i + x.size].reshape(x.shape); i += x.size
        for li, L in enumerate(net.layers):
            if L["params"]: net.update_layer(li)
        t = torch.tensor([float(c["pzx"].sum())], dtype=torch.float64)
        dist.all_reduce(t)
        lnp = t.item()
    if rank == 0:
        q.put((net.get_params(), lnp))
    dist.barrier()
    dist.destroy_process_group()


def test_two_ranks_equal_one_process_on_the_whole_batch():
    import torch.multiprocessing as mp
    from oracle import net as onet
    cfg = synth.config("tiny_bi"); cfg.update(S=6, T=16)
    steps = 3
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, cfg, steps, q)) for r in range(2)]
    for p in procs: p.start()
    params_dp, lnp_dp = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60); assert p.exitcode == 0
    layers = synth.make_model(max_grad=0.05, **cfg)
    full = synth.make_batch(**cfg)
    net = onet.OracleNet(layers, "f32")
    net.set_train_options(1e-3, 0.9)
    for _ in range(steps):
        o = onet.train_step(net, full, "f32")
    assert rel_err(params_dp, net.get_params()) < 1e-5
    assert abs(lnp_dp - float(o["pzx"].sum())) / abs(float(o["pzx"].sum())) < 1e-5
