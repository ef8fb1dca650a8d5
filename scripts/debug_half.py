#!/usr/bin/env python3
"""Debug: one training step of a small net in GEMM modes 1 and 2; where do the modes part?"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from eesen_amd import synth, _lib
from eesen_amd.api import Net, Ctc, CuMatrix
from tests.util import rel_err, split_params

lib = _lib.load()
for name, over in (("cfg2", dict(T=24, layers=1, H=128)), ("small_bi", {}), ("cfg2", dict(T=20, layers=1, S=30)), ("cfg2", dict(T=24, layers=2, H=256))):
    cfg = synth.config(name); cfg.update(over)
    layers = synth.make_model(**cfg); batch = synth.make_batch(**cfg)
    res = {}
    for mode in (1, 2):
        lib.eesen_set_gemm_mode(mode)
        net = Net.from_layers(layers); net.SetTrainOptions(1.0, 0.0); ctc = Ctc()
        runs = []
        for it in range(2):
            net.SetSeqLengths(batch.lens)
            out = net.Propagate(batch.feats)
            diff = ctc.EvalParallel(batch.lens, out, batch.labels)
            idf = CuMatrix(batch.T * batch.S, cfg["D"])
            net.BackpropagateNoUpdate(diff, idf)
            runs.append((out.numpy(), diff.numpy(), idf.numpy(), net.GetGrads()))
        res[mode] = runs
    lib.eesen_set_gemm_mode(-1)
    print(name, over, {k: cfg[k] for k in ("H", "S", "T", "D", "layers")})
    a, b = res[1][0], res[2][0]
    print("  mode2 vs mode1: net_out %.2e diff %.2e in_diff %.2e grads %.2e | mode2 run-to-run in_diff equal: %s grads equal: %s" % (
        rel_err(b[0], a[0]), rel_err(b[1], a[1]), rel_err(b[2], a[2]), rel_err(b[3], a[3]),
        np.array_equal(res[2][0][2], res[2][1][2]), np.array_equal(res[2][0][3], res[2][1][3])))
    print("  in_diff mode2: nan %d, zero rows %d of %d; max %.3e (mode1 %.3e)" % (np.isnan(b[2]).sum(), (np.abs(b[2]).max(1) == 0).sum(), b[2].shape[0], np.nanmax(np.abs(b[2])), np.abs(a[2]).max()))
    for (li, nm, x), (_, _, y) in zip(split_params(layers, b[3]), split_params(layers, a[3])):
        e = rel_err(x, y)
        if e > 1e-5: print("   L%d %-8s %.2e  max2 %.3e max1 %.3e nan %d" % (li, nm, e, np.nanmax(np.abs(x)), np.abs(y).max(), np.isnan(x).sum()))
