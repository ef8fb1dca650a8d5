"""Generates tests/golden/*.npz from THE REFERENCE ITSELF (oracle/_ref/libeesen_ref.so: the reference's
src/net + src/cpucompute compiled unmodified, plus its CUDA CTC kernel bodies run on the CPU).
TEST INFRASTRUCTURE.  Run in the authoring container (where /root/reference exists):

    python -m oracle.make_golden

Each fixture holds the inputs (so nothing depends on RNG stream stability) and the reference's outputs of one
trainer step (train-ctc-parallel.cc:195-207) with lr = 1, momentum = 0, <MaxGrad> 0, which turns the parameter
delta into the gradient: net_out, alpha, beta, pzx, diff, in_diff, params before / after,
greedy-decode error counts.  The reference holds no golden vectors of its own for this path (TESTFILES empty in
src/net/Makefile:10), so these outputs of the reference code are the pin.
"""
from __future__ import annotations

import os
import sys
import tempfile

import numpy as np

from eesen_amd import nnet_io, synth
from oracle import refbind

OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")
CASES = {
    "tiny_bi": dict(cfg="tiny_bi"),
    "small_uni": dict(cfg="small_uni"),
    "small_bi": dict(cfg="small_bi"),
    "proj_bi": dict(cfg="tiny_bi", layers=3, proj=12, H=12, T=20, S=4),
    "ragged_bi": dict(cfg="tiny_bi", T=30, S=5, min_frac=0.3, K=5, repeat_frac=0.5),
    # dropout variants: the reference draws the masks from its host RNG; they are stored with the outputs
    "dropout_bi": dict(cfg="tiny_bi", T=20, S=4, H=16, min_frac=0.5,
                       dropout=[dict(forward=0.25, fw_step=True, recurrent=0.25, rec_step=True, rnndrop=True),
                                dict(forward=0.2, fw_seq=True, recurrent=0.3, rec_seq=True, nml=True)]),
    "dropout_twiddle": dict(cfg="tiny_bi", T=16, S=3, H=8,
                            dropout=[dict(forward=0.3, fw_step=True, recurrent=0.3, rec_step=True, nml=True, twiddle=True),
                                     dict(forward=0.3, fw_step=True, recurrent=0.3, rec_seq=True, rnndrop=True, twiddle=True)]),
}


def make_case(name: str, spec: dict) -> dict:
    spec = dict(spec)
    drops = spec.pop("dropout", None)
    cfg = synth.config(spec.pop("cfg"))
    cfg.update(spec)
    layers = synth.make_model(**cfg)
    if drops:
        for L, d in zip([l for l in layers if l["type"].startswith("BiLstm")], drops):
            L["dropout"] = d
    batch = synth.make_batch(**cfg)
    path = tempfile.mktemp(suffix=".nnet")
    nnet_io.write_nnet(path, layers, binary=False)     # text, so the reference parses exactly these values
    ref = refbind.RefNet(path)
    os.unlink(path)
    before = ref.get_params()
    assert np.array_equal(before, nnet_io.flatten_params(layers)), "reference parsed different weights"
    ref.set_train_options(1.0, 0.0)
    ref.set_seq_lengths(batch.lens)
    net_out = ref.propagate(batch.feats)
    extra = {}
    if drops:
        extra["dropout"] = np.array(repr(drops))
        for li, L in enumerate(layers):
            if L.get("dropout"):
                m = ref.dropout_masks(li)
                extra[f"m{li}_fwd"] = m["fwd"]
                extra[f"m{li}_rec"] = np.hstack([m["rec_fw"], m["rec_bw"]])     # fw columns, then bw
                extra[f"m{li}_coin"] = np.array(int(m["twiddle_apply_forward"]))
    ctc = refbind.cuda_ctc_eval_parallel(net_out, batch.T, batch.S, batch.lens, batch.label_ids, batch.label_off)
    ne, nr = ref.error_rate_mseq(net_out, batch.T, batch.S, batch.lens, batch.label_ids, batch.label_off)
    in_diff = ref.backpropagate(ctc["diff"], True)
    after = ref.get_params()
    meta = {k: v for k, v in cfg.items() if isinstance(v, (int, float, str))}
    return dict(meta=np.array(repr(meta)), feats=batch.feats, lens=batch.lens, label_ids=batch.label_ids, label_off=batch.label_off,
                params=before, net_out=net_out, alpha=ctc["alpha"], beta=ctc["beta"], pzx=ctc["pzx"], diff=ctc["diff"],
                in_diff=in_diff, params_after=after, errors=np.array([ne, nr], np.int64), **extra)


def compressed_feature_case():
    """A compressed feature archive written by the reference's own CompressedMatrixWriter (`CM` for > 8 rows, `CM2` below,
    /root/reference/src/cpucompute/compressed-matrix.cc:41-110) plus what its CopyToMat decodes: pins eesen_amd.kaldi_io."""
    import ctypes as C
    lib = refbind._load()
    rng = np.random.default_rng(11)
    mats = [(f"utt{i}", (rng.standard_normal((r, 6)) * (1 + i)).astype(np.float32) + np.float32(i)) for i, r in enumerate([23, 5, 9, 8, 1, 40])]
    mats.append(("const", np.full((12, 6), 2.5, np.float32)))          # zero range: the header's percentile spacing rules kick in
    n = len(mats)
    keys = (C.c_char_p * n)(*[k.encode() for k, _ in mats])
    ptrs = (C.c_void_p * n)(*[m.ctypes.data for _, m in mats])
    rows = (C.c_int * n)(*[m.shape[0] for _, m in mats])
    decoded = np.zeros((sum(m.shape[0] for _, m in mats), 6), np.float32)
    path = os.path.join(OUT, "compressed_feats.ark")
    lib.ref_write_compressed_feats.argtypes = [C.c_char_p, C.c_int, C.POINTER(C.c_char_p), C.POINTER(C.c_void_p), C.POINTER(C.c_int), C.c_int, C.c_void_p]
    assert lib.ref_write_compressed_feats(("ark:" + path).encode(), n, keys, ptrs, rows, 6, decoded.ctypes.data_as(C.c_void_p)) == 0
    np.savez_compressed(os.path.join(OUT, "compressed_feats.npz"), decoded=decoded, rows=np.array([m.shape[0] for _, m in mats], np.int32),
                        keys=np.array([k for k, _ in mats]), original=np.concatenate([m for _, m in mats]))
    print("compressed_feats", decoded.shape, os.path.getsize(path), "bytes")


def main():
    assert refbind.build_if_possible(), "oracle/_ref could not be built (needs /root/reference)"
    os.makedirs(OUT, exist_ok=True)
    compressed_feature_case()
    if "--compressed-only" in sys.argv:
        return
    for name, spec in CASES.items():
        d = make_case(name, spec)
        np.savez_compressed(os.path.join(OUT, f"{name}.npz"), **d)
        print(name, {k: getattr(v, "shape", None) for k, v in d.items() if k in ("feats", "params", "alpha")}, "pzx", d["pzx"][:3])


if __name__ == "__main__":
    main()
