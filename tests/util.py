"""Shared helpers of the parity tests."""
import numpy as np


def rel_err(a, b, eps=1e-12):
    """||a-b||_inf / max(||b||_inf, eps): the per-tensor metric of SURVEY.md section 7 (hard part ii)."""
    a = np.asarray(a, np.float64); b = np.asarray(b, np.float64)
    return float(np.max(np.abs(a - b)) / max(float(np.max(np.abs(b))), eps)) if a.size else 0.0


def err_metrics(a, b, rel_floor=1e-3, eps=1e-300):
    """Three views of the same difference (the max-norm ratio alone is the most forgiving metric for tensors that a
    few large entries dominate):
      maxnorm  ||a-b||_inf / ||b||_inf            (rel_err above: the bar SURVEY.md section 7 names)
      l2       ||a-b||_2 / ||b||_2
      p999     the 99.9th percentile of |a-b| / |b| over the elements with |b| > rel_floor * ||b||_inf
    b is the reference."""
    a = np.asarray(a, np.float64).ravel(); b = np.asarray(b, np.float64).ravel()
    if not a.size:
        return dict(maxnorm=0.0, l2=0.0, p999=0.0)
    d = np.abs(a - b); ab = np.abs(b); mx = float(ab.max())
    sel = ab > rel_floor * mx
    return dict(maxnorm=float(d.max() / max(mx, eps)), l2=float(np.sqrt(np.square(d).sum()) / max(np.sqrt(np.square(b).sum()), eps)),
                p999=float(np.percentile(d[sel] / ab[sel], 99.9)) if sel.any() else 0.0)


def valid_mask(lens, T, S):
    """[T*S] bool: row t*S+s is a real frame of sequence s."""
    t = np.arange(T)[:, None]
    return (t < np.asarray(lens)[None, :]).reshape(T * S)


def split_params(layers, flat):
    """Split a Net::GetParams-ordered flat vector into named tensors [(layer_idx, name, array)]."""
    out, i = [], 0
    names_lstm = ["Wx", "Wm", "bias", "pi", "pf", "po"]
    for li, L in enumerate(layers):
        t, din, dout = L["type"], L["input_dim"], L["output_dim"]
        if t in ("BiLstmParallel", "LstmParallel"):
            nd = 2 if t == "BiLstmParallel" else 1
            H = dout // nd
            for d in range(nd):
                for nm, n in zip(names_lstm, [4 * H * din, 4 * H * H, 4 * H, H, H, H]):
                    out.append((li, f"{nm}_{'fw' if d == 0 else 'bw'}", flat[i:i + n])); i += n
        elif t == "AffineTransform":
            out.append((li, "W", flat[i:i + dout * din])); i += dout * din
            out.append((li, "b", flat[i:i + dout])); i += dout
    assert i == flat.size
    return out


def load_golden(name):
    """tests/golden/<name>.npz -> (cfg dict, layers with the fixture's weights, Batch, fixture arrays)."""
    import ast
    import os
    from eesen_amd import synth
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", name + ".npz"))
    cfg = ast.literal_eval(str(g["meta"]))
    layers = synth.make_model(**cfg)
    flat, i = g["params"], 0
    for L in layers:
        for k, p in enumerate(L["params"]):
            L["params"][k] = flat[i:i + p.size].reshape(p.shape).astype(np.float32); i += p.size
    assert i == flat.size
    if "dropout" in g:      # options + the masks the reference drew (oracle/make_golden.py)
        for L, d in zip([l for l in layers if l["type"].startswith("BiLstm")], ast.literal_eval(str(g["dropout"]))):
            L["dropout"] = d
    off = g["label_off"]
    labels = [g["label_ids"][off[s]:off[s + 1]].astype(np.int32) for s in range(len(off) - 1)]
    S = len(g["lens"])
    batch = synth.Batch(feats=g["feats"], lens=g["lens"].astype(np.int32), labels=labels, T=g["feats"].shape[0] // S, S=S)
    return cfg, layers, batch, g


GOLDEN = ["tiny_bi", "small_uni", "small_bi", "proj_bi", "ragged_bi"]
GOLDEN_DROPOUT = ["dropout_bi", "dropout_twiddle"]


def golden_masks(layers, g):
    """[(layer index, fwd | None, rec | None, coin)] of a dropout fixture."""
    out = []
    for li, L in enumerate(layers):
        if L.get("dropout"):
            fwd, rec = g[f"m{li}_fwd"], g[f"m{li}_rec"]
            out.append((li, fwd if fwd.size else None, rec if rec.size else None, int(g[f"m{li}_coin"])))
    return out


def diff_bound(net_out, batch, diff32, tol=1e-4):
    """Bar for comparing two fp32 evaluations of the CTC gradient `diff` = y*sum(gamma) - gamma, gamma = exp(alpha + beta - ln p - ln y):
    the exponent carries the fp32 round-off of |alpha| ~ T, so two CORRECT fp32 evaluations with a different summation order
    (the HIP bulk pass reduces the blank's positions across the wave, the reference folds them serially) differ by ~ulp(|alpha|)
    relative: 1e-4 where that floor allows, else 3x the fp32 oracle's own distance to an fp64 evaluation on the same
    probabilities.  Returns (bound, fp64 diff)."""
    from oracle import net as onet
    arb = onet.ctc_eval_parallel(net_out, batch.T, batch.S, batch.lens, batch.label_ids, batch.label_off, "f64")
    return max(tol, 3.0 * rel_err(diff32, arb["diff"])), arb["diff"]
