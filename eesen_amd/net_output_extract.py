#!/usr/bin/env python3
"""net-output-extract on MI355X: forward pass for decoding (/root/reference/src/netbin/net-output-extract.cc).

Usage: python -m eesen_amd.net_output_extract [options] <model-in> <feature-rspecifier> <feature-wspecifier>
e.g.:  python -m eesen_amd.net_output_extract --class-frame-counts=label.counts --apply-log=true net ark:feats.ark ark:out.ark

Per utterance: Net::Feedforward (net.cc:110-132) -> optional ApplyLog -> optional ClassPrior::SubtractOnLogpost
(class-prior.cc:30-91), written as a float-matrix table.  The reference converts <BiLstmParallel> to the single-sequence
<BiLstm> on read (layer.cc:164-170); here the same kernels run with S = 1 (or --num-sequence utterances padded together,
which changes nothing on valid frames because padding is masked in both directions).
"""
from __future__ import annotations

import os
import sys
import time

import numpy as np

FLT_MAX = float(np.finfo(np.float32).max)


def class_log_priors(counts_path: str, prior_cutoff: float = 1e-10, blank_scale: float = 1.0) -> np.ndarray:
    """ClassPrior::ClassPrior (class-prior.cc:30-77): counts -> floor -> blank scaling -> normalise -> log, with
    FLT_MAX/2 added for classes below the cutoff so that they get zero likelihood."""
    txt = open(counts_path).read().replace("[", " ").replace("]", " ")
    pri = np.array(txt.split(), dtype=np.float64)
    mask = np.zeros(pri.size, np.float32)
    low = pri < prior_cutoff
    pri[low] = prior_cutoff
    mask[low] = FLT_MAX / 2
    if blank_scale != 1.0:
        pri[0] *= blank_scale
    pri = np.log(pri / pri.sum())
    return pri.astype(np.float32) + mask


def main(argv=None) -> int:
    # the reference's ParseOptions conventions (eesen_amd/parse_options.py); options and help texts of net-output-extract.cc:39-47 and
    # ClassPriorOptions::Register (src/net/class-prior.h:46-57), then this tool's own
    from eesen_amd.parse_options import ParseOptions, ParseError
    ap = ParseOptions("Perform a forward pass through the network for classification/feature extraction.\n"
                      "\n"
                      "Usage:  net-output-extract [options] <model-in> <feature-rspecifier> <feature-wspecifier>\n"
                      "e.g.: \n"
                      "net-output-extract net ark:features.ark ark:output.ark\n", prog="net-output-extract")
    ap.register("class-frame-counts", "", "Vector with frame-counts of classes to compute log-priors. (priors are typically subtracted from "
                                          "log-posteriors or pre-softmax activations)")
    ap.register("prior-scale", 1.0, "Scaling factor to be applied on class-log-priors")
    ap.register("prior-cutoff", 1e-10, "Classes with priors lower than cutoff will have 0 likelihood")
    ap.register("blank-scale", 1.0, "Scale probability of class 0 (blank) by this factor")
    ap.register("apply-log", False, "Transform network output to logscale")
    ap.register("use-gpu", "yes", "yes|no|optional (accepted for the recipes' command lines; this tool always runs on the GPU)")
    ap.register("num-sequence", 1, "Utterances forwarded together (1 = the reference's one utterance at a time)")
    ap.register("frame-limit", 1e5, "Max number of frames forwarded together", kind="double")
    ap.register("device", 0, "GPU index")
    try:
        o = ap.read(argv)
    except ParseError as e:
        print(str(e), file=sys.stderr)
        return 255
    if len(o.args) != 3:
        ap.print_usage()
        return 1
    model_filename, feature_rspecifier, feature_wspecifier = o.args
    try:
        import ctypes as C
        from eesen_amd import kaldi_io, _lib, frontend
        from eesen_amd.api import Net, CuMatrix
        from eesen_amd.batching import interleave
        kind, out_path, text = kaldi_io._parse_specifier(feature_wspecifier)
        if kind != "ark":
            raise kaldi_io.KaldiIOError("only ark: output is supported")
        net = Net(o.device).Read(model_filename)
        net.SetTestMode()                                  # net-output-extract.cc:76
        log_pri = class_log_priors(o.class_frame_counts, o.prior_cutoff, o.blank_scale) if o.class_frame_counts else None
        K = net.OutputDim()
        if log_pri is not None and log_pri.size != K:
            raise kaldi_io.KaldiIOError(f"Dimensionality mismatch, class_frame_counts {log_pri.size} class_output_llk {K}")
        t0 = time.time()
        num_done = tot_t = 0
        # decode_ctc_lat.sh:92-95 feeds this tool `apply-cmvn ... | splice-feats ... | subsample-feats ... | add-deltas ... |`
        pipe = frontend.parse_feature_pipeline(feature_rspecifier) if not os.environ.get("EESEN_HOST_FEATURE_PIPES") else None
        feeder = None
        if pipe is not None:
            from eesen_amd.api import Feeder
            feeder = Feeder(o.device, slots=1)
            feeder.set_pipeline(pipe.stages)

        def flush(group):
            """Propagates one group and returns its (key, matrix) results: the writer streams them out as they are produced
            (the decoding scripts pipe this tool: `net-output-extract ... ark:- | latgen-faster ...`, decode_ctc_lat.sh:100)."""
            nonlocal num_done, tot_t
            results = []
            if pipe is not None:    # raw matrices: the filters of the rspecifier pipe run on the device (eesen_amd.frontend)
                for _, m in group:
                    if m.shape[1] != net.InputDim():
                        raise kaldi_io.KaldiIOError(f"feature dimension {m.shape[1]} does not match the net's InputDim {net.InputDim()}")
                lens = np.array([m.shape[0] for _, m in group], np.int32)
                T = int(lens.max())
                slot = feeder.submit([m for _, m in group])
                net.SetSeqLengths(lens)
                out = net.Propagate(feeder.acquire(slot))
                feeder.release(slot)
            else:
                feats, lens, T = interleave([m for _, m in group], net.InputDim())
                net.SetSeqLengths(lens)
                out = net.Propagate(feats)
            if o.apply_log or log_pri is not None:
                _lib.check(_lib.load().eesen_op_log_sub_prior(o.device, None, C.c_void_p(out.ptr), out.rows, out.cols, out.stride, int(o.apply_log),
                                                              log_pri.ctypes.data_as(C.c_void_p) if log_pri is not None else None, o.prior_scale))
            host = out.numpy().reshape(T, len(group), K)
            for s, (key, m) in enumerate(group):
                results.append((key, np.ascontiguousarray(host[: m.shape[0], s, :])))
                num_done += 1; tot_t += m.shape[0]
            return results

        def produce():
            group, max_len = [], 0
            table = (frontend.read_raw(pipe, warn=lambda m: print(f"WARNING (net-output-extract:main()) {m}", file=sys.stderr))
                     if pipe is not None else kaldi_io.read_mat_table(feature_rspecifier))
            for key, mat in table:
                if group and (len(group) == o.num_sequence or max(max_len, mat.shape[0]) * (len(group) + 1) > o.frame_limit):
                    yield from flush(group)
                    group, max_len = [], 0
                group.append((key, mat)); max_len = max(max_len, mat.shape[0])
            if group:
                yield from flush(group)

        kaldi_io.write_mat_ark(out_path, produce(), text=text)
        el = max(time.time() - t0, 1e-9)
        print(f"LOG (net-output-extract:main()) Done {num_done} files in {el / 60:g}min, (fps {tot_t / el:g})", file=sys.stderr)
        return 0 if num_done else 255
    except Exception as e:
        print(f"ERROR (net-output-extract:main()) {e}", file=sys.stderr)
        return 255


if __name__ == "__main__":
    sys.exit(main())
