#!/usr/bin/env python3
"""train-ctc-parallel on MI355X: the command-line contract of /root/reference/src/netbin/train-ctc-parallel.cc.

Usage: python -m eesen_amd.train_ctc_parallel [options] <feature-rspecifier> <labels-rspecifier> <model-in> [<model-out>]
e.g.:  python -m eesen_amd.train_ctc_parallel --learn-rate=4e-5 --momentum=0.9 --num-sequence=10 --frame-limit=25000 \\
           scp:feature.scp ark:labels.ark nnet.init nnet.iter1

Same options, same <Nnet> model files, same stderr contract (`TOKEN_ACCURACY >> x% <<`, grepped by
asr_egs/wsj/steps/train_ctc_parallel.sh:146,158), exit code 0 / 255 (the reference returns -1, :259-263).
Multi-GPU: instead of --num-jobs/--job-id file averaging (src/net/communicator.h), launch one process per GPU with
torch.distributed.run; every rank reads ITS OWN feature list (as the reference's per-job scp shards) and the fresh
gradients are all-reduced over RCCL every minibatch (eesen_amd/parallel.py).
"""
from __future__ import annotations

import argparse
import os
import sys
import time


def _bool(v: str) -> bool:
    if v.lower() in ("true", "t", "1", "yes"):
        return True
    if v.lower() in ("false", "f", "0", "no"):
        return False
    raise argparse.ArgumentTypeError(f"bad boolean '{v}'")


def log(msg: str, level: str = "LOG"):
    print(f"{level} (train-ctc-parallel:main():eesen_amd/train_ctc_parallel.py) {msg}", file=sys.stderr, flush=True)


def _prefetch(it, depth: int = 2):
    """Reads and assembles the next minibatches on a host thread while the GPU works on the current one (the reference
    reads, pads and uploads synchronously between steps, train-ctc-parallel.cc:149-198)."""
    import queue
    import threading
    q: "queue.Queue" = queue.Queue(maxsize=depth)
    END = object()

    def run():
        try:
            for x in it:
                q.put(x)
            q.put(END)
        except BaseException as e:   # surface reader errors in the consumer
            q.put(e)

    threading.Thread(target=run, daemon=True).start()
    while True:
        x = q.get()
        if x is END:
            return
        if isinstance(x, BaseException):
            raise x
        yield x


def build_parser() -> argparse.ArgumentParser:
    ap = argparse.ArgumentParser(prog="train-ctc-parallel", add_help=True,
                                 description="Perform one iteration of CTC training by SGD; multiple utterances are processed in parallel.")
    # NetTrainOptions (src/net/train-opts.h:45-51)
    ap.add_argument("--learn-rate", type=float, default=0.008)
    ap.add_argument("--momentum", type=float, default=0.0)
    ap.add_argument("--adagrad-epsilon", type=float, default=1e-6)
    ap.add_argument("--rms-prop-rho", type=float, default=0.9)
    # train-ctc-parallel.cc:48-80
    ap.add_argument("--binary", type=_bool, default=True, help="Write model in binary mode")
    ap.add_argument("--cross-validate", type=_bool, default=False, help="Perform cross-validation (no backpropagation)")
    ap.add_argument("--sequence-out-file", default="")
    ap.add_argument("--num-sequence", type=int, default=5, help="Number of sequences processed in parallel")
    ap.add_argument("--frame-limit", type=float, default=100000, help="Max number of frames to be processed")
    ap.add_argument("--report-step", type=int, default=100, help="Step (number of sequences) for status reporting")
    ap.add_argument("--num-jobs", type=int, default=1)
    ap.add_argument("--job-id", type=int, default=1)
    ap.add_argument("--utts-per-avg", type=int, default=500)
    ap.add_argument("--opt-algorithm", default="SGD", help="Optimization algorithm (SGD|Adagrad|RMSProp)")
    ap.add_argument("--verbose", type=int, default=0)
    ap.add_argument("--device", type=int, default=None, help="GPU index (default: LOCAL_RANK or 0)")
    ap.add_argument("args", nargs="*")
    return ap


def main(argv=None) -> int:
    ap = build_parser()
    o = ap.parse_args(argv)
    n_expected = 3 if o.cross_validate else 4
    if len(o.args) != n_expected:
        ap.print_usage(sys.stderr)
        return 1
    feature_rspecifier, targets_rspecifier, model_filename = o.args[:3]
    target_model_filename = None if o.cross_validate else o.args[3]
    try:
        from eesen_amd import kaldi_io
        from eesen_amd.api import Net, Ctc, CuMatrix, EesenError, Feeder
        from eesen_amd.batching import assemble, AssemblyStats

        world = int(os.environ.get("WORLD_SIZE", "1"))
        rank = int(os.environ.get("RANK", "0"))
        local = int(os.environ.get("LOCAL_RANK", "0"))
        if o.num_jobs != 1 and world == 1:
            raise EesenError(-1, "--num-jobs > 1: file-based model averaging is replaced by the RCCL gradient all-reduce; "
                                 "launch one process per GPU with `python -m torch.distributed.run --nproc-per-node N`")
        if o.sequence_out_file:
            raise EesenError(-1, "--sequence-out-file is not supported")
        dist = None
        if world > 1:
            import torch
            import torch.distributed as dist
            torch.cuda.set_device(local)
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
        dev = o.device if o.device is not None else (local if world > 1 else 0)

        net = Net(dev).Read(model_filename)
        net.SetTrainOptions(o.learn_rate, o.momentum)
        net.SetUpdateAlgorithm(o.opt_algorithm, o.adagrad_epsilon, o.rms_prop_rho)   # net.SetUpdateAlgorithm(opt), :114
        if o.cross_validate:                                                          # :116-119: dropout layers draw masks only in training
            net.SetTestMode()
        else:
            net.SetTrainMode()
            if dist is not None:      # every rank its own masks
                net.SetDropoutSeed(777 + rank)
        if dist is not None and not o.cross_validate:
            from eesen_amd.parallel import GradAllReducer
            net.grad_hook = GradAllReducer(net)
        ctc = Ctc(dev)
        feat_dim = net.InputDim()
        targets = kaldi_io.read_vec_int_table(targets_rspecifier)
        stats = AssemblyStats()
        log(("CROSS-VALIDATION" if o.cross_validate else "TRAINING") + " STARTED")
        t0 = time.time()
        num_done, total_frames, seq_since_report = 0, 0, 0
        obj_prog = err_prog = ref_prog = 0.0
        # the reader thread parses the archives; padding + interleave + H2D of batch n+1 run on the device feeder's own
        # stream while batch n trains (the reference pads on the host and copies synchronously, train-ctc-parallel.cc:186-198)
        batches = _prefetch(assemble(kaldi_io.read_mat_table(feature_rspecifier), targets, o.num_sequence, o.frame_limit, feat_dim, stats,
                                     interleaved=False))
        feeder = Feeder(dev, slots=2)

        def stage():
            b = next(batches, None)
            return (b, feeder.submit(b.mats)) if b is not None else (None, -1)

        diff = None
        staged = stage()
        while True:
            mb, slot = staged
            if dist is not None:     # ranks may hold different numbers of minibatches: keep stepping until all are done
                import torch
                flag = torch.tensor([1.0 if mb is not None else 0.0], device=f"cuda:{local}")
                dist.all_reduce(flag)
                if flag.item() == 0:
                    break
            elif mb is None:
                break
            if mb is not None:
                net.SetSeqLengths(mb.lens)
                net_out = net.Propagate(feeder.acquire(slot))
                feeder.release(slot)
                if diff is None or diff.rows != net_out.rows:
                    diff = CuMatrix(net_out.rows, net_out.cols, dev, zero=False)
                ctc.EvalParallel(mb.lens, net_out, mb.labels, diff)
                ne, nr = ctc.ErrorRateMSeq(mb.lens, net_out, mb.labels)
                if not o.cross_validate:
                    net.Backpropagate(diff)
                staged = stage()                     # next batch: staged while the GPU runs this one's backward pass
                num_done += mb.S
                total_frames += mb.T * mb.S          # padded frames, as the reference counts them (:215)
                obj_prog += float(ctc.pzx.sum()); err_prog += ne; ref_prog += nr; seq_since_report += mb.S
                if o.verbose >= 1 and seq_since_report >= o.report_step:     # ctc-loss.cc:180-192
                    st = ctc.stats()
                    log(f"After {st['sequences']} sequences ({st['frames'] / (100.0 * 3600):g}Hr): Obj(log[Pzx]) = {obj_prog / seq_since_report:g}"
                        f"   TokenAcc = {100.0 * (1.0 - err_prog / max(ref_prog, 1)):g}%", "VLOG[1]")
                    obj_prog = err_prog = ref_prog = 0.0; seq_since_report = 0
            elif not o.cross_validate:
                # this rank is out of data but others are not: contribute a zero gradient to the collective
                import torch
                from eesen_amd.parallel import grad_tensor
                grad_tensor(net).zero_()
                net.grad_hook(net)
                net.Update()
                staged = stage()
        for w in stats.warnings:
            log(w, "WARNING")
        net.Synchronize()
        if not o.cross_validate and rank == 0:
            net.Write(target_model_filename, o.binary)
        el = max(time.time() - t0, 1e-9)
        log(f"Done {num_done} files, {stats.num_no_tgt_mat} with no targets, 0 with other errors. "
            f"[{'CROSS-VALIDATION' if o.cross_validate else 'TRAINING'}, {el / 60:g} min, fps{total_frames / el:g}]")
        if dist is not None:
            from eesen_amd.parallel import allreduce_stats
            st = ctc.stats()
            tot = allreduce_stats([st["err_tokens"], st["ref_tokens"]], device=f"cuda:{local}")
            if rank == 0:
                log(f"\nTOTAL TOKEN_ACCURACY >> {100.0 * (1.0 - tot[0] / max(tot[1], 1)):g}% <<")
            dist.destroy_process_group()
        log(ctc.Report())
        return 0
    except Exception as e:      # train-ctc-parallel.cc:260-263
        print(str(e), file=sys.stderr)
        return 255


if __name__ == "__main__":
    sys.exit(main())
