#!/usr/bin/env python3
"""train-ctc-parallel on MI355X: the command-line contract of /root/reference/src/netbin/train-ctc-parallel.cc.

Usage: python -m eesen_amd.train_ctc_parallel [options] <feature-rspecifier> <labels-rspecifier> <model-in> [<model-out>]
e.g.:  python -m eesen_amd.train_ctc_parallel --learn-rate=4e-5 --momentum=0.9 --num-sequence=10 --frame-limit=25000 \\
           scp:feature.scp ark:labels.ark nnet.init nnet.iter1

Same options, same <Nnet> model files, same stderr contract (`TOKEN_ACCURACY >> x% <<`, grepped by
asr_egs/wsj/steps/train_ctc_parallel.sh:146,158), exit code 0 / 255 (the reference returns -1, :259-263).
Multi-GPU: one process per GPU -- either `--num-jobs=N --job-id=J` (J = 1..N, as the recipes pass them,
train_ctc_parallel_h.sh) or a launcher that exports RANK / WORLD_SIZE / LOCAL_RANK (torch.distributed.run).  Instead of
the reference's file-based model averaging every --utts-per-avg utterances (src/net/communicator.h) the ranks form an
RCCL communicator inside libeesen_hip.so (no torch in the process) and sum their FRESH gradients every minibatch, one
bucket per layer under the backward pass.  Data: as in the reference, the feature rspecifier a job receives IS that job's
shard (the recipes launch `JOB=1:$nj ... --job-id=JOB scp:feats_tr.JOB.scp`, queue.pl substitutes JOB before the process
starts, prep_scps.sh has dealt the list: train_ctc_parallel_h.sh:96,141-143) -- nothing is sharded further.  For launchers
that hand every rank the SAME command line: a JOB that stands alone is replaced by the job id here, and
--shard-shared-list=true makes rank r train minibatches r, r + N, ... of a list all ranks read
(eesen_amd.parallel.shard_minibatches).  Ranks may hold different numbers of minibatches: one that runs out keeps stepping
with a zero gradient until every rank is out of data (Net.LiveRanks; no host round trip per step).  Rank 0 writes the model
and prints the merged TOKEN_ACCURACY.
"""
from __future__ import annotations

import os
import sys
import time


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


USAGE = ("Perform one iteration of CTC training by SGD.\n"
         "The updates are done per-utterance and by processing multiple utterances in parallel.\n"
         "\n"
         "Usage: train-ctc-parallel [options] <feature-rspecifier> <labels-rspecifier> <model-in> [<model-out>]\n"
         "e.g.: \n"
         "train-ctc-parallel scp:feature.scp ark:labels.ark nnet.init nnet.iter1\n")


def build_parser():
    """The reference's ParseOptions conventions (eesen_amd/parse_options.py): --config=<file>, --print-args, --help, --verbose, `--x=y`
    before the positional arguments.  Options and help texts of train-ctc-parallel.cc:44-80 and NetTrainOptions::Register
    (src/net/train-opts.h:45-51), then this tool's own."""
    from eesen_amd.parse_options import ParseOptions
    po = ParseOptions(USAGE, prog="train-ctc-parallel")
    po.register("learn-rate", 0.008, "Learning rate")
    po.register("momentum", 0.0, "Momentum")
    po.register("adagrad-epsilon", 1e-6, "Epsilon for numerical stability for all adaptive optimizers (Adagrad, RMSProp)")
    po.register("rms-prop-rho", 0.9, "Rho parameter for RMSProp")
    po.register("binary", True, "Write model  in binary mode")
    po.register("cross-validate", False, "Perform cross-validation (no backpropagation)")
    po.register("sequence-out-file", "", "output file for the generated sequence")
    po.register("num-sequence", 5, "Number of sequences processed in parallel")
    po.register("frame-limit", 100000.0, "Max number of frames to be processed", kind="double")
    po.register("report-step", 100, "Step (number of sequences) for status reporting")
    po.register("num-jobs", 1, "Number subjobs in multi-GPU mode")
    po.register("job-id", 1, "Subjob id in multi-GPU mode")
    po.register("utts-per-avg", 500, "Number of utterances to process per average (default is 250)")
    po.register("opt-algorithm", "SGD", "Optimization algorithm (SGD|Adagrad|RMSProp)")
    po.register("device", -1, "GPU index (default: $LOCAL_RANK, else job-id - 1 with several jobs, else 0)")
    po.register("comm-addr", "", "Rendezvous address of job 1 (default $MASTER_ADDR or 127.0.0.1)")
    po.register("comm-port", 0, "Rendezvous port (default $EESEN_COMM_PORT, else $MASTER_PORT + 17)")
    po.register("comm-timeout", 300, "Seconds to wait for the other jobs at the rendezvous")
    po.register("shard-shared-list", False, "All jobs were handed the SAME feature list: job J trains minibatches J-1, J-1+N, ... of it")
    po.register("allow-identical-lists", False, "The jobs' feature rspecifiers read the same but name DIFFERENT data (node-local shards under one path): do not refuse them")
    return po


def main(argv=None) -> int:
    from eesen_amd.parse_options import ParseError
    ap = build_parser()
    try:
        o = ap.read(argv)
    except ParseError as e:      # KALDI_ERR -> main's catch: message, -1 (train-ctc-parallel.cc:259-263)
        print(str(e), file=sys.stderr)
        return 255
    n_expected = 3 if o.cross_validate else 4
    if len(o.args) != n_expected:
        ap.print_usage()
        return 1
    feature_rspecifier, targets_rspecifier, model_filename = o.args[:3]
    target_model_filename = None if o.cross_validate else o.args[3]
    try:
        from eesen_amd import kaldi_io, frontend
        from eesen_amd.api import Net, Ctc, CuMatrix, EesenError, Feeder
        from eesen_amd.batching import assemble, AssemblyStats

        from eesen_amd.api import Comm
        from eesen_amd.parallel import shard_minibatches, job_rspecifier

        if o.num_jobs > 1:                      # the reference's own way to say "I am job J of N"
            world, rank = o.num_jobs, o.job_id - 1
            if not 0 <= rank < world:
                raise EesenError(-1, "--job-id must lie in 1..--num-jobs")
        else:
            world, rank = int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0"))
        local = int(os.environ.get("LOCAL_RANK", str(rank if world > 1 else 0)))
        dev = o.device if o.device >= 0 else local
        comm = None
        if world > 1:
            port = o.comm_port or int(os.environ.get("EESEN_COMM_PORT", 0)) or int(os.environ.get("MASTER_PORT", "29500")) + 17
            comm = Comm(dev, rank, world, o.comm_addr or os.environ.get("MASTER_ADDR", "127.0.0.1"), port, o.comm_timeout)
            log(f"job {rank + 1} of {world} joined the RCCL communicator on GPU {dev}")

        net = Net(dev).Read(model_filename)
        net.SetTrainOptions(o.learn_rate, o.momentum)
        net.SetUpdateAlgorithm(o.opt_algorithm, o.adagrad_epsilon, o.rms_prop_rho)   # net.SetUpdateAlgorithm(opt), :114
        if o.cross_validate:                                                          # :116-119: dropout layers draw masks only in training
            net.SetTestMode()
        else:
            net.SetTrainMode()
            if comm is not None:
                net.SetDropoutSeed(777 + rank)     # every rank its own masks
                net.SetComm(comm)                  # per-layer gradient buckets, summed under the backward pass
        ctc = Ctc(dev)
        ctc.SetGuard(net)      # a minibatch computed from a timed-out forward pass never reaches the statistics
        if o.sequence_out_file:                                                       # :134-137
            log(f"Sequences will be written to {o.sequence_out_file} in order from feature file")
            ctc.SetSequenceOutFile(o.sequence_out_file)
        feat_dim = net.InputDim()
        targets = kaldi_io.read_vec_int_table(targets_rspecifier)
        stats = AssemblyStats()
        log(("CROSS-VALIDATION" if o.cross_validate else "TRAINING") + " STARTED")
        t0 = time.time()
        num_done, total_frames, seq_since_report = 0, 0, 0
        last = dict(obj_sum=0.0, err_tokens=0, ref_tokens=0)
        # the reader thread parses the archives; padding + interleave + H2D of batch n+1 run on the device feeder's own
        # stream while batch n trains (the reference pads on the host and copies synchronously, train-ctc-parallel.cc:186-198)
        mine = job_rspecifier(feature_rspecifier, rank, world)
        if mine != feature_rspecifier:
            log(f"feature rspecifier of job {rank + 1}: {mine}")
            if o.shard_shared_list:
                raise EesenError(-1, "--shard-shared-list with a per-job (JOB) feature list")
        feature_rspecifier = mine
        if comm is not None:
            # every job trains the WHOLE list it was given; the same list on every job would be N copies of the same gradient, silently
            # (ADVICE r3): the jobs compare their rspecifiers and refuse a shared one unless --shard-shared-list=true deals it out or
            # --allow-identical-lists=true says the same words name different data on every node.  EVERY job enters the collective and
            # the two switches travel with the hash (jobs that disagree get a message, not a hang); FNV-1a as in the native trainer, so a
            # mixed run compares like with like.
            h = 1469598103934665603
            for c in feature_rspecifier.encode():
                h = ((h ^ c) * 1099511628211) & 0xFFFFFFFFFFFFFFFF
            hv = float(h >> 16)
            sh, al = float(bool(o.shard_shared_list)), float(bool(o.allow_identical_lists))
            mx = comm.allreduce([hv, -hv, sh, -sh, al, -al], op=1)
            if mx[2] != -mx[3]:
                raise EesenError(-1, "the jobs disagree on --shard-shared-list: pass the same value to every job")
            if mx[4] != -mx[5]:
                raise EesenError(-1, "the jobs disagree on --allow-identical-lists: pass the same value to every job")
            if not o.shard_shared_list and not o.allow_identical_lists and mx[0] == -mx[1]:
                raise EesenError(-1, f"all {world} jobs were given the same feature rspecifier '{feature_rspecifier}': each job trains its "
                                     "whole list, so they would all train the same minibatches.  Hand every job its own list (feats.JOB.scp: a "
                                     "JOB that stands alone is replaced by the job id), pass --shard-shared-list=true to deal ONE shared list "
                                     "out, or --allow-identical-lists=true if the path names different data on every node")
        feeder = Feeder(dev, slots=2)
        # a feature rspecifier that is a pipe of the reference's own filters (apply-cmvn | splice-feats | subsample-feats |
        # add-deltas, train_ctc_parallel.sh:95-110): read the raw table here and run the filters on the device
        pipe = frontend.parse_feature_pipeline(feature_rspecifier) if not os.environ.get("EESEN_HOST_FEATURE_PIPES") else None
        if pipe is not None:
            feeder.set_pipeline(pipe.stages)
            table = frontend.read_raw(pipe, warn=lambda m: log(m, "WARNING"))
        else:
            table = kaldi_io.read_mat_table(feature_rspecifier)
        groups = assemble(table, targets, o.num_sequence, o.frame_limit, feat_dim, stats, interleaved=False)
        if world > 1 and o.shard_shared_list:
            groups = shard_minibatches(groups, rank, world)
        batches = _prefetch(groups)

        def stage():
            b = next(batches, None)
            return (b, feeder.submit(b.mats)) if b is not None else (None, -1)

        diff = diff_buf = None
        staged = stage()
        zero_steps = 0
        while True:
            mb, slot = staged
            if mb is None:
                # Ranks may hold different numbers of minibatches.  One that is out of data keeps stepping with a zero gradient
                # through the same collectives until NO rank had a minibatch in a step (the liveness word that rides with the
                # top layer's bucket); that closing round, which all ranks take together, leaves the model untouched.
                # Cross-validation exchanges nothing: every rank simply finishes.
                if comm is None or o.cross_validate:
                    break
                net.BackpropagateZero()
                net.Update()
                if net.LiveRanks() == 0:
                    break
                zero_steps += 1
                continue
            net.SetSeqLengths(mb.lens)
            net_out = net.Propagate(feeder.acquire(slot))
            feeder.release(slot)
            if diff_buf is None or diff_buf.rows < net_out.rows or diff_buf.cols != net_out.cols:
                net.Synchronize()                # the old matrix may still be read by queued kernels
                # (grows by half: the list is sorted by length, so T rises from minibatch to minibatch -- a new matrix and a drained
                # device per minibatch otherwise)
                diff_buf = CuMatrix(max(net_out.rows, 0 if diff_buf is None else diff_buf.rows * 3 // 2), net_out.cols, dev, zero=False)
            diff = CuMatrix.view(diff_buf.ptr, net_out.rows, net_out.cols, diff_buf.stride, dev, keepalive=diff_buf)
            # neither call waits for the device (the reference's calls return nothing and only accumulate, ctc-loss.cc:171-192)
            ctc.EvalParallel(mb.lens, net_out, mb.labels, diff, want_pzx=False)
            ctc.ErrorRateMSeq(mb.lens, net_out, mb.labels, deferred=True)
            if not o.cross_validate:
                net.Backpropagate(diff)
            staged = stage()                     # next batch: staged while the GPU runs this one's backward pass
            num_done += mb.S
            total_frames += mb.T * mb.S          # padded frames, as the reference counts them (:215)
            seq_since_report += mb.S
            if o.verbose >= 1 and seq_since_report >= o.report_step:     # ctc-loss.cc:180-192: progress since the last report
                st = ctc.stats()
                log(f"After {st['sequences']} sequences ({st['frames'] / (100.0 * 3600):g}Hr): "
                    f"Obj(log[Pzx]) = {(st['obj_sum'] - last['obj_sum']) / seq_since_report:g}"
                    f"   TokenAcc = {100.0 * (1.0 - (st['err_tokens'] - last['err_tokens']) / max(st['ref_tokens'] - last['ref_tokens'], 1)):g}%", "VLOG[1]")
                last = st; seq_since_report = 0
        for w in stats.warnings:
            log(w, "WARNING")
        if zero_steps:
            log(f"job {rank + 1} ran out of minibatches {zero_steps} step(s) before the last job")
        net.Synchronize()
        if not o.cross_validate:                                                      # :236-240
            log(net.Info())
            log(net.InfoGradient())
        if ctc.Dropped():
            log(f"{ctc.Dropped()} minibatch(es) were computed from a timed-out forward pass and are not in the statistics", "WARNING")
        if not o.cross_validate and rank == 0:
            net.Write(target_model_filename, o.binary)
        el = max(time.time() - t0, 1e-9)
        log(f"Done {num_done} files, {stats.num_no_tgt_mat} with no targets, {stats.num_other_error} with other errors. "
            f"[{'CROSS-VALIDATION' if o.cross_validate else 'TRAINING'}, {el / 60:g} min, fps{total_frames / el:g}]")
        if comm is not None:     # comm_touch_done (communicator.h:121-170): job 1 merges the jobs' Errors / Refs; ONE accuracy line per run
            st = ctc.stats()
            tot = comm.allreduce([st["err_tokens"], st["ref_tokens"]])
            if rank == 0:
                log(f"\nTOTAL TOKEN_ACCURACY >> {100.0 * (1.0 - tot[0] / max(tot[1], 1)):g}% <<")
                log(f"\nTOKEN_ACCURACY >> {100.0 * (1.0 - tot[0] / max(tot[1], 1)):g}% <<")
            net.SetComm(None)
        else:
            log(ctc.Report())
        return 0
    except Exception as e:      # train-ctc-parallel.cc:260-263
        print(str(e), file=sys.stderr)
        return 255


if __name__ == "__main__":
    sys.exit(main())
