// train_ctc_parallel.cc -- the reference's trainer binary (src/netbin/train-ctc-parallel.cc) over the C-ABI of include/eesen_hip.h.
//
// Host C++ only: no HIP headers, no torch, no Python.  Same options, positional arguments, stderr protocol and exit codes
// as the reference binary.  What differs from the reference's loop (:144-215) is only where the work happens: minibatch
// padding + interleave + upload run on the device feeder's own stream under the previous step (eesen_feeder_*), and
// Propagate / CTC / Backpropagate are the HIP path.
//
// Multi-GPU: --num-jobs=N --job-id=J (J = 1..N, as the recipes pass them: train_ctc_parallel_h.sh `JOB=1:$nj`) start one
// process per GPU.  Instead of the reference's file-based model averaging every --utts-per-avg utterances (:208-235,
// src/net/communicator.h) the jobs form an RCCL communicator (rank = J-1; job 1 hands out the id over TCP on
// --comm-addr / --comm-port, default $MASTER_ADDR and $EESEN_COMM_PORT | $MASTER_PORT+17 | 29517) and sum their
// gradients every minibatch (eesen_net_set_comm: per-layer buckets under the backward pass).  Data: as in the reference,
// the feature rspecifier a job receives IS that job's shard -- the recipes launch `JOB=1:$nj ... --job-id=JOB
// scp:feats_tr.JOB.scp`, queue.pl / run.pl substitute JOB before the process starts, and prep_scps.sh has dealt the list
// (train_ctc_parallel_h.sh:96,141-143) -- so nothing is sharded further.  For launchers that hand every rank the SAME
// command line: a `JOB` that stands alone (not part of a longer word) is replaced by the job id here, and
// --shard-shared-list=true makes job J train minibatches J-1, J-1+N, ... of a list all jobs read.  Jobs may hold different
// numbers of minibatches: one that runs out keeps stepping with a zero gradient until every job is out of data (the
// liveness word of eesen_net_live_ranks; no host round trip per step).  Job 1 writes the model and prints the merged
// TOKEN_ACCURACY.
//
// Tables: `ark:file`, `ark,t:file`, `scp:file` for the features (float matrices: binary FM, text, compressed CM / CM2 --
// src/cpucompute/matrix.cc:968-994, compressed-matrix.cc:437-520) and the labels (int32 vectors, binary or text --
// src/util/kaldi-holder-inl.h:190-260).
#include <cctype>
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <iostream>
#include <map>
#include <memory>
#include <sstream>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../../include/eesen_hip.h"
#include "../../../include/eesen_hip_info.h"
#include "kaldi_tables.h"
#include "feat_pipeline.h"
#include "parse_options.h"

namespace {
using namespace ktab;

void ck(int rc) {  // KALDI_ERR: message + std::runtime_error (src/base/kaldi-error.cc:168-182)
  if (rc != EESEN_OK) throw std::runtime_error(eesen_last_error());
}
void log_line(const char* level, const std::string& msg) {
  std::cerr << level << " (train-ctc-parallel:main():eesen_amd/csrc/tools/train_ctc_parallel.cc) " << msg << std::endl;
}
std::string fmt_g(double v) {  // what operator<< prints for a float/double by default
  std::ostringstream o;
  o << v;
  return o.str();
}

// ---------------------------------------------------------------------------------------------------- options
struct Options {  // train-ctc-parallel.cc:45-80, NetTrainOptions train-opts.h:29-62
  float learn_rate = 0.008f, momentum = 0.f, adagrad_epsilon = 1e-6f, rms_prop_rho = 0.9f;
  bool binary = true, cross_validate = false, shard_shared_list = false, allow_identical_lists = false;
  int num_sequence = 5, report_step = 100, num_jobs = 1, job_id = 1, utts_per_avg = 500, verbose = 0, device = -1;
  int comm_port = 0, comm_timeout = 300;
  double frame_limit = 100000;
  std::string opt_algorithm = "SGD", sequence_out_file, comm_addr;
  std::vector<std::string> args;
};
const char* kUsage =   // train-ctc-parallel.cc:36-42
    "Perform one iteration of CTC training by SGD.\n"
    "The updates are done per-utterance and by processing multiple utterances in parallel.\n"
    "\n"
    "Usage: train-ctc-parallel [options] <feature-rspecifier> <labels-rspecifier> <model-in> [<model-out>]\n"
    "e.g.: \n"
    "train-ctc-parallel scp:feature.scp ark:labels.ark nnet.init nnet.iter1\n";

// The reference's ParseOptions conventions (tools/parse_options.h): --config=<file>, --print-args, --help, --verbose, `--x=y` before
// the positional arguments, "Invalid option" + usage + exit code -1 for anything else.  Options and help texts of
// train-ctc-parallel.cc:44-80 and NetTrainOptions::Register (train-opts.h:45-51), then this tool's own.
Options parse_options(int argc, char** argv, eesen_tools::ParseOptions* po) {
  Options o;
  po->Register("learn-rate", &o.learn_rate, "Learning rate");
  po->Register("momentum", &o.momentum, "Momentum");
  po->Register("adagrad-epsilon", &o.adagrad_epsilon, "Epsilon for numerical stability for all adaptive optimizers (Adagrad, RMSProp)");
  po->Register("rms-prop-rho", &o.rms_prop_rho, "Rho parameter for RMSProp");
  po->Register("binary", &o.binary, "Write model  in binary mode");
  po->Register("cross-validate", &o.cross_validate, "Perform cross-validation (no backpropagation)");
  po->Register("sequence-out-file", &o.sequence_out_file, "output file for the generated sequence");
  po->Register("num-sequence", &o.num_sequence, "Number of sequences processed in parallel");
  po->Register("frame-limit", &o.frame_limit, "Max number of frames to be processed");
  po->Register("report-step", &o.report_step, "Step (number of sequences) for status reporting");
  po->Register("num-jobs", &o.num_jobs, "Number subjobs in multi-GPU mode");
  po->Register("job-id", &o.job_id, "Subjob id in multi-GPU mode");
  po->Register("utts-per-avg", &o.utts_per_avg, "Number of utterances to process per average (default is 250)");
  po->Register("opt-algorithm", &o.opt_algorithm, "Optimization algorithm (SGD|Adagrad|RMSProp)");
  po->Register("device", &o.device, "GPU index (default: $LOCAL_RANK, else job-id - 1 with several jobs, else 0)");
  po->Register("comm-addr", &o.comm_addr, "Rendezvous address of job 1 (default $MASTER_ADDR or 127.0.0.1)");
  po->Register("comm-port", &o.comm_port, "Rendezvous port (default $EESEN_COMM_PORT, else $MASTER_PORT + 17)");
  po->Register("comm-timeout", &o.comm_timeout, "Seconds to wait for the other jobs at the rendezvous");
  po->Register("shard-shared-list", &o.shard_shared_list, "All jobs were handed the SAME feature list: job J trains minibatches J-1, J-1+N, ... of it");
  po->Register("allow-identical-lists", &o.allow_identical_lists, "The jobs' feature rspecifiers read the same but name DIFFERENT data (node-local shards under one path): do not refuse them");
  po->Read(argc, argv);
  o.verbose = po->Verbose();
  for (int i = 1; i <= po->NumArgs(); ++i) o.args.push_back(po->GetArg(i));
  return o;
}

struct Minibatch {
  std::vector<Mat> mats;
  std::vector<std::vector<int32_t>> labels;
  std::vector<int> frames;              // what the net and the CTC see: frames BEHIND the feature pipeline
  std::vector<int> raw_frames;          // rows of the matrices handed to the feeder
  std::vector<const float*> cmvn;       // per utterance [2 x D_raw] offsets / scales (device feature front end with a CMVN stage)
  int T = 0;
};

}  // namespace

int main(int argc, char** argv) {
  try {
    setenv("HSA_ENABLE_IPC_MODE_LEGACY", "0", /*overwrite*/ 0);   // dmabuf IPC for RCCL peer access; read when the ROCr runtime initialises
    eesen_tools::ParseOptions po(kUsage);
    const Options o = parse_options(argc, argv, &po);
    if ((int)o.args.size() != 4 - (o.cross_validate ? 1 : 0)) {  // :82-85
      po.PrintUsage();
      return 1;
    }
    if (o.num_jobs < 1 || o.job_id < 1 || o.job_id > o.num_jobs) throw std::runtime_error("--job-id must lie in 1..--num-jobs");
    const int world = o.num_jobs, rank = o.job_id - 1;
    std::string feature_rspecifier = o.args[0];
    const std::string targets_rspecifier = o.args[1], model_filename = o.args[2];
    const std::string target_model_filename = o.cross_validate ? "" : o.args[3];
    // The list a job receives is its own shard (reference semantics).  A launcher that cannot substitute JOB itself may leave
    // it to us -- only with several jobs, and only a JOB that stands alone (`feats.JOB.scp`, not `exp/JOBS/` or `$JOBNAME`).
    if (world > 1) {
      auto word = [](char c) { return std::isalnum((unsigned char)c) || c == '_'; };
      bool rewritten = false;
      for (size_t at = 0; (at = feature_rspecifier.find("JOB", at)) != std::string::npos;) {
        const bool alone = (at == 0 || !word(feature_rspecifier[at - 1])) && (at + 3 >= feature_rspecifier.size() || !word(feature_rspecifier[at + 3]));
        if (!alone) { at += 3; continue; }
        feature_rspecifier.replace(at, 3, std::to_string(o.job_id));
        rewritten = true;
      }
      if (rewritten) log_line("LOG", "feature rspecifier of job " + std::to_string(o.job_id) + ": " + feature_rspecifier);
      if (rewritten && o.shard_shared_list) throw std::runtime_error("--shard-shared-list with a per-job (JOB) feature list");
    }
    const bool own_list = !o.shard_shared_list;
    const int device = o.device >= 0 ? o.device : (getenv("LOCAL_RANK") ? atoi(getenv("LOCAL_RANK")) : (world > 1 ? rank : 0));

    eesen_net_t* net = nullptr;
    eesen_ctc_t* ctc = nullptr;
    eesen_feeder_t* feeder = nullptr;
    eesen_comm_t* comm = nullptr;
    if (world > 1) {
      const std::string addr = !o.comm_addr.empty() ? o.comm_addr : (getenv("MASTER_ADDR") ? getenv("MASTER_ADDR") : "127.0.0.1");
      const int port = o.comm_port ? o.comm_port : getenv("EESEN_COMM_PORT") ? atoi(getenv("EESEN_COMM_PORT"))
                                                 : getenv("MASTER_PORT") ? atoi(getenv("MASTER_PORT")) + 17 : 29517;
      ck(eesen_comm_create_tcp(device, addr.c_str(), port, rank, world, o.comm_timeout, &comm));
      log_line("LOG", "job " + std::to_string(o.job_id) + " of " + std::to_string(world) + " joined the RCCL communicator on GPU " + std::to_string(device));
      {
        // Every job trains the WHOLE list it was given (reference semantics: the recipes hand job J its own feats_tr.J.scp).  A launcher
        // that hands all ranks the same command line without a JOB to substitute would make N jobs train identical minibatches and
        // sum N copies of the same gradient -- silently, an N times larger step on duplicated data.  The jobs compare
        // their rspecifiers: identical on every job = one shared list, refused unless --shard-shared-list=true deals it out or
        // --allow-identical-lists=true says the same words name different data on every node (node-local shards).
        // EVERY job enters this collective, whatever its options, and the two switches travel with the hash: jobs that disagree
        // on them get a message instead of a hang in a collective only some of them entered.  FNV-1a, the same in both trainers.
        unsigned long long h = 1469598103934665603ull;
        for (unsigned char c : feature_rspecifier) { h ^= c; h *= 1099511628211ull; }
        const double hv = (double)(h >> 16);             // 48 bits: exact in a double
        const double sh = o.shard_shared_list ? 1.0 : 0.0, al = o.allow_identical_lists ? 1.0 : 0.0;
        double mm[6] = {hv, -hv, sh, -sh, al, -al};
        ck(eesen_comm_allreduce_host(comm, mm, 6, /*max*/ 1));
        if (mm[2] != -mm[3]) throw std::runtime_error("the jobs disagree on --shard-shared-list: pass the same value to every job");
        if (mm[4] != -mm[5]) throw std::runtime_error("the jobs disagree on --allow-identical-lists: pass the same value to every job");
        if (own_list && !o.allow_identical_lists && mm[0] == -mm[1])
          throw std::runtime_error("all " + std::to_string(world) + " jobs were given the same feature rspecifier '" + feature_rspecifier +
                                   "': each job trains its whole list, so they would all train the same minibatches.  Hand every job its "
                                   "own list (feats.JOB.scp: a JOB that stands alone is replaced by the job id), pass --shard-shared-list=true "
                                   "to deal ONE shared list out, or --allow-identical-lists=true if the path names different data on every node");
      }
    }
    ck(eesen_net_create(device, nullptr, &net));
    ck(eesen_net_read(net, model_filename.c_str()));                                   // :111
    ck(eesen_net_set_train_options(net, o.learn_rate, o.momentum));                    // :112-113
    ck(eesen_net_set_adaptive_options(net, o.adagrad_epsilon, o.rms_prop_rho));
    ck(eesen_net_set_update_algorithm(net, o.opt_algorithm.c_str()));                  // :114
    ck(eesen_net_set_train_mode(net, o.cross_validate ? 0 : 1));                       // :116-119
    if (comm) {
      ck(eesen_net_set_dropout_seed(net, 777ull + (unsigned long long)rank));          // every job its own masks
      if (!o.cross_validate) ck(eesen_net_set_comm(net, comm));
    }
    ck(eesen_ctc_create(device, nullptr, &ctc));
    ck(eesen_ctc_set_guard(ctc, net));   // a minibatch computed from a timed-out forward pass never reaches the statistics
    ck(eesen_feeder_create(device, nullptr, 2, &feeder));
    if (!o.sequence_out_file.empty()) {                                                // :134-137
      log_line("LOG", "Sequences will be written to " + o.sequence_out_file + " in order from feature file");
      ck(eesen_ctc_set_sequence_out_file(ctc, o.sequence_out_file.c_str()));
    }
    int feat_dim = 0, K = 0;
    ck(eesen_net_input_dim(net, &feat_dim));
    ck(eesen_net_output_dim(net, &K));

    // A feature rspecifier that is a pipe of the reference's own filters (apply-cmvn | splice-feats | subsample-feats | add-deltas,
    // steps/train_ctc_parallel.sh:95-110): read the RAW table here, run the filters on the device inside the batch assembly.
    Pipeline pipe;
    const bool piped = !getenv("EESEN_HOST_FEATURE_PIPES") && parse_feature_pipeline(feature_rspecifier, &pipe);
    std::unique_ptr<CmvnTable> cmvn_table;
    if (piped) {
      ck(eesen_feeder_set_pipeline(feeder, pipe.stages.data(), (int)pipe.stages.size()));
      if (!pipe.cmvn.empty()) cmvn_table.reset(new CmvnTable(pipe.cmvn, pipe.utt2spk, pipe.norm_vars));
    }
    FeatureReader feature_reader(piped ? pipe.source : feature_rspecifier);
    const std::map<std::string, std::vector<int32_t>> targets_reader = read_targets(targets_rspecifier);
    log_line("LOG", std::string(o.cross_validate ? "CROSS-VALIDATION" : "TRAINING") + " STARTED");   // :133
    const auto t0 = std::chrono::steady_clock::now();
    long num_done = 0, num_no_tgt_mat = 0, num_other_error = 0;
    double total_frames = 0;
    std::vector<std::string> warnings;

    // the while(1) loop of :144-183: greedy groups of up to num_sequence utterances within frame_limit padded frames
    long batch_index = 0;
    auto next_group = [&](Minibatch* mb) -> bool {
      mb->mats.clear(); mb->labels.clear(); mb->frames.clear(); mb->raw_frames.clear(); mb->cmvn.clear(); mb->T = 0;
      int max_frame_num = 0;
      for (; !feature_reader.Done(); feature_reader.Next()) {
        const std::string utt = feature_reader.Key();
        auto tg = targets_reader.find(utt);
        if (tg == targets_reader.end()) {                                               // :152-156
          warnings.push_back(utt + ", missing targets");
          ++num_no_tgt_mat;
          continue;
        }
        if (tg->second.empty() || tg->second.size() > 2047) {   // (include/eesen_hip.h, eesen_ctc_eval_parallel: lattices of up to 4096 positions)  // what the CTC cannot take (the reference reads alpha column -1 for an empty transcript, ctc-loss.cc:151)
          warnings.push_back(utt + (tg->second.empty() ? ", empty transcript; ignoring" : ", more labels than the 2047 a lattice sweep holds; ignoring"));
          ++num_other_error;
          continue;
        }
        Mat& mat = feature_reader.Value();
        int rows = mat.rows, cols = mat.cols;
        const float* cm = nullptr;
        if (piped) {  // what the filters would have dropped never reaches the trainer (apply-cmvn.cc:87-92, add-deltas.cc:55-58, subsample-feats.cc:87-92)
          if (cmvn_table) {
            const std::vector<float>* n = cmvn_table->lookup(utt);
            if (!n) { warnings.push_back("No normalization statistics available for key " + utt + ", producing no output for this utterance"); continue; }
            if (CmvnTable::dim(*n) != mat.cols)
              throw std::runtime_error("Dim mismatch in ApplyCmvn: cmvn 2x" + std::to_string(CmvnTable::dim(*n) + 1) + ", feats " + std::to_string(mat.rows) + "x" + std::to_string(mat.cols));
            cm = n->data();
          }
          if (mat.rows == 0) { warnings.push_back("Empty feature matrix for key " + utt); continue; }
          ck(eesen_feeder_pipeline_shape(feeder, mat.cols, mat.rows, &cols, &rows));
          if (rows == 0) { warnings.push_back("For utterance " + utt + ", output would have no rows, producing no output."); continue; }
        }
        if (rows > o.frame_limit) {                                                     // :161-164
          warnings.push_back(utt + ", has too many frames; ignoring: " + std::to_string(rows) + " > " + fmt_g(o.frame_limit));
          continue;
        }
        if (cols != feat_dim) throw std::runtime_error("feature dimension " + std::to_string(cols) + " does not match the net's InputDim " + std::to_string(feat_dim));
        const int new_max = std::max(max_frame_num, rows);
        if ((double)new_max * (mb->mats.size() + 1) > o.frame_limit) break;             // :170-172: opens the next group, reader not advanced
        max_frame_num = new_max;
        mb->frames.push_back(rows);
        mb->raw_frames.push_back(mat.rows);
        mb->cmvn.push_back(cm);
        mb->labels.push_back(tg->second);
        mb->mats.push_back(std::move(mat));
        if ((int)mb->mats.size() == o.num_sequence) { feature_reader.Next(); break; }   // :179-182
      }
      mb->T = max_frame_num;
      return !mb->mats.empty();
    };
    auto next_batch = [&](Minibatch* mb) -> bool {  // with a shared list, job J trains groups J-1, J-1+N, ...: neighbours in the
      for (;;) {                                    // length-sorted list land in the same synchronous step (prep_scps.sh:37-76 deals alike)
        if (!next_group(mb)) return false;
        const bool mine = own_list || world == 1 || batch_index % world == rank;
        ++batch_index;
        if (mine) return true;
      }
    };
    auto stage = [&](const Minibatch& mb) -> int {  // padding + interleave + upload on the feeder's stream (replaces :186-195)
      std::vector<const float*> ptr(mb.mats.size());
      for (size_t s = 0; s < mb.mats.size(); ++s) ptr[s] = mb.mats[s].v.data();
      int slot = -1;
      if (piped)
        ck(eesen_feeder_submit_raw(feeder, ptr.data(), mb.raw_frames.data(), nullptr, cmvn_table ? mb.cmvn.data() : nullptr, (int)mb.mats.size(),
                                   mb.mats[0].cols, &slot));
      else
        ck(eesen_feeder_submit(feeder, ptr.data(), mb.frames.data(), nullptr, (int)mb.mats.size(), feat_dim, &slot));
      return slot;
    };

    Minibatch cur, nxt;
    bool have = next_batch(&cur);
    int slot = have ? stage(cur) : -1;
    float* diff = nullptr;
    long diff_cap = 0;
    double obj_prog = 0, err_prog = 0, ref_prog = 0;
    long seq_since_report = 0;
    long zero_steps = 0;
    for (;;) {
      if (!have) {
        // Jobs may hold different numbers of minibatches.  One that is out of data keeps stepping with a zero gradient through
        // the same collectives until NO job had a minibatch in a step (the liveness word that rides with the top layer's
        // bucket); that closing round, which all jobs take together, leaves the model untouched.  Cross-validation exchanges
        // nothing, so every job simply finishes.
        if (!comm || o.cross_validate) break;
        int live = 0;
        ck(eesen_net_backpropagate_zero(net));
        ck(eesen_net_update(net));
        ck(eesen_net_live_ranks(net, &live));
        if (live == 0) break;
        ++zero_steps;
        continue;
      }
      const int S = (int)cur.mats.size();
      float* feats = nullptr;
      int T = 0, S2 = 0, ld = 0;
      ck(eesen_feeder_acquire(feeder, slot, &feats, &T, &S2, &ld));
      ck(eesen_net_set_seq_lengths(net, cur.frames.data(), S));                        // :195
      const float* net_out = nullptr;
      int out_cols = 0, out_ld = 0;
      ck(eesen_net_propagate(net, feats, T * S, ld, /*in_is_device*/ 1, &net_out, &out_cols, &out_ld));   // :198
      ck(eesen_feeder_release(feeder, slot));
      std::vector<int> ids, off(1, 0);
      for (const auto& l : cur.labels) { ids.insert(ids.end(), l.begin(), l.end()); off.push_back((int)ids.size()); }
      if ((long)T * S * out_ld > diff_cap) {   // (grows by half: the list is sorted by length, T rises from minibatch to minibatch)
        if (diff) { ck(eesen_net_synchronize(net)); ck(eesen_dev_free(device, diff)); }
        diff_cap = std::max((long)T * S * out_ld, diff_cap + diff_cap / 2);
        ck(eesen_dev_alloc(device, diff_cap * 4, reinterpret_cast<void**>(&diff)));
      }
      // Neither call waits for the device: ln p and the decoded ids come back through pinned slots and join the statistics when
      // they are read (the reference's calls return nothing either and only accumulate, ctc-loss.cc:171-192,235-298)
      ck(eesen_ctc_eval_parallel(ctc, cur.frames.data(), S, net_out, T * S, out_cols, out_ld, ids.data(), off.data(), diff, out_ld, nullptr));  // :199
      ck(eesen_ctc_error_rate_mseq(ctc, cur.frames.data(), S, net_out, T * S, out_cols, out_ld, ids.data(), off.data(), nullptr, nullptr));     // :202
      if (!o.cross_validate) {                                                          // :206-208
        ck(eesen_net_backpropagate(net, diff, out_ld, nullptr, 0));
        ck(eesen_net_update(net));
      }
      have = next_batch(&nxt);                     // next batch: read and staged while the GPU runs this one's backward pass
      slot = have ? stage(nxt) : -1;
      num_done += S;
      total_frames += (double)T * S;                                                    // padded frames, as the reference counts them (:215)
      seq_since_report += S;
      if (o.verbose >= 1 && seq_since_report >= o.report_step) {                        // ctc-loss.cc:180-192: progress since the last report
        double obj; long seqs, frames, e, r;
        ck(eesen_ctc_stats(ctc, &obj, &seqs, &frames, &e, &r));
        log_line("VLOG[1]", "After " + std::to_string(seqs) + " sequences (" + fmt_g(frames / (100.0 * 3600)) + "Hr): Obj(log[Pzx]) = " +
                                fmt_g((obj - obj_prog) / seq_since_report) + "   TokenAcc = " +
                                fmt_g(100.0 * (1.0 - (e - err_prog) / std::max((double)r - ref_prog, 1.0))) + "%");
        obj_prog = obj; err_prog = (double)e; ref_prog = (double)r; seq_since_report = 0;
      }
      std::swap(cur, nxt);
    }
    for (const auto& w : warnings) log_line("WARNING", w);
    if (zero_steps) log_line("LOG", "job " + std::to_string(o.job_id) + " ran out of minibatches " + std::to_string(zero_steps) + " step(s) before the last job");
    ck(eesen_net_synchronize(net));
    if (!o.cross_validate) {                                                            // :236-240
      log_line("LOG", eesen_hip::NetInfo(net, 0));
      log_line("LOG", eesen_hip::NetInfo(net, 1, o.opt_algorithm != "SGD"));
    }
    if (!o.cross_validate && rank == 0) ck(eesen_net_write(net, target_model_filename.c_str(), o.binary ? 1 : 0));   // :244-246 (all ranks hold the same model)
    const double el = std::max(1e-9, std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count());
    log_line("LOG", "Done " + std::to_string(num_done) + " files, " + std::to_string(num_no_tgt_mat) + " with no targets, " +
                        std::to_string(num_other_error) + " with other errors. [" + (o.cross_validate ? "CROSS-VALIDATION" : "TRAINING") + ", " +
                        fmt_g(el / 60) + " min, fps" + fmt_g(total_frames / el) + "]");              // :247-252
    double obj; long seqs, frames, e, r, dropped = 0;
    ck(eesen_ctc_stats(ctc, &obj, &seqs, &frames, &e, &r));
    ck(eesen_ctc_dropped(ctc, &dropped));
    if (dropped) log_line("WARNING", std::to_string(dropped) + " minibatch(es) were computed from a timed-out forward pass and are not in the statistics");
    if (comm) {  // comm_touch_done (communicator.h:121-170): job 1 merges the jobs' Errors / Refs and reports the total
      double tot[2] = {(double)e, (double)r};
      ck(eesen_comm_allreduce_host(comm, tot, 2, 0));
      if (rank == 0) {
        log_line("LOG", "\nTOTAL TOKEN_ACCURACY >> " + fmt_g(100.0 * (1.0 - tot[0] / tot[1])) + "% <<");
        log_line("LOG", "\nTOKEN_ACCURACY >> " + fmt_g(100.0 * (1.0 - tot[0] / tot[1])) + "% <<");       // the line the recipes grep, once per run
      }
    } else {
      log_line("LOG", "\nTOKEN_ACCURACY >> " + fmt_g(100.0 * (1.0 - (double)e / (double)r)) + "% <<");     // ctc-loss.cc:300-304
    }
    if (diff) eesen_dev_free(device, diff);
    eesen_feeder_destroy(feeder);
    eesen_ctc_destroy(ctc);
    if (comm) eesen_net_set_comm(net, nullptr);
    eesen_net_destroy(net);
    if (comm) eesen_comm_destroy(comm);
    return 0;
  } catch (const std::exception& e) {  // :260-263
    std::cerr << e.what() << std::endl;
    return 255;
  }
}
