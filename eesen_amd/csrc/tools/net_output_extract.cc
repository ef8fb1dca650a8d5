// net_output_extract.cc -- the reference's forward-for-decoding binary (src/netbin/net-output-extract.cc) over the C-ABI of
// include/eesen_hip.h, host C++ only.  Per utterance: Net::Feedforward (net.cc:110-132) -> optional ApplyLog -> optional
// ClassPrior::SubtractOnLogpost (class-prior.cc:30-91), written as a float-matrix table.  The reference converts
// <BiLstmParallel> to the single-sequence <BiLstm> on read (layer.cc:164-170); here the same kernels run with S = 1, or with
// --num-sequence utterances padded together (an extension: padding is masked in both directions, so valid frames do not
// change by a bit).
#include <algorithm>
#include <cfloat>
#include <chrono>
#include <cmath>
#include <iostream>
#include <sstream>

#include "../../../include/eesen_hip.h"
#include "kaldi_tables.h"
#include "feat_pipeline.h"
#include "parse_options.h"

namespace {
using namespace ktab;

void ck(int rc) {
  if (rc != EESEN_OK) throw std::runtime_error(eesen_last_error());
}
// ClassPrior::ClassPrior (class-prior.cc:30-77): counts -> floor -> blank scaling -> normalise -> log, with FLT_MAX/2 added
// for the classes below the cutoff so that they get zero likelihood
std::vector<float> class_log_priors(const std::string& path, double prior_cutoff, double blank_scale) {
  std::ifstream f(path);
  if (!f) throw std::runtime_error("cannot open " + path);
  std::string txt((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
  for (char& c : txt) if (c == '[' || c == ']') c = ' ';
  std::istringstream ss(txt);
  std::vector<double> pri;
  for (double v; ss >> v;) pri.push_back(v);
  std::vector<float> mask(pri.size(), 0.f);
  for (size_t i = 0; i < pri.size(); ++i)
    if (pri[i] < prior_cutoff) { pri[i] = prior_cutoff; mask[i] = FLT_MAX / 2; }
  if (blank_scale != 1.0 && !pri.empty()) pri[0] *= blank_scale;
  double sum = 0;
  for (double v : pri) sum += v;
  std::vector<float> out(pri.size());
  for (size_t i = 0; i < pri.size(); ++i) out[i] = (float)std::log(pri[i] / sum) + mask[i];
  return out;
}

}  // namespace

int main(int argc, char** argv) {
  try {
    std::string class_frame_counts;
    float prior_scale = 1.f;
    double prior_cutoff = 1e-10, blank_scale = 1.0, frame_limit = 1e5;
    bool apply_log = false;
    int num_sequence = 1, device = 0;
    std::string use_gpu = "yes";
    // the reference's ParseOptions conventions (parse_options.h); options and help texts of net-output-extract.cc:39-47 and
    // ClassPriorOptions::Register (src/net/class-prior.h:46-57), then this tool's own
    eesen_tools::ParseOptions po(
        "Perform a forward pass through the network for classification/feature extraction.\n"
        "\n"
        "Usage:  net-output-extract [options] <model-in> <feature-rspecifier> <feature-wspecifier>\n"
        "e.g.: \n"
        "net-output-extract net ark:features.ark ark:output.ark\n");
    po.Register("class-frame-counts", &class_frame_counts, "Vector with frame-counts of classes to compute log-priors. (priors are typically "
                                                           "subtracted from log-posteriors or pre-softmax activations)");
    po.Register("prior-scale", &prior_scale, "Scaling factor to be applied on class-log-priors");
    po.Register("prior-cutoff", &prior_cutoff, "Classes with priors lower than cutoff will have 0 likelihood");
    po.Register("blank-scale", &blank_scale, "Scale probability of class 0 (blank) by this factor");
    po.Register("apply-log", &apply_log, "Transform network output to logscale");
    po.Register("use-gpu", &use_gpu, "yes|no|optional (accepted for the recipes' command lines; this tool always runs on the GPU)");
    po.Register("num-sequence", &num_sequence, "Utterances forwarded together (1 = the reference's one utterance at a time)");
    po.Register("frame-limit", &frame_limit, "Max number of frames forwarded together");
    po.Register("device", &device, "GPU index");
    po.Read(argc, argv);
    std::vector<std::string> args;
    for (int i = 1; i <= po.NumArgs(); ++i) args.push_back(po.GetArg(i));
    if (args.size() != 3) {
      po.PrintUsage();
      return 1;
    }
    eesen_net_t* net = nullptr;
    eesen_feeder_t* feeder = nullptr;
    ck(eesen_net_create(device, nullptr, &net));
    ck(eesen_net_read(net, args[0].c_str()));
    ck(eesen_net_set_train_mode(net, 0));                                    // net-output-extract.cc:76
    ck(eesen_feeder_create(device, nullptr, 1, &feeder));
    int D = 0, K = 0;
    ck(eesen_net_input_dim(net, &D));
    ck(eesen_net_output_dim(net, &K));
    std::vector<float> log_pri;
    if (!class_frame_counts.empty()) {
      log_pri = class_log_priors(class_frame_counts, prior_cutoff, blank_scale);
      if ((int)log_pri.size() != K)
        throw std::runtime_error("Dimensionality mismatch, class_frame_counts " + std::to_string(log_pri.size()) + " class_output_llk " + std::to_string(K));
    }
    // decode_ctc_lat.sh:92-95 feeds this tool `apply-cmvn ... | splice-feats ... | subsample-feats ... | add-deltas ... |`: recognised,
    // the raw table is read here and the filters run on the device (feat_pipeline.h)
    Pipeline pipe;
    const bool piped = !getenv("EESEN_HOST_FEATURE_PIPES") && parse_feature_pipeline(args[1], &pipe);
    std::unique_ptr<CmvnTable> cmvn_table;
    if (piped) {
      ck(eesen_feeder_set_pipeline(feeder, pipe.stages.data(), (int)pipe.stages.size()));
      if (!pipe.cmvn.empty()) cmvn_table.reset(new CmvnTable(pipe.cmvn, pipe.utt2spk, pipe.norm_vars));
    }
    FeatureReader reader(piped ? pipe.source : args[1]);
    MatrixWriter writer(args[2]);
    const auto t0 = std::chrono::steady_clock::now();
    long num_done = 0;
    double tot_t = 0;
    std::vector<std::pair<std::string, Mat>> group;
    std::vector<int> out_frames;            // per utterance of the group: frames behind the pipeline
    std::vector<const float*> cmvn;
    std::vector<float> host;
    auto flush = [&]() {
      const int S = (int)group.size();
      std::vector<const float*> ptr(S);
      std::vector<int> frames(out_frames), raw_frames(S);
      for (int s = 0; s < S; ++s) { ptr[s] = group[s].second.v.data(); raw_frames[s] = group[s].second.rows; }
      int slot = 0, T = 0, S2 = 0, ld = 0;
      float* feats = nullptr;
      if (piped) ck(eesen_feeder_submit_raw(feeder, ptr.data(), raw_frames.data(), nullptr, cmvn_table ? cmvn.data() : nullptr, S, group[0].second.cols, &slot));
      else ck(eesen_feeder_submit(feeder, ptr.data(), frames.data(), nullptr, S, D, &slot));
      ck(eesen_feeder_acquire(feeder, slot, &feats, &T, &S2, &ld));
      ck(eesen_net_set_seq_lengths(net, frames.data(), S));
      const float* out = nullptr;
      int oc = 0, old = 0;
      ck(eesen_net_propagate(net, feats, T * S, ld, 1, &out, &oc, &old));
      ck(eesen_feeder_release(feeder, slot));
      if (apply_log || !log_pri.empty())                                       // :103-112
        ck(eesen_op_log_sub_prior(device, nullptr, const_cast<float*>(out), T * S, K, old, apply_log ? 1 : 0,
                                  log_pri.empty() ? nullptr : log_pri.data(), prior_scale));
      host.resize((size_t)T * S * K);
      ck(eesen_net_get_output(net, host.data(), (long)host.size()));
      for (int s = 0; s < S; ++s) {  // rows t*S + s of the dense [T*S x K] copy
        writer.Write(group[s].first, host.data() + (size_t)s * K, frames[s], K, S * K);
        ++num_done;
        tot_t += frames[s];
      }
      group.clear(); out_frames.clear(); cmvn.clear();
    };
    int max_len = 0;
    for (; !reader.Done(); reader.Next()) {
      Mat& m = reader.Value();
      int rows = m.rows, cols = m.cols;
      const float* cm = nullptr;
      if (piped) {  // what the filters would have dropped (apply-cmvn.cc:87-92, add-deltas.cc:55-58, subsample-feats.cc:87-92)
        const std::string& utt = reader.Key();
        if (cmvn_table) {
          const std::vector<float>* n = cmvn_table->lookup(utt);
          if (!n) { std::cerr << "WARNING (net-output-extract:main()) No normalization statistics available for key " << utt << ", producing no output for this utterance" << std::endl; continue; }
          if (CmvnTable::dim(*n) != m.cols)
            throw std::runtime_error("Dim mismatch in ApplyCmvn: cmvn 2x" + std::to_string(CmvnTable::dim(*n) + 1) + ", feats " + std::to_string(m.rows) + "x" + std::to_string(m.cols));
          cm = n->data();
        }
        if (m.rows == 0) { std::cerr << "WARNING (net-output-extract:main()) Empty feature matrix for key " << utt << std::endl; continue; }
        ck(eesen_feeder_pipeline_shape(feeder, m.cols, m.rows, &cols, &rows));
        if (rows == 0) { std::cerr << "WARNING (net-output-extract:main()) For utterance " << utt << ", output would have no rows, producing no output." << std::endl; continue; }
      }
      if (cols != D) throw std::runtime_error("feature dimension " + std::to_string(cols) + " does not match the net's InputDim " + std::to_string(D));
      if (!group.empty() && ((int)group.size() == num_sequence || (double)std::max(max_len, rows) * (group.size() + 1) > frame_limit)) {
        flush();
        max_len = 0;
      }
      max_len = std::max(max_len, rows);
      out_frames.push_back(rows);
      cmvn.push_back(cm);
      group.emplace_back(reader.Key(), std::move(m));
    }
    if (!group.empty()) flush();
    const double el = std::max(1e-9, std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count());
    std::cerr << "LOG (net-output-extract:main()) Done " << num_done << " files in " << el / 60 << "min, (fps " << tot_t / el << ")" << std::endl;
    eesen_feeder_destroy(feeder);
    eesen_net_destroy(net);
    return num_done ? 0 : 255;
  } catch (const std::exception& e) {
    std::cerr << "ERROR (net-output-extract:main()) " << e.what() << std::endl;
    return 255;
  }
}
