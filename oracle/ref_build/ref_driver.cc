// oracle/ref_build/ref_driver.cc -- TEST INFRASTRUCTURE (oracle), not product code.
//
// A thin extern "C" driver over the REAL reference classes, compiled together with the reference's
// own, unmodified sources (see Makefile) into oracle/_ref/libeesen_ref.so.  Nothing here restates
// reference arithmetic: it only marshals plain float arrays into eesen::Matrix / CuMatrix (CPU
// mode: CuDevice is never enabled, so every CuMatrix op runs the src/cpucompute branch) and calls
//   eesen::Net::{Read,Write,SetTrainOptions,SetSeqLengths,Propagate,Backpropagate,GetParams}
//     (/root/reference/src/net/net.h:48-161, net.cc:67-108,181-195,279-334)
//   eesen::Ctc::ErrorRateMSeq (/root/reference/src/net/ctc-loss.cc:235-298).
// The reference's CTC loss itself has no CPU branch; it is provided by ref_cuda_emul.cc.
//
// Used by: tests/ (checker), oracle/make_golden.py (fixture generation), bench.py's
// cpu_baseline leg (kind "reference").  Never by the product path.

#include <string>
#include <vector>
#include <cstring>
#include <stdexcept>

#include "net/net.h"
#include "net/ctc-loss.h"
#include "net/bilstm-layer.h"
#include "net/bilstm-parallel-layer.h"
#include "net/train-opts.h"
#include "cpucompute/matrix.h"
#include "cpucompute/compressed-matrix.h"
#include "gpucompute/cuda-matrix.h"
#include "util/kaldi-table.h"
#include "util/kaldi-holder.h"
#include "util/table-types.h"

using namespace eesen;

namespace {

thread_local std::string g_last_error;

struct RefNet {
  Net net;
  Ctc ctc;
};

void ToCu(const float* src, int rows, int cols, CuMatrix<BaseFloat>* dst) {
  Matrix<BaseFloat> m(rows, cols, kUndefined);
  for (int r = 0; r < rows; r++) memcpy(m.RowData(r), src + (size_t)r * cols, sizeof(float) * cols);
  dst->Resize(rows, cols, kUndefined);
  dst->CopyFromMat(m);
}

void FromCu(const CuMatrixBase<BaseFloat>& src, float* dst) {
  Matrix<BaseFloat> m(src.NumRows(), src.NumCols(), kUndefined);
  src.CopyToMat(&m);
  for (int r = 0; r < m.NumRows(); r++) memcpy(dst + (size_t)r * m.NumCols(), m.RowData(r), sizeof(float) * m.NumCols());
}

}  // namespace

#define REF_TRY try {
#define REF_CATCH(ret)                                      \
  } catch (const std::exception& e) {                       \
    g_last_error = e.what();                                \
    return ret;                                             \
  }

extern "C" {

// SciPy's OpenBLAS threading control (symbol is prefixed like the rest of that library).
void scipy_openblas_set_num_threads(int);
int scipy_openblas_get_num_threads(void);

const char* ref_last_error() { return g_last_error.c_str(); }

void ref_set_blas_threads(int n) { scipy_openblas_set_num_threads(n); }
int ref_get_blas_threads() { return scipy_openblas_get_num_threads(); }

void* ref_net_read(const char* path) {
  REF_TRY
  RefNet* h = new RefNet();
  h->net.Read(std::string(path));
  // the trainer's defaults (netbin/train-ctc-parallel.cc:79,113-119); update_algorithm is otherwise uninitialised
  h->net.SetUpdateAlgorithm("SGD");
  h->net.SetTrainMode();
  return h;
  REF_CATCH(nullptr)
}

void ref_net_free(void* p) { delete static_cast<RefNet*>(p); }

int ref_net_write(void* p, const char* path, int binary) {
  REF_TRY
  static_cast<RefNet*>(p)->net.Write(std::string(path), binary != 0);
  return 0;
  REF_CATCH(-1)
}

int ref_net_input_dim(void* p) { return static_cast<RefNet*>(p)->net.InputDim(); }
int ref_net_output_dim(void* p) { return static_cast<RefNet*>(p)->net.OutputDim(); }
int ref_net_num_params(void* p) { return static_cast<RefNet*>(p)->net.NumParams(); }

int ref_net_get_params(void* p, float* out) {
  REF_TRY
  Vector<BaseFloat> v;
  static_cast<RefNet*>(p)->net.GetParams(&v);
  memcpy(out, v.Data(), sizeof(float) * v.Dim());
  return v.Dim();
  REF_CATCH(-1)
}

int ref_net_set_train_options(void* p, float learn_rate, float momentum) {
  REF_TRY
  NetTrainOptions o;
  o.learn_rate = learn_rate;
  o.momentum = momentum;
  static_cast<RefNet*>(p)->net.SetTrainOptions(o);
  return 0;
  REF_CATCH(-1)
}

int ref_net_set_update_algorithm(void* p, const char* alg) {
  REF_TRY
  static_cast<RefNet*>(p)->net.SetUpdateAlgorithm(std::string(alg));
  return 0;
  REF_CATCH(-1)
}

int ref_net_set_seq_lengths(void* p, const int* lens, int S) {
  REF_TRY
  std::vector<int> v(lens, lens + S);
  static_cast<RefNet*>(p)->net.SetSeqLengths(v);
  return 0;
  REF_CATCH(-1)
}

// in: [rows x InputDim] (row = t*S+s), out: [rows x OutputDim]
int ref_net_propagate(void* p, const float* in, int rows, float* out) {
  REF_TRY
  RefNet* h = static_cast<RefNet*>(p);
  CuMatrix<BaseFloat> cin, cout;
  ToCu(in, rows, h->net.InputDim(), &cin);
  h->net.Propagate(cin, &cout);
  FromCu(cout, out);
  return 0;
  REF_CATCH(-1)
}

// out_diff: [rows x OutputDim]; in_diff (nullable): [rows x InputDim].  Includes the per-layer Update
// (net.cc:101-104), exactly like the reference trainer.
int ref_net_backpropagate(void* p, const float* out_diff, int rows, float* in_diff) {
  REF_TRY
  RefNet* h = static_cast<RefNet*>(p);
  CuMatrix<BaseFloat> cdiff, cin_diff;
  ToCu(out_diff, rows, h->net.OutputDim(), &cdiff);
  h->net.Backpropagate(cdiff, in_diff ? &cin_diff : NULL);
  if (in_diff) FromCu(cin_diff, in_diff);
  return 0;
  REF_CATCH(-1)
}

// The same loop as Net::Backpropagate (net.cc:88-108) -- every layer's own Backpropagate + Update, top down, through the public
// Layer / TrainableLayer interface -- except that a BiLstm layer's four [(T+2)S x 7H] state buffers are released as soon as the
// layer is done with them.  Net keeps them all until the next minibatch (bilstm-layer.h:1131-1136), which at BASELINE
// configs[2]'s global minibatch (S = 256, T = 1000, 4 x 512: 59 GB) or at the 3000-frame bucket of configs[4] does not fit the
// authoring container's memory.  Arithmetic and order of operations are the reference's; nothing is restated.
namespace {
struct BufPeek : public BiLstm {
  static CuMatrix<BaseFloat> BiLstm::*pf() { return &BufPeek::propagate_buf_fw_; }
  static CuMatrix<BaseFloat> BiLstm::*pb() { return &BufPeek::propagate_buf_bw_; }
  static CuMatrix<BaseFloat> BiLstm::*bf() { return &BufPeek::backpropagate_buf_fw_; }
  static CuMatrix<BaseFloat> BiLstm::*bb() { return &BufPeek::backpropagate_buf_bw_; }
};
}  // namespace

int ref_net_backpropagate_lowmem(void* p, const float* out_diff, int rows, float* in_diff) {
  REF_TRY
  Net& net = static_cast<RefNet*>(p)->net;
  const std::vector<CuMatrix<BaseFloat> >& act = net.PropagateBuffer();
  const int L = net.NumLayers();
  if ((int)act.size() != L + 1) throw std::runtime_error("ref_net_backpropagate_lowmem: Propagate first");
  CuMatrix<BaseFloat> d_out, d_in;
  ToCu(out_diff, rows, net.OutputDim(), &d_out);
  for (int i = L - 1; i >= 0; i--) {
    Layer& l = net.GetLayer(i);
    l.Backpropagate(act[i], act[i + 1], d_out, &d_in);
    if (l.IsTrainable()) dynamic_cast<TrainableLayer&>(l).Update(act[i], d_out, sgd_update);
    if (BiLstm* b = dynamic_cast<BiLstm*>(&l)) {
      (b->*BufPeek::pf()).Resize(0, 0); (b->*BufPeek::pb()).Resize(0, 0);
      (b->*BufPeek::bf()).Resize(0, 0); (b->*BufPeek::bb()).Resize(0, 0);
    }
    d_out.Swap(&d_in);
    d_in.Resize(0, 0);
  }
  if (in_diff) FromCu(d_out, in_diff);
  return 0;
  REF_CATCH(-1)
}

// Copies layer `layer`'s input activation buffer (propagate_buf_[layer]) — layer==NumLayers gives the output.
int ref_net_get_propagate_buf(void* p, int layer, float* out, int capacity) {
  REF_TRY
  const std::vector<CuMatrix<BaseFloat> >& b = static_cast<RefNet*>(p)->net.PropagateBuffer();
  if (layer < 0 || layer >= (int)b.size()) return -1;
  if ((long)b[layer].NumRows() * b[layer].NumCols() > capacity) return -2;
  FromCu(b[layer], out);
  return b[layer].NumCols();
  REF_CATCH(-1)
}

// Real Ctc::ErrorRateMSeq (argmax → collapse repeats → drop blanks → Levenshtein), ctc-loss.cc:235-298.
int ref_ctc_error_rate_mseq(void* p, const float* net_out, int T, int S, int K, const int* frame_num_utt,
                            const int* label_ids, const int* label_off, float* num_err, int* num_ref) {
  REF_TRY
  RefNet* h = static_cast<RefNet*>(p);
  CuMatrix<BaseFloat> cout;
  ToCu(net_out, T * S, K, &cout);
  std::vector<int> lens(frame_num_utt, frame_num_utt + S);
  std::vector<std::vector<int> > labels(S);
  for (int s = 0; s < S; s++) labels[s].assign(label_ids + label_off[s], label_ids + label_off[s + 1]);
  std::string out_file;
  float e0 = h->ctc.NumErrorTokens();
  int r0 = h->ctc.NumRefTokens();
  h->ctc.ErrorRateMSeq(lens, cout, labels, out_file);
  *num_err = h->ctc.NumErrorTokens() - e0;
  *num_ref = h->ctc.NumRefTokens() - r0;
  return 0;
  REF_CATCH(-1)
}

// ---- dropout: the reference draws its masks from the host RNG inside PropagateFnc
// (bilstm-parallel-layer.h:46-94); they are protected members of BiLstm (bilstm-layer.h:1054-1062), read here through
// member pointers formed in a derived class (no change to the reference, no access-specifier macros) so that the SAME masks
// can be fed to the restatement and to the HIP library.
namespace {
struct MaskPeek : public BiLstm {
  static CuMatrix<BaseFloat> BiLstm::*fwd() { return &MaskPeek::forward_drop_mask_; }
  static CuMatrix<BaseFloat> BiLstm::*rec_fw() { return &MaskPeek::recurrent_drop_mask_fw_; }
  static CuMatrix<BaseFloat> BiLstm::*rec_bw() { return &MaskPeek::recurrent_drop_mask_bw_; }
  static bool BiLstm::*twiddle_apply() { return &MaskPeek::twiddle_apply_forward; }
};
}  // namespace

int ref_net_set_mode(void* p, int train) {
  REF_TRY
  if (train) static_cast<RefNet*>(p)->net.SetTrainMode();
  else static_cast<RefNet*>(p)->net.SetTestMode();
  return 0;
  REF_CATCH(-1)
}

// Masks of BiLstm(Parallel) layer `layer` as left behind by the last Propagate.  dims = {fwd rows, fwd cols, rec rows, rec cols
// (per direction), twiddle_apply_forward}; a buffer may be NULL to query the dims only.  Returns -2 if the layer is no BiLstm.
int ref_net_get_dropout_masks(void* p, int layer, float* fwd, float* rec_fw, float* rec_bw, int* dims) {
  REF_TRY
  Net& net = static_cast<RefNet*>(p)->net;
  BiLstm* l = dynamic_cast<BiLstm*>(&net.GetLayer(layer));
  if (!l) return -2;
  const CuMatrix<BaseFloat>& f = l->*MaskPeek::fwd();
  const CuMatrix<BaseFloat>& rf = l->*MaskPeek::rec_fw();
  const CuMatrix<BaseFloat>& rb = l->*MaskPeek::rec_bw();
  dims[0] = f.NumRows(); dims[1] = f.NumCols(); dims[2] = rf.NumRows(); dims[3] = rf.NumCols();
  dims[4] = (l->*MaskPeek::twiddle_apply()) ? 1 : 0;
  if (fwd && f.NumRows()) FromCu(f, fwd);
  if (rec_fw && rf.NumRows()) FromCu(rf, rec_fw);
  if (rec_bw && rb.NumRows()) FromCu(rb, rec_bw);
  return 0;
  REF_CATCH(-1)
}

// ---- Kaldi tables through the reference's OWN reader / writer classes (util/kaldi-table.h, util/table-types.h) --------
// Used to pin eesen_amd/kaldi_io.py: files written here must be read identically by our reader and vice versa.
int ref_write_feats(const char* wspecifier, int n, const char** keys, const float** mats, const int* rows, int cols) {
  REF_TRY
  BaseFloatMatrixWriter w(wspecifier);
  for (int i = 0; i < n; i++) {
    Matrix<BaseFloat> m(rows[i], cols, kUndefined);
    for (int r = 0; r < rows[i]; r++) memcpy(m.RowData(r), mats[i] + (size_t)r * cols, sizeof(float) * cols);
    w.Write(keys[i], m);
  }
  return 0;
  REF_CATCH(-1)
}

int ref_write_labels(const char* wspecifier, int n, const char** keys, const int** labs, const int* lens) {
  REF_TRY
  Int32VectorWriter w(wspecifier);
  for (int i = 0; i < n; i++) w.Write(keys[i], std::vector<int32>(labs[i], labs[i] + lens[i]));
  return 0;
  REF_CATCH(-1)
}

// Sequential read of a feature table: number of utterances, total rows, and sum over all elements of value * (1 + col).
int ref_read_feats_summary(const char* rspecifier, int* n_utts, long* total_rows, double* checksum, char* keys_out, int keys_cap) {
  REF_TRY
  SequentialBaseFloatMatrixReader r(rspecifier);
  int n = 0; long rows = 0; double cs = 0; std::string keys;
  for (; !r.Done(); r.Next()) {
    const Matrix<BaseFloat>& m = r.Value();
    for (int i = 0; i < m.NumRows(); i++)
      for (int j = 0; j < m.NumCols(); j++) cs += (double)m(i, j) * (1 + j);
    rows += m.NumRows(); n++;
    keys += r.Key(); keys += " ";
  }
  *n_utts = n; *total_rows = rows; *checksum = cs;
  if (keys_out && keys_cap > 0) { strncpy(keys_out, keys.c_str(), keys_cap - 1); keys_out[keys_cap - 1] = 0; }
  return 0;
  REF_CATCH(-1)
}

// Compressed feature archives (what `copy-feats --compress=true` writes): the reference's CompressedMatrix does the lossy
// encoding AND tells us what its own CopyToMat decodes, so our reader can be pinned bit-exactly against it.
// `decoded` [sum rows x cols] receives the reference's decompression of every matrix, back to back.
int ref_write_compressed_feats(const char* wspecifier, int n, const char** keys, const float** mats, const int* rows, int cols,
                               float* decoded) {
  REF_TRY
  CompressedMatrixWriter w(wspecifier);
  size_t o = 0;
  for (int i = 0; i < n; i++) {
    Matrix<BaseFloat> m(rows[i], cols, kUndefined);
    for (int r = 0; r < rows[i]; r++) memcpy(m.RowData(r), mats[i] + (size_t)r * cols, sizeof(float) * cols);
    CompressedMatrix cm(m);
    w.Write(keys[i], cm);
    Matrix<BaseFloat> back(rows[i], cols, kUndefined);
    cm.CopyToMat(&back);
    for (int r = 0; r < rows[i]; r++, o += cols) memcpy(decoded + o, back.RowData(r), sizeof(float) * cols);
  }
  return 0;
  REF_CATCH(-1)
}

// Random-access read of one label vector (RandomAccessInt32VectorReader, as train-ctc-parallel.cc:125 uses it).
int ref_read_labels(const char* rspecifier, const char* key, int* out, int cap) {
  REF_TRY
  RandomAccessInt32VectorReader r(rspecifier);
  if (!r.HasKey(key)) return -2;
  const std::vector<int32>& v = r.Value(key);
  if ((int)v.size() > cap) return -3;
  for (size_t i = 0; i < v.size(); i++) out[i] = v[i];
  return (int)v.size();
  REF_CATCH(-1)
}

}  // extern "C"
