/* eesen_hip_layer.h -- Seam 2: a layer type for the reference's OWN eesen::Net whose compute runs in
 * libeesen_hip.so.
 *
 * A maintainer who keeps /root/reference/src/net/net.{h,cc} -- the layer chain, the buffers, Net::Read / Write, the per-layer
 * Update calls (net.cc:67-108) -- and wants only the LSTM layers on the MI355X drops this class in for
 * eesen::BiLstmParallel (src/net/bilstm-parallel-layer.h:31-913).  It IS a BiLstmParallel: the twelve parameter tensors,
 * ReadData / WriteData, GetParams, NumParams, Info, Scale / Add, the dropout options and their model-file tokens stay the
 * reference's (src/net/bilstm-layer.h:147-1031); what it overrides are the plug-in points of src/net/layer.h:82-164 and
 * src/net/trainable-layer.h:56-63:
 *
 *   PropagateFnc(in, out)                      -> eesen_net_propagate on a ONE-layer eesen_net_t
 *   BackpropagateFnc(in, out, out_diff, in_diff)-> eesen_net_backpropagate (gradients stay on the device)
 *   Update(input, diff, rule)                  -> eesen_net_update, then the twelve tensors are read back into the members
 *   SetSeqLengths, SetTrainMode / SetTestMode, SetTrainOptions pass through.
 *
 * Registration: `kMarkerMap` / `NewLayerOfType` (src/net/layer.cc:37-106) are a static table, so a tree that adopts this adds
 * one line there (`case l_BiLstm_Parallel: ans = new HipBiLstmParallel(input_dim, output_dim)`); without touching the
 * reference, HipBiLstmParallel::Adopt(net) swaps every BiLstmParallel of a Net that has been read (Net::SetLayer, net.h:65-67).
 *
 * The matrices a layer is handed are whatever the surrounding build's CuMatrix is.  In the CPU build of the reference (no
 * HAVE_CUDA: what oracle/ref_build compiles) that is host memory, and eesen_net_propagate(in_is_device = 0) uploads it; a
 * build whose CuMatrix lives in HBM passes the device pointers straight through (`kOnDevice`).  This seam costs the fusion
 * across layers (each layer's input GEMM cannot run under the previous layer's recurrence, activations cross the boundary
 * per layer): it is the minimal-adoption path, not what bench.py measures.  oracle/ref_build/seam2_check.cc compiles it
 * against the reference's own Net and holds it to the reference layer (tests/test_gpu_cli.py).
 */
#ifndef EESEN_HIP_LAYER_H_
#define EESEN_HIP_LAYER_H_

#include <stdexcept>
#include <string>
#include <vector>

#include "net/bilstm-parallel-layer.h" /* the reference's */
#include "net/net.h"                   /* the reference's */

#include "eesen_hip.h"

namespace eesen {

/* CuMatrixBase keeps its raw pointer protected (src/gpucompute/cuda-matrix.h:296-311; the reference's own kernels reach it as
 * friends).  A layer that hands the storage to another library needs it: taken through a pointer to the protected member,
 * which a derived class may form (no object of this type ever exists). */
template <typename Real>
struct CuRaw : public CuMatrixBase<Real> {
  static const Real* Get(const CuMatrixBase<Real>& m) {
    const Real* (CuMatrixBase<Real>::*f)() const = &CuRaw::Data;
    return (m.*f)();
  }
  static Real* Get(CuMatrixBase<Real>* m) {
    Real* (CuMatrixBase<Real>::*f)() = &CuRaw::Data;
    return (m->*f)();
  }
};

class HipBiLstmParallel : public BiLstmParallel {
 public:
  static const bool kOnDevice = false; /* CuMatrix data of this build is host memory (CPU build of the reference) */

  HipBiLstmParallel(int32 input_dim, int32 output_dim) : BiLstmParallel(input_dim, output_dim), h_(NULL), dev_stale_(true), dev_(0) {}
  explicit HipBiLstmParallel(const BiLstmParallel& other) : BiLstmParallel(other), h_(NULL), dev_stale_(true), dev_(0) {}
  HipBiLstmParallel(const HipBiLstmParallel& other) : BiLstmParallel(other), h_(NULL), dev_stale_(true), dev_(other.dev_), lens_(other.lens_) {}
  ~HipBiLstmParallel() {
    if (d_out_) eesen_dev_free(dev_, d_out_);
    if (d_in_) eesen_dev_free(dev_, d_in_);
    if (h_) eesen_net_destroy(h_);
  }

  Layer* Copy() const { return new HipBiLstmParallel(*this); }

  /* every BiLstmParallel of a Net that has been read becomes a HipBiLstmParallel with the same parameters and options */
  static int Adopt(Net* net, int device = 0) {
    int n = 0;
    for (int32 i = 0; i < net->NumLayers(); ++i) {
      if (net->GetLayer(i).GetType() != Layer::l_BiLstm_Parallel) continue;
      HipBiLstmParallel* hl = new HipBiLstmParallel(dynamic_cast<const BiLstmParallel&>(net->GetLayer(i)));
      hl->dev_ = device;
      net->SetLayer(i, hl); /* takes ownership, deletes the reference layer */
      ++n;
    }
    return n;
  }

  void SetSeqLengths(std::vector<int>& sequence_lengths) {
    BiLstmParallel::SetSeqLengths(sequence_lengths);
    lens_ = sequence_lengths;
  }

  void ReadData(std::istream& is, bool binary) {
    BiLstmParallel::ReadData(is, binary);
    dev_stale_ = true;
  }

  void PropagateFnc(const CuMatrixBase<BaseFloat>& in, CuMatrixBase<BaseFloat>* out) {
    Ensure();
    if (lens_.empty()) throw std::runtime_error("HipBiLstmParallel: SetSeqLengths must precede Propagate");
    Check(eesen_net_set_train_mode(h_, in_train ? 1 : 0));
    Check(eesen_net_set_seq_lengths(h_, lens_.data(), (int)lens_.size()));
    const float* o = NULL;
    int oc = 0, old = 0;
    Check(eesen_net_propagate(h_, CuRaw<BaseFloat>::Get(in), in.NumRows(), in.Stride(), kOnDevice ? 1 : 0, &o, &oc, &old));
    /* the layer's output: rows of [m_fw | m_bw] (bilstm-parallel-layer.h:409-419) */
    for (int32 r = 0; r < out->NumRows(); ++r)
      Check(eesen_dev_copy(dev_, CuRaw<BaseFloat>::Get(out) + (size_t)r * out->Stride(), o + (size_t)r * old, (long)out->NumCols() * sizeof(BaseFloat), kOnDevice ? 3 : 2));
  }

  void BackpropagateFnc(const CuMatrixBase<BaseFloat>& in, const CuMatrixBase<BaseFloat>& out, const CuMatrixBase<BaseFloat>& out_diff,
                        CuMatrixBase<BaseFloat>* in_diff) {
    (void)in; (void)out;
    const int32 rows = out_diff.NumRows(), co = out_diff.NumCols(), ci = in_diff->NumCols();
    const int32 ldo = (co + 3) & ~3, ldi = (ci + 3) & ~3;
    Scratch(&d_out_, &cap_out_, (long)rows * ldo);
    Scratch(&d_in_, &cap_in_, (long)rows * ldi);
    for (int32 r = 0; r < rows; ++r)
      Check(eesen_dev_copy(dev_, d_out_ + (size_t)r * ldo, CuRaw<BaseFloat>::Get(out_diff) + (size_t)r * out_diff.Stride(), (long)co * sizeof(BaseFloat), kOnDevice ? 3 : 1));
    Check(eesen_net_backpropagate(h_, d_out_, ldo, d_in_, ldi));
    for (int32 r = 0; r < rows; ++r)
      Check(eesen_dev_copy(dev_, CuRaw<BaseFloat>::Get(in_diff) + (size_t)r * in_diff->Stride(), d_in_ + (size_t)r * ldi, (long)ci * sizeof(BaseFloat), kOnDevice ? 3 : 2));
  }

  /* bilstm-layer.h:846-955: momentum fold, clipping to +-max_grad_, the parameter step -- on the device; then the members
   * follow, so that WriteData / GetParams / Info of the reference see the updated tensors */
  void Update(const CuMatrixBase<BaseFloat>& input, const CuMatrixBase<BaseFloat>& diff, const UpdateRule rule = sgd_update) {
    (void)input; (void)diff;
    Ensure();
    Check(eesen_net_set_train_options(h_, opts_.learn_rate, opts_.momentum));
    Check(eesen_net_set_adaptive_options(h_, opts_.adagrad_epsilon, opts_.rmsprop_rho));
    Check(eesen_net_set_update_algorithm(h_, rule == adagrad_update ? "Adagrad" : rule == rmsprop_update ? "RMSProp" : "SGD"));
    Check(eesen_net_update(h_));
    PullParams();
  }

 private:
  void Check(int rc) const {
    if (rc != EESEN_OK) throw std::runtime_error(std::string("eesen_hip: ") + eesen_last_error());
  }
  void Scratch(float** p, long* cap, long n) {
    if (n <= *cap) return;
    if (*p) Check(eesen_dev_free(dev_, *p));
    void* q = NULL;
    Check(eesen_dev_alloc(dev_, n * (long)sizeof(float), &q));
    *p = static_cast<float*>(q);
    *cap = n;
  }
  /* the one-layer net on the device, with the members' current values (GetParams order = eesen_net_set_params order,
   * bilstm-layer.h:1000-1031) and the layer's options (learn_rate_coef_, max_grad_, the nine dropout tokens) */
  void Ensure() {
    if (!h_) {
      Check(eesen_net_create(dev_, NULL, &h_));
      Check(eesen_net_add_layer(h_, EESEN_LAYER_BILSTM_PARALLEL, InputDim(), OutputDim(), learn_rate_coef_, max_grad_));
      Check(eesen_net_finalize(h_));
      const float nine[9] = {forward_dropout, (float)forward_step_dropout, (float)forward_sequence_dropout, (float)recurrent_step_dropout,
                             (float)recurrent_sequence_dropout, (float)rnndrop, (float)no_mem_loss_dropout, recurrent_dropout, 0.f};
      Check(eesen_net_set_layer_dropout(h_, 0, nine));
      dev_stale_ = true;
    }
    if (dev_stale_) {
      Vector<BaseFloat> w;
      GetParams(&w);
      Check(eesen_net_set_params(h_, w.Data(), w.Dim()));
      dev_stale_ = false;
    }
  }
  void PullParams() {
    Vector<BaseFloat> w(NumParams());
    Check(eesen_net_get_params(h_, w.Data(), w.Dim()));
    const int32 H = cell_dim_, D = InputDim();
    int32 off = 0;
    CuMatrix<BaseFloat>* mats[4] = {&wei_gifo_x_fw_, &wei_gifo_m_fw_, &wei_gifo_x_bw_, &wei_gifo_m_bw_};
    CuVector<BaseFloat>* vecs[8] = {&bias_fw_, &phole_i_c_fw_, &phole_f_c_fw_, &phole_o_c_fw_, &bias_bw_, &phole_i_c_bw_, &phole_f_c_bw_, &phole_o_c_bw_};
    for (int dir = 0; dir < 2; ++dir) {
      const int32 dims[2] = {D, H};
      for (int k = 0; k < 2; ++k) {
        Matrix<BaseFloat> m(4 * H, dims[k]);
        m.CopyRowsFromVec(w.Range(off, 4 * H * dims[k]));
        mats[dir * 2 + k]->CopyFromMat(m);
        off += 4 * H * dims[k];
      }
      const int32 vd[4] = {4 * H, H, H, H};
      for (int k = 0; k < 4; ++k) {
        vecs[dir * 4 + k]->CopyFromVec(w.Range(off, vd[k]));
        off += vd[k];
      }
    }
  }

  eesen_net_t* h_;
  bool dev_stale_;
  int dev_;
  std::vector<int> lens_;
  float* d_out_ = NULL;
  float* d_in_ = NULL;
  long cap_out_ = 0, cap_in_ = 0;
};

}  // namespace eesen
#endif /* EESEN_HIP_LAYER_H_ */
