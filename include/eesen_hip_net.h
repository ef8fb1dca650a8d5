/* eesen_hip_net.h -- the C++ seam: eesen::Net / eesen::Ctc / eesen::CuMatrix over the C-ABI of eesen_hip.h.
 *
 * With this header (and the four one-line forwarding headers of include/eesen_seam/, which shadow net/net.h,
 * net/ctc-loss.h, net/communicator.h and gpucompute/cuda-device.h on the include path) the reference's OWN trainer,
 * /root/reference/src/netbin/train-ctc-parallel.cc, compiles UNMODIFIED against libeesen_hip.so:
 *
 *   g++ -std=c++11 -I<repo>/include/eesen_seam -I<repo>/include -I<eesen>/src ... <eesen>/src/netbin/train-ctc-parallel.cc \
 *       <eesen base/util/cpucompute objects> -L<repo>/eesen_amd/lib -leesen_hip
 *
 * (oracle/ref_build/Makefile target `seam` does exactly that; tests/test_gpu_cli.py checks that the resulting binary and
 * eesen_amd/bin/train-ctc-parallel write byte-identical models.)  The reference's base / util / cpucompute layers (logging,
 * ParseOptions, Kaldi tables, the host Matrix) stay the reference's; everything below eesen::Net / eesen::Ctc --
 * src/net/{net,layer,*-layer,ctc-loss}, src/gpucompute -- is replaced.
 *
 * Each class states the reference interface it stands in for.  Only what train-ctc-parallel.cc (and net-output-extract.cc)
 * call is provided.  Errors: KALDI_ERR semantics, i.e. std::runtime_error carrying eesen_last_error().
 */
#ifndef EESEN_HIP_NET_H_
#define EESEN_HIP_NET_H_

#include <cstdlib>
#include <sstream>
#include <stdexcept>
#include <string>
#include <vector>

#include "base/kaldi-common.h"     /* the reference's: int32, BaseFloat, KALDI_LOG / KALDI_ERR */
#include "cpucompute/matrix.h"     /* the reference's host Matrix<BaseFloat> */
#include "net/train-opts.h"        /* the reference's NetTrainOptions (src/net/train-opts.h:29-62) */
#include "util/kaldi-io.h"

#include <cstdio>
#include <cstring>
#include <fstream>

#include "eesen_hip.h"
#include "eesen_hip_info.h"

namespace eesen {

inline void HipCheck(int rc) {  /* KALDI_ERR: message + std::runtime_error (src/base/kaldi-error.cc:168-182) */
  if (rc != EESEN_OK) throw std::runtime_error(std::string("eesen_hip: ") + eesen_last_error());
}

/* Which job of how many this process is.  A launcher may say so through RANK / WORLD_SIZE (torch.distributed.run, bench.py);
 * the recipes say it on the trainer's own command line, `--num-jobs=$nj --job-id=JOB` (train_ctc_parallel_h.sh:141-143, JOB
 * substituted by queue.pl), which train-ctc-parallel.cc parses into locals this seam never sees before the first minibatch --
 * so the command line is read here as well (/proc/self/cmdline).  rank = job id - 1. */
struct SeamJob {
  int rank, world;
  SeamJob() : rank(0), world(1) {
    const char* ws = std::getenv("WORLD_SIZE");
    const char* rk = std::getenv("RANK");
    if (ws && rk && std::atoi(ws) > 1) { world = std::atoi(ws); rank = std::atoi(rk); return; }
    std::ifstream f("/proc/self/cmdline", std::ios::binary);
    std::string arg;
    int nj = 1, id = 1;
    while (std::getline(f, arg, '\0')) {
      if (arg.compare(0, 11, "--num-jobs=") == 0) nj = std::atoi(arg.c_str() + 11);
      else if (arg.compare(0, 9, "--job-id=") == 0) id = std::atoi(arg.c_str() + 9);
    }
    if (nj > 1 && id >= 1 && id <= nj) { world = nj; rank = id - 1; }
  }
};
inline const SeamJob& Job() { static const SeamJob j; return j; }

inline int HipDevice() {  /* CuDevice::SelectGpuId (src/gpucompute/cuda-device.cc:73-140): EESEN_DEVICE, else LOCAL_RANK, else job id - 1 */
  const char* e = std::getenv("EESEN_DEVICE");
  if (!e) e = std::getenv("LOCAL_RANK");
  if (e) return std::atoi(e);
  int n = 0;
  if (Job().world > 1 && eesen_device_count(&n) == EESEN_OK && n > 0) return Job().rank % n;
  return 0;
}

/* The one communicator of the process (a job = one rank), created on first need: by Net::SetTrainMode when training, by
 * comm_touch_done when cross-validating (which exchanges nothing but the final counts). */
struct SeamComm {
  eesen_comm_t* comm;
  eesen_net_t* net;   /* the training net the communicator is attached to (NULL: none) */
  SeamComm() : comm(NULL), net(NULL) {}
  ~SeamComm() { if (comm) eesen_comm_destroy(comm); }
  eesen_comm_t* Get() {
    if (!comm && Job().world > 1) {
      const char* addr = std::getenv("MASTER_ADDR");
      const char* cp = std::getenv("EESEN_COMM_PORT");
      const char* mp = std::getenv("MASTER_PORT");
      const int port = cp ? std::atoi(cp) : (mp ? std::atoi(mp) + 17 : 29517);
      HipCheck(eesen_comm_create_tcp(HipDevice(), addr ? addr : "127.0.0.1", port, Job().rank, Job().world, 300, &comm));
    }
    return comm;
  }
};
inline SeamComm& TheComm() { static SeamComm c; return c; }
inline eesen_net_t*& LastNet() { static eesen_net_t* n = NULL; return n; }  /* the process's Net (the trainers hold exactly one) */

/* CuMatrixBase / CuMatrix (src/gpucompute/cuda-matrix.h:39-447): {data, MatrixDim{rows, cols, stride}}.  A CuMatrix built
 * from a host Matrix keeps the HOST pointer (eesen_net_propagate uploads it itself, as the reference's constructor would);
 * otherwise it owns device memory or is a view of memory the Net owns. */
template <typename Real>
class CuMatrixBase {
 public:
  int32 NumRows() const { return rows_; }
  int32 NumCols() const { return cols_; }
  int32 Stride() const { return stride_; }
  const Real* Data() const { return data_; }
  Real* Data() { return data_; }
  bool OnHost() const { return host_; }

 protected:
  CuMatrixBase() : data_(NULL), rows_(0), cols_(0), stride_(0), host_(false) {}
  Real* data_;
  int32 rows_, cols_, stride_;
  bool host_;
};

template <typename Real>
class CuMatrix : public CuMatrixBase<Real> {
 public:
  CuMatrix() : own_(false) {}
  explicit CuMatrix(const MatrixBase<Real>& m) : own_(false) {  /* CuMatrix(const MatrixBase&) cuda-matrix.h:367 */
    this->data_ = const_cast<Real*>(m.Data());
    this->rows_ = m.NumRows(); this->cols_ = m.NumCols(); this->stride_ = m.Stride();
    this->host_ = true;
  }
  ~CuMatrix() { Release(); }
  /* device storage of at least rows x cols, rows 16-byte aligned (callee-resizes-output convention, layer.h:192, ctc-loss.cc:103) */
  void Resize(int32 rows, int32 cols) {
    const int32 ld = (cols + 3) & ~3;
    if (own_ && !this->host_ && (long)rows * ld <= cap_) { this->rows_ = rows; this->cols_ = cols; this->stride_ = ld; return; }
    Release();
    void* p = NULL;
    cap_ = (long)rows * ld;
    HipCheck(eesen_dev_alloc(HipDevice(), cap_ * (long)sizeof(Real), &p));
    this->data_ = static_cast<Real*>(p);
    this->rows_ = rows; this->cols_ = cols; this->stride_ = ld; this->host_ = false;
    own_ = true;
  }
  /* a view of memory owned elsewhere (the Net's output buffer) */
  void View(const Real* p, int32 rows, int32 cols, int32 stride) {
    Release();
    this->data_ = const_cast<Real*>(p);
    this->rows_ = rows; this->cols_ = cols; this->stride_ = stride; this->host_ = false;
  }
  void CopyToMat(Matrix<Real>* dst) const {  /* cuda-matrix.cc CopyToMat */
    dst->Resize(this->rows_, this->cols_);
    for (int32 r = 0; r < this->rows_; ++r)
      HipCheck(eesen_dev_copy(HipDevice(), dst->RowData(r), this->data_ + (size_t)r * this->stride_, (long)this->cols_ * sizeof(Real), 2));
  }

 private:
  CuMatrix(const CuMatrix&);
  CuMatrix& operator=(const CuMatrix&);
  void Release() {
    if (own_ && this->data_) eesen_dev_free(HipDevice(), this->data_);
    this->data_ = NULL; own_ = false; cap_ = 0;
  }
  bool own_;
  long cap_ = 0;
};

/* eesen::Net (src/net/net.h:37-175) for <BiLstmParallel> / <LstmParallel> / <AffineTransform> / <Softmax> stacks */
class Net {
 public:
  Net() : h_(NULL), attached_(false) { HipCheck(eesen_net_create(HipDevice(), NULL, &h_)); LastNet() = h_; }
  ~Net() {
    if (LastNet() == h_) LastNet() = NULL;
    if (attached_) { eesen_net_set_comm(h_, NULL); TheComm().net = NULL; }
    if (h_) eesen_net_destroy(h_);
  }
  void Read(const std::string& file) { HipCheck(eesen_net_read(h_, file.c_str())); }                         /* net.cc:279-309 */
  /* net.cc:325-334.  With several jobs EVERY job of the reference trainer writes the target model (train-ctc-parallel.cc:244-246);
   * behind this seam they hold identical models, so each writes its own temporary file and renames it over the target */
  void Write(const std::string& file, bool binary) const {
    if (Job().world <= 1) { HipCheck(eesen_net_write(h_, file.c_str(), binary)); return; }
    std::ostringstream tmp;
    tmp << file << ".job" << Job().rank + 1 << ".$$";
    HipCheck(eesen_net_write(h_, tmp.str().c_str(), binary));
    if (std::rename(tmp.str().c_str(), file.c_str()) != 0) throw std::runtime_error("eesen_hip: cannot rename " + tmp.str() + " to " + file);
  }
  void SetTrainOptions(const NetTrainOptions& o) {                                                          /* net.h:147-153 */
    HipCheck(eesen_net_set_train_options(h_, o.learn_rate, o.momentum));
    HipCheck(eesen_net_set_adaptive_options(h_, o.adagrad_epsilon, o.rmsprop_rho));
  }
  void SetUpdateAlgorithm(std::string opt) { HipCheck(eesen_net_set_update_algorithm(h_, opt.c_str())); }   /* net.cc:481-497 */
  void SetTrainMode() {                                                                                     /* net.cc:405-412 */
    HipCheck(eesen_net_set_train_mode(h_, 1));
    /* one job of several (SeamJob): join the RCCL communicator; from here on Backpropagate sums the gradients over the jobs
     * (replaces comm_avg_weights, communicator.h:39-119) */
    if (!attached_ && Job().world > 1) {
      HipCheck(eesen_net_set_dropout_seed(h_, 777ull + (unsigned long long)Job().rank));
      HipCheck(eesen_net_set_comm(h_, TheComm().Get()));
      attached_ = true;
      TheComm().net = h_;
    }
  }
  void SetTestMode() { HipCheck(eesen_net_set_train_mode(h_, 0)); }                                         /* net.cc:396-403 */
  void SetSeqLengths(std::vector<int>& lens) { HipCheck(eesen_net_set_seq_lengths(h_, lens.data(), (int)lens.size())); }  /* net.h:157 */
  int32 InputDim() const { int d = 0; HipCheck(eesen_net_input_dim(h_, &d)); return d; }                    /* net.cc:139-142 */
  int32 OutputDim() const { int d = 0; HipCheck(eesen_net_output_dim(h_, &d)); return d; }
  int32 NumParams() const { long n = 0; HipCheck(eesen_net_num_params(h_, &n)); return (int32)n; }
  /* Net::Propagate (net.cc:67-86): `in` is what the trainer builds from the padded host matrix (train-ctc-parallel.cc:198),
   * or a device matrix; *out becomes a view of the Net-owned output, valid until the next Propagate */
  void Propagate(const CuMatrixBase<BaseFloat>& in, CuMatrix<BaseFloat>* out) {
    const float* o = NULL;
    int oc = 0, old = 0;
    HipCheck(eesen_net_propagate(h_, in.Data(), in.NumRows(), in.Stride(), in.OnHost() ? 0 : 1, &o, &oc, &old));
    out->View(o, in.NumRows(), oc, old);
  }
  void Feedforward(const CuMatrixBase<BaseFloat>& in, CuMatrix<BaseFloat>* out) { Propagate(in, out); }     /* net.cc:110-132 */
  /* Net::Backpropagate (net.cc:88-108): gradients, then every layer's Update (:101-104).  With a communicator attached to the
   * handle (eesen_net_set_comm) the per-layer gradient all-reduce runs in between, inside the library. */
  void Backpropagate(const CuMatrixBase<BaseFloat>& out_diff, CuMatrix<BaseFloat>* in_diff) {
    if (in_diff) in_diff->Resize(out_diff.NumRows(), InputDim());
    HipCheck(eesen_net_backpropagate(h_, out_diff.Data(), out_diff.Stride(), in_diff ? in_diff->Data() : NULL, in_diff ? in_diff->Stride() : 0));
    HipCheck(eesen_net_update(h_));
  }
  /* net.cc:336-366: topology, layer markers and the MomentStatistics of every parameter tensor / momentum buffer (device-side
   * reductions, eesen_net_tensor_moments; strings arranged by eesen_hip_info.h) */
  std::string Info() const { return eesen_hip::NetInfo(h_, 0); }
  std::string InfoGradient() const { return eesen_hip::NetInfo(h_, 1); }
  eesen_net_t* Handle() { return h_; }

 private:
  Net(const Net&);
  Net& operator=(const Net&);
  eesen_net_t* h_;
  bool attached_;
};

/* eesen::Ctc (src/net/ctc-loss.h:29-90), the multi-sequence entry points */
class Ctc {
 public:
  Ctc() : h_(NULL), report_step_(100), frames_per_sec_(100.0f), seq_progress_(0), obj_last_(0), err_last_(0), ref_last_(0) {
    HipCheck(eesen_ctc_create(HipDevice(), NULL, &h_));
  }
  ~Ctc() { if (h_) eesen_ctc_destroy(h_); }
  /* ctc-loss.cc:101-194; diff is resized by the callee (:103).  Nothing here waits for the device: ln p joins the objective
   * when it has arrived, as the reference's call only accumulates obj_progress_ (:171-177). */
  void EvalParallel(const std::vector<int32>& frame_num_utt, const CuMatrixBase<BaseFloat>& net_out,
                    std::vector<std::vector<int32> >& label, CuMatrix<BaseFloat>* diff) {
    const int S = (int)frame_num_utt.size();
    Csr(label, S);
    if (!guarded_ && LastNet()) { HipCheck(eesen_ctc_set_guard(h_, LastNet())); guarded_ = true; }  /* a timed-out forward pass never reaches the statistics */
    diff->Resize(net_out.NumRows(), net_out.NumCols());
    HipCheck(eesen_ctc_eval_parallel(h_, frame_num_utt.data(), S, net_out.Data(), net_out.NumRows(), net_out.NumCols(), net_out.Stride(),
                                     ids_.data(), off_.data(), diff->Data(), diff->Stride(), NULL));
    seq_progress_ += S;
    if (seq_progress_ >= report_step_) {  /* :180-192 */
      double obj = 0; long seqs = 0, frames = 0, e = 0, r = 0;
      HipCheck(eesen_ctc_stats(h_, &obj, &seqs, &frames, &e, &r));
      KALDI_VLOG(1) << "After " << seqs << " sequences (" << frames / (frames_per_sec_ * 3600) << "Hr): Obj(log[Pzx]) = "
                    << (obj - obj_last_) / seq_progress_ << "   TokenAcc = " << 100.0 * (1.0 - (double)(e - err_last_) / std::max<long>(r - ref_last_, 1)) << "%";
      obj_last_ = obj; err_last_ = e; ref_last_ = r; seq_progress_ = 0;
    }
  }
  /* ctc-loss.cc:235-298 (returns void: the host part runs deferred, under the device's backward pass) */
  void ErrorRateMSeq(const std::vector<int>& frame_num_utt, const CuMatrixBase<BaseFloat>& net_out, std::vector<std::vector<int> >& label,
                     std::string& out) {
    const int S = (int)frame_num_utt.size();
    Csr(label, S);
    if (out != seq_out_) { HipCheck(eesen_ctc_set_sequence_out_file(h_, out.empty() ? NULL : out.c_str())); seq_out_ = out; }
    HipCheck(eesen_ctc_error_rate_mseq(h_, frame_num_utt.data(), S, net_out.Data(), net_out.NumRows(), net_out.NumCols(), net_out.Stride(),
                                       ids_.data(), off_.data(), NULL, NULL));
  }
  void SetReportStep(int32 s) { report_step_ = s; }
  void SetFramesPerSec(float f) { frames_per_sec_ = f; }
  float NumErrorTokens() const { long e = 0; HipCheck(eesen_ctc_stats(h_, NULL, NULL, NULL, &e, NULL)); return (float)e; }   /* ctc-loss.h:62 */
  int32 NumRefTokens() const { long r = 0; HipCheck(eesen_ctc_stats(h_, NULL, NULL, NULL, NULL, &r)); return (int32)r; }
  std::string Report() {  /* ctc-loss.cc:300-304; the line train_ctc_parallel.sh:146,158 greps */
    std::ostringstream oss;
    oss << "\nTOKEN_ACCURACY >> " << 100.0 * (1.0 - NumErrorTokens() / NumRefTokens()) << "% <<";
    return oss.str();
  }

 private:
  void Csr(const std::vector<std::vector<int32> >& label, int S) {  /* vector<vector<int32>> -> CSR */
    ids_.clear(); off_.assign(1, 0);
    for (int s = 0; s < S; ++s) { ids_.insert(ids_.end(), label[s].begin(), label[s].end()); off_.push_back((int)ids_.size()); }
    if (ids_.empty()) ids_.push_back(0);
  }
  eesen_ctc_t* h_;
  int32 report_step_;
  float frames_per_sec_;
  long seq_progress_;
  double obj_last_;
  long err_last_, ref_last_;
  std::vector<int> ids_, off_;
  std::string seq_out_;
  bool guarded_ = false;
};

/* src/net/communicator.h: the multi-job mode of the reference trainer (--num-jobs / --job-id / --utts-per-avg) averages MODELS
 * through files.  Behind this seam a job is one rank of an RCCL communicator that sums GRADIENTS every minibatch inside
 * eesen_net_backpropagate (eesen_net_set_comm): the jobs hold identical models at every step, so "averaging" is the identity.
 * What is left of the protocol happens where the reference trainer calls comm_touch_done, once, after its loop
 * (train-ctc-parallel.cc:226-232):
 *   * jobs with FEWER minibatches than others (the recipes' per-job lists are only approximately even, prep_scps.sh:37-76;
 *     the reference lets a sub-job "give up if the main job finishes first", communicator.h:104-112) keep stepping here with a
 *     zero gradient through the same collectives until no job had a minibatch in a step (eesen_net_live_ranks); the closing
 *     round leaves the model untouched, so the result equals one process on the union of the minibatches;
 *   * the jobs' error / reference token counts are summed over the communicator (the done-files of communicator.h:121-170)
 *     and job 1 prints the `TOTAL TOKEN_ACCURACY` line the recipes grep in its log (train_ctc_parallel_h.sh:147,171). */
inline std::string comm_done_filename(const std::string& base, int job_id) {  /* communicator.h:29-31 */
  std::ostringstream os;
  os << base << ".done.job" << job_id;
  return os.str();
}
inline std::string comm_avg_model_name(const std::string& base, int count) {  /* communicator.h:33-35 */
  std::ostringstream os;
  os << base << ".avg" << count;
  return os.str();
}
inline void comm_avg_weights(Net&, const int&, const int&, const int&, const std::string&, const std::string&) {}  /* communicator.h:39-119 */
inline void comm_touch_done(Ctc& ctc, const int& job_id, const int& num_jobs, const std::string& base_done_filename) {  /* communicator.h:121-170 */
  KALDI_LOG << "Writing done file for job" << job_id;
  {  /* :125-129: the file itself, for whoever looks at the experiment directory */
    std::ofstream out(comm_done_filename(base_done_filename, job_id).c_str());
    out << "Errors " << ctc.NumErrorTokens() << " Refs " << ctc.NumRefTokens();
  }
  if (num_jobs <= 1 || Job().world <= 1) return;
  eesen_comm_t* comm = TheComm().Get();
  if (TheComm().net) {  /* a training job: out of minibatches, others may not be */
    long zero_steps = 0;
    for (;;) {
      int live = 0;
      HipCheck(eesen_net_backpropagate_zero(TheComm().net));
      HipCheck(eesen_net_update(TheComm().net));
      HipCheck(eesen_net_live_ranks(TheComm().net, &live));
      if (live == 0) break;
      ++zero_steps;
    }
    if (zero_steps) KALDI_LOG << "job " << job_id << " ran out of minibatches " << zero_steps << " step(s) before the last job";
  }
  double tot[2] = {(double)ctc.NumErrorTokens(), (double)ctc.NumRefTokens()};
  HipCheck(eesen_comm_allreduce_host(comm, tot, 2, 0));
  if (job_id == 1) {  /* :132, :168 */
    KALDI_LOG << "Collecting stats from " << num_jobs << " done files";
    KALDI_LOG << "\nTOTAL TOKEN_ACCURACY >> " << 100.0 * (1.0 - tot[0] / tot[1]) << "% <<";
  }
}

}  // namespace eesen
#endif  /* EESEN_HIP_NET_H_ */
