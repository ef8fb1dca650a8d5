// net.h -- host-side Net / Ctc objects behind the C-ABI (include/eesen_hip.h).
#pragma once
#include <string>
#include <vector>

#include "kernels.h"
#include "tuning.h"

namespace eesen {

inline int pad4(int x) { return (x + 3) & ~3; }
constexpr size_t kCtlWords = 2 * kCtlHalf + 32;  // counters [0, kCtlHalf) forward, [kCtlHalf, 2 kCtlHalf) backward, last word = error flag

struct Layer {
  int kind = 0, din = 0, dout = 0;
  float coef = 1.f, max_grad = 0.f;
  // parameter block inside the net-wide flat buffers (library-internal layout)
  size_t p_off = 0, p_n = 0;
  // LSTM kinds
  int H = 0, ndir = 0;
  // FILE-side dimensions: what the model file, the Net::GetParams order and eesen_net_layer_info carry.  They differ from the internal
  // ones (din, dout, H -- what every kernel sees) only around an LSTM layer whose cell count per direction is not a multiple of 4:
  // the library pads such a layer with cells that are identically zero and stay so (zero rows, bias and peepholes; zero columns
  // wherever their output is read -- the argument and its test: eesen_amd/model_tools.py pad-cells, tests/test_gpu_parity.py), because
  // the kernels fetch the state four cells at a time.  Parameter I/O (for_each_param), the statistics and the model writer map
  // between the two; nothing else knows.
  int din_f = 0, dout_f = 0, Hf = 0;
  // where this layer's FILE input column d lives in its internal input: in_nb runs of in_hf columns, each at a stride of in_hi
  // (in_nb = 0: identity); out_*: the same for the columns this layer hands on
  int in_nb = 0, in_hf = 0, in_hi = 0, out_nb = 0, out_hf = 0, out_hi = 0;
  int in_col(int d) const { return in_nb ? (d / in_hf) * in_hi + d % in_hf : d; }
  size_t off_wx = 0, off_bias = 0, off_wm = 0, off_peep = 0;  // relative to p_off
  DevBuf<float> WmT;      // [ndir][H x 4H], rebuilt after every parameter change
  DevBuf<float> G, C, Y;  // activations
  DevBuf<float> X;        // exchange copy of Y in the persistent forward kernel's fetch order (LstmLayerDev::X)
  // AffineTransform
  size_t off_w = 0, off_b = 0;
  DevBuf<float> out;  // affine / softmax output [rows x pad4(dout)]
  // operand bounds of the two-plane GEMMs (Net::amax): per row / per column of the weights, the input activation, the gradient
  struct Bounds { DevBuf<float> rows, cols; } bw, bx, bd;
  // Dropout options of BiLstm(Parallel) in model-file token order (bilstm-layer.h:331-373): ForwardDropoutFactor,
  // ForwardTimeStepDropout, ForwardSequenceDropout, RecurrentTimeStepDropout, RecurrentSequenceDropout, RNNDrop,
  // NoMemLossDropout, RecurrentDropoutFactor, TwiddleForward (booleans stored as 0 / 1)
  float drop[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
  bool has_dropout() const { return drop[0] > 0.f || drop[5] != 0.f || drop[6] != 0.f; }
  DevBuf<float> fmask;  // forward-dropout mask [T*S x ndir*H] of the current minibatch
  DevBuf<float> rmask;  // recurrent-dropout mask [(T+2)*S x ndir*H], indexed like C
  DevBuf<float> Yd;     // layer output after forward dropout [T*S x ndir*H]
  // what the last Propagate applied (Backpropagate must use the same, :888-889)
  bool cur_fwd_drop = false;
  int cur_drop_mode = 0;   // 0 none, 1 no-memory-loss, 2 RNNDrop
  bool cur_twiddle_coin = false;
  // one-shot injected masks (tests pin the kernels against the oracle with IDENTICAL masks): host copies until Propagate
  std::vector<float> inj_fmask, inj_rmask;
  int inj_rmask_rows = 0, inj_coin = -1;
  // the activation the next layer (and this layer's consumers) read
  const float* output(int S) const { return cur_fwd_drop ? Yd.p : Y.p + (size_t)S * ndir * H; }
  bool is_lstm() const { return kind == EESEN_LAYER_LSTM_PARALLEL || kind == EESEN_LAYER_BILSTM_PARALLEL; }
  bool nonparallel = false;  // read from a <BiLstm> / <Lstm> marker (Net::Write keeps the layer's own marker, layer.cc:37-46)
  bool is_activation() const { return kind == EESEN_LAYER_SIGMOID || kind == EESEN_LAYER_TANH; }
  bool trainable() const { return is_lstm() || kind == EESEN_LAYER_AFFINE; }
  long file_params() const;
  const char* marker() const;  // Layer::TypeToMarker (layer.cc:37-46, 68-79): the token the model file carries for this layer
};

class PhaseTimer {
 public:
  ~PhaseTimer();
  void enable(bool on) { on_ = on; }
  bool enabled() const { return on_; }
  // accumulate: spans pile up over several steps and are summed (and cleared) by one collect() -- no host
  // synchronisation per step is needed to time a multi-step region
  void set_accumulate(bool a) { accumulate_ = a; used_ = 0; }
  void reset() { if (!accumulate_) used_ = 0; }
  int begin(hipStream_t st, int phase);  // returns a span index (-1 when disabled)
  void end(hipStream_t st, int idx);
  void collect(float* out, int nphase);  // seconds per phase; synchronises on the recorded events
  int spans(int* phases, float* secs, int cap);  // every recorded span in record order (phase, seconds); does not clear
 private:
  struct Span { hipEvent_t a, b; int phase; };
  std::vector<Span> spans_;
  size_t used_ = 0;
  bool on_ = false, accumulate_ = false;
};

struct Comm;  // comm.cpp: RCCL communicator (one rank per GPU)
void comm_check_alive(const Comm* c);  // throws EESEN_ERR_COMM once the communicator's watchdog has aborted it (null: no-op)
constexpr size_t kLiveWords = 4;       // floats behind the gradient buffer: [0] = the data-parallel liveness word (comm.cpp)
// registers per SIMD lane one RCCL all-reduce workgroup needs to become resident on a CU: ncclDevKernel_Generic_{1,2,4} are 512-thread
// workgroups (two waves per SIMD) of 248-256 registers per lane -- at 256 threads per block one wave per SIMD: 256
// (profiles/r05_rccl_kernel_descriptors.md, read from librccl's gfx950 code object)
constexpr int kRcclVgprsPerSimdLane = 256;

struct Net {
  int device = 0;
  hipStream_t st = nullptr;
  bool own_stream = false;
  std::vector<Layer> layers;
  bool finalized = false;
  size_t P = 0;  // floats in the flat buffers (with alignment padding)
  DevBuf<float> params, corr, fresh;
  float lr = 0.f, mmt = 0.f;
  // UpdateRule (trainable-layer.h:38) + NetTrainOptions::adagrad_epsilon / rmsprop_rho (train-opts.h:33-42)
  int rule = 0;  // 0 SGD, 1 Adagrad, 2 RMSProp
  float ada_eps = 1e-6f, rms_rho = 0.9f, rms_one_minus_rho = 0.1f;
  DevBuf<float> accu;         // squared-gradient accumulators, parameter layout; allocated on first use / Read
  bool accu_init = false;
  void init_accu();
  void set_accu(const float* host, long n);
  // current minibatch
  std::vector<int> lens;
  DevBuf<int> lens_d;
  int T = 0, S = 0, rows = 0;
  bool propagated = false;
  DevBuf<float> input;  // [rows x pad4(din0)]
  struct HostStage { float* p = nullptr; size_t cap = 0; hipEvent_t ev = nullptr; bool busy = false; } in_stage[2];  // pinned staging of HOST inputs
  unsigned in_stage_idx = 0;
  const float* out_ptr = nullptr;
  int out_cols = 0, out_ld = 0;
  DevBuf<float> out_f;   // the net's output in the FILE's columns when the last layer is an LSTM layer padded to a multiple of 4 cells (forward_pass)
  // backward scratch
  DevBuf<float> DGb[2], DCF, dA, dB, ws, ws2;
  DevBuf<float> bwd_px;       // partial-sum exchange space of the K-split backward kernel (wide layers)
  DevBuf<float> bwd_dgh, bwd_ex;   // ... of its fp16-plane form: the gate gradients as planes, the inverse powers (LstmLayerDev::DGH / EX)
  hipStream_t st2 = nullptr;  // side stream: weight-gradient GEMMs under the next layer's recurrence
  hipEvent_t ev_rec = nullptr, ev_grad[2] = {nullptr, nullptr};
  bool overlap = true;
  hipEvent_t ev_gate_reset = nullptr, ev_gate_done = nullptr;  // gated / early input GEMM of the next layer (forward)
  DevBuf<unsigned> mile;      // progress milestone of the running forward recurrence (LstmLayerDev::milestone)
  bool gate_fwd = true;
  bool fwd_bf16 = false;      // eesen_net_set_forward_precision(1 | 2): forward GEMMs on bf16-rounded operands (BASELINE config 4)
  bool fwd_bf16_rec = false;  // ... (1 only): and the forward time recurrence on one bf16 plane of W_m / a bf16 exchange of m_t (lstm_fwd_persistent_bf16_kernel)
  int info_fwd_bf16 = 0;      // layers of the last Propagate whose recurrence ran on that kernel
  DevBuf<unsigned> ctl;       // arrival counters of the persistent recurrence kernels + [last] error word
  Tuning tn;                  // every run-time switch, read from the environment when the Net is created (tuning.h)
  int persistent = 1;         // EESEN_PERSISTENT=0 forces the one-launch-per-step kernels; also cleared by a recovery
  int info_fwd_persistent = 0, info_bwd_persistent = 0, info_lstm_layers = 0;   // of the last Propagate / Backpropagate (tests)
  int spin_limit = 400000;
  float flight_ns = 0.f;      // measured increment flight between two CUs of this device
  int delay_fwd = 61, delay_bwd = 43, delay_bwd_side = 64, poll_raw = 0;   // first-poll delays derived from it, wall-clock ticks of 10 ns
  int recoveries = 0;         // times a timed-out persistent kernel made the net fall back to the per-step kernels
  DevBuf<unsigned long long> trace;  // EESEN_TRACE=1 debug timeline
  void check_device_error(bool consumer);
  int steps_since_clean = 0;  // Propagates enqueued since the error word was last seen clear
  // set_seq_lengths does not drain the stream: the lengths go through a pinned staging word-array (the previous copy has
  // long completed; waiting for it bounds the host's run-ahead to one step), and the persistent kernels' error word is
  // polled through an asynchronous copy enqueued behind every Propagate / Update (full check in sync()).
  int* lens_pin = nullptr;
  size_t lens_pin_cap = 0;
  hipEvent_t lens_ev = nullptr, err_ev = nullptr;
  unsigned* err_pin = nullptr;
  bool err_armed = false;
  void poll_device_error();      // non-blocking
  void arm_device_error_poll();  // enqueue the copy of the error word
  // dropout
  bool in_train = true;                       // BiLstm::in_train (bilstm-layer.h:38), Net::SetTrainMode / SetTestMode
  unsigned long long drop_seed = 777, drop_counter = 0;   // masks are a pure function of (seed, draw counter, element)
  void set_train_mode(bool train) { in_train = train; }
  void set_layer_dropout(int layer, const float* nine);
  void get_layer_dropout(int layer, float* nine) const;
  void set_dropout_masks(int layer, const float* fwd, long fwd_n, const float* rec, int rec_rows, long rec_n, int coin);
  void get_dropout_masks(int layer, float* fwd_host, float* rec_host, int* info4);
  size_t ws_floats = 0;
  // Operand bounds of the two-plane fp16 GEMMs (gemm.hip, mode 2; kernels.h: GemmBound).  Every GEMM operand is scaled per row of
  // op(A) / column of op(B) by the power of two that its own bound asks for, so a dot product's precision does not depend on what
  // the rest of the tensor holds.  Per layer (Layer::bw / bx / bd): the bounds of the rows AND of the columns of [W] the weight matrix
  // its GEMMs multiply with (W_x / W; measured after every parameter change), [X] its input activation (measured in Propagate where the
  // structure gives no bound, reused by Backpropagate), [D] the gradient it multiplies in Backpropagate (DG of an LSTM layer, out_diff
  // of an affine one) -- one pass per tensor (amax_rows_cols).  What the structure bounds takes one word: `amax[0]` = 1.0 for LSTM
  // outputs (|m| = |o tanh c| < 1) and sigmoid / tanh / softmax outputs.  amax[1 + li]: max |W_m| of LSTM layer li (the fp16-plane
  // forward recurrence, lstm_persistent.hip).  The GEMM of a PART of a tensor takes the same words, so results do not depend on how a
  // GEMM is cut.
  DevBuf<float> amax, amax_ws;
  bool wamax_valid = false, wm_valid = false;
  bool amx_valid = false;     // the [X] bounds are those of the last Propagate
  float* am_wm(int li) { return amax.p + 1 + li; }
  GemmBound bound_one() const { return GemmBound{amax.p, 0}; }
  bool x_is_bounded(int li) const;                    // layer li's input needs no measurement (bounded by 1)
  void measure(const float* P, long rows, int cols, int ld, DevBuf<float>& out_rows, DevBuf<float>& out_cols);
  void ensure_weight_amax();
  PhaseTimer timer;
  // data-parallel exchange (comm.cpp): with a communicator attached, Backpropagate sums every layer's fresh gradients
  // over the ranks on the communicator's stream as soon as that layer's weight-gradient kernels are enqueued (the point
  // of the reference's per-layer Update, net.cc:98-104), and Update waits bucket by bucket
  Comm* comm = nullptr;                       // not owned
  std::vector<hipEvent_t> ev_ready, ev_bucket;
  std::vector<char> bucket_pending;
  // EESEN_COMM_DEFER=1 (tuning.h): the buckets of a backward pass are issued when its last recurrence has run instead of as each
  // layer's gradients are enqueued -- no all-reduce kernel then competes with a persistent grid for CUs (comm.cpp)
  std::vector<int> deferred_buckets;
  bool exchange_deferred = false;             // this backward pass's schedule (decided per minibatch: exchange_deferred_for_minibatch)
  bool overlap_for_minibatch() const;         // weight-gradient GEMMs on the side stream for the current shape (net.cpp)
  bool exchange_deferred_for_minibatch() const;
  std::string plan_string() const;            // eesen_net_plan_string
  hipEvent_t ev_bwd_done = nullptr;
  void issue_bucket(int li);
  void fail_step_buckets() noexcept;          // Backpropagate threw with peers waiting: complete the step's collective sequence (comm.cpp)
  void flush_deferred_buckets();
  std::vector<struct Ctc*> guards;   // the Ctc objects guarding on this Net's error word (eesen_ctc_set_guard): unhooked in ~Net
  bool grads_sanitized = false;   // this step's gradients went through an all-reduce that zeroed them on a raised error word: update() must apply
  bool live_valid = false;        // the liveness word behind the gradient buffer was written for THIS step (backpropagate / backpropagate_zero)
  std::vector<int> bucket_log;                // layer order of the last Backpropagate's buckets (tests)
  float* live_pin = nullptr;                  // pinned landing slot of the liveness word
  void set_comm(Comm* c);
  int top_trainable() const;
  int live_ranks();
  void bucket_allreduce(int li, hipStream_t producer);   // (deferred with EESEN_COMM_DEFER=1: see flush_deferred_buckets)
  void wait_buckets_host();
  void backpropagate_zero();
  void allreduce_grads(Comm* c);

  Net(int device, void* stream);
  ~Net();
  void add_layer(int kind, int din, int dout, float coef, float max_grad);
  void finalize();
  long num_params() const;
  void set_params(const float* host, long n);
  void get_flat(const DevBuf<float>& buf, float* host, long n);  // params or fresh grads in Net::GetParams order
  void set_seq_lengths(const int* lens, int S);
  void propagate(const float* in, int rows, int ld, bool is_device);
  void forward_pass();
  void backpropagate(const float* out_diff, int ldd, float* in_diff, int ldi);
  void backpropagate_impl(const float* out_diff, int ldd, float* in_diff, int ldi);
  void update();
  void refresh_derived();  // W_m^T copies
  // MomentStatistics (utils-functions.h:50-82) of the tensors of one layer in the reference's Info() / InfoGradient() order
  // (bilstm-layer.h:496-560, lstm-layer.h:175-196, affine-trans-layer.h:145-159).  which: 0 parameters, 1 the momentum
  // buffers (*_corr_), 2 the adaptive accumulators (*_corr_accu; zeros until an adaptive rule ran).  Returns the tensor count.
  int tensor_moments(int which, int layer, double* out6_host, int cap_tensors);
  void read(const std::string& path);
  void write(const std::string& path, bool binary);
  void sync();
};

struct Ctc {
  int device = 0;
  hipStream_t st = nullptr;
  bool own_stream = false;
  DevBuf<float> logp, alpha, beta, pzx_d;
  DevBuf<int> labx, lens_d, lablens_d, cls_off, cls_pos, ids_d;
  std::vector<int> last_lens;
  int last_T = 0, last_S = 0, last_Lpad = 0, last_Lprime = 0;
  int sweep_waves = 0;   // EESEN_CTC_WAVES when this object was created (tuning.h): 0 = the default number of waves per lattice
  double obj_sum = 0;
  long sequences = 0, frames = 0, err_tokens = 0, ref_tokens = 0;
  hipEvent_t ev[4] = {nullptr, nullptr, nullptr, nullptr};
  PhaseTimer timer;  // eesen_ctc_set_profiling(2): spans accumulate over many calls, read once (no per-call synchronisation)
  // Nothing in a training step has to stall the host: label staging goes through two alternating pinned slots, and the
  // per-sequence ln p / the greedy-decode ids of a call whose caller did not ask for them (NULL result pointers) come
  // back through pinned slots that are folded into the statistics at the next call that needs them ("deferred").
  struct Pin {
    void* p = nullptr;
    size_t cap = 0;            // bytes
    hipEvent_t ev = nullptr;   // last use of the slot by the device
    bool busy = false;
  };
  Pin stage[2];                // label expansion + class position lists (H2D)
  unsigned stage_idx = 0;
  struct PendingPzx { Pin pin; int S = 0; long nframes = 0; bool active = false; } ppzx[2];
  struct PendingErr { Pin pin, probs; int S = 0, K = 0, rows = 0; bool active = false, with_probs = false, guarded = false; std::vector<int> frames, ids, off; } perr[2];
  // eesen_ctc_set_guard: the error word of the Net whose outputs this Ctc evaluates.  Its value travels back with every
  // minibatch's ln p / decoded ids; a minibatch computed while it was set (a timed-out persistent forward pass: garbage
  // activations) is DROPPED from the statistics instead of being folded into the objective and TOKEN_ACCURACY.
  const unsigned* guard = nullptr;
  struct Net* guard_net = nullptr;   // whose error word `guard` points at: that Net unhooks the guard when it is destroyed first (eesen_ctc_set_guard)
  long dropped = 0;
  std::string seq_out;         // --sequence-out-file of the trainer (ctc-loss.cc:247-250,282-291): decoded sequences are appended here
  unsigned ppzx_idx = 0, perr_idx = 0;
  void* pin_reserve(Pin& pin, size_t bytes);  // waits for the slot's last use, grows it, returns the host pointer
  void flush_pzx(PendingPzx& q);
  void flush_err(PendingErr& q, int* num_err, int* num_ref);
  void flush();                // fold every deferred result into the statistics (blocks until they have arrived)

  Ctc(int device, void* stream);
  ~Ctc();
  void eval_parallel(const int* frame_num_utt, int S, const float* net_out, int rows, int K, int ld, const int* label_ids,
                     const int* label_off, float* diff, int ldd, float* pzx_host);
  void error_rate_mseq(const int* frame_num_utt, int S, const float* net_out, int rows, int K, int ld,
                       const int* label_ids, const int* label_off, int* num_err, int* num_ref);
  void get_alpha_beta(float* alpha_host, float* beta_host, int* Lprime);
  void phase_times(float* out3);
};

}  // namespace eesen
