// kernels.h -- host-side launchers of the gfx950 kernels (definitions in gemm.hip, lstm.hip, ctc.hip, optim.hip).
// All pointers are device pointers; every launcher enqueues on `st` and returns without synchronising.
#pragma once
#include "common.h"

namespace eesen {

// ---------------------------------------------------------------------------------------- gemm.hip
// C[M x N] = alpha * op(A) * op(B) + beta * C (+ bias[n]).  fp32 in, fp32 accumulate on
// v_mfma_f32_32x32x2_f32.  a_kc: A stored [M x K] (k contiguous) else [K x M]; b_kc: B stored [N x K]
// else [K x N].  All base pointers 16-byte aligned and all leading dimensions multiples of 4.
// `ws`/`ws_floats`: split-K workspace (may be null => no split).  Replaces the cublasSgemm behind
// CuMatrixBase::AddMatMat (/root/reference/src/gpucompute/cuda-matrix.cc:604-639).
// Mode 2 (two fp16 planes): a bound of an operand's magnitudes, in device words -- per_index = 1: one word per M index of op(A) /
// N index of op(B) (the bound of that row / column: the scale is then the dot product's own), 0: one word for the whole operand.
// p = null: measured by a pass over the operand before the launch.  Ignored by the other modes.
struct GemmBound {
  const float* p = nullptr;
  int per_index = 0;
};
void gemm_f32(hipStream_t st, bool a_kc, bool b_kc, int M, int N, int K, float alpha, const float* A, int lda,
              const float* B, int ldb, float beta, float* C, int ldc, const float* bias, float* ws,
              size_t ws_floats, int extra_lds_bytes = 0,   // extra_lds_bytes: unused dynamic LDS = occupancy cap per CU
              bool bf16_operands = false,                  // round both operands to bf16, one bf16 MFMA product, fp32 accumulate
              GemmBound bound_a = {}, GemmBound bound_b = {});

// Arithmetic of every gemm_f32 / gemm_f32_nt_gated call: 0 = f32-input MFMA, 1 = 3-way bf16 split (six products), 2 = two fp16
// planes (three products; gemm.hip); -1 = follow EESEN_GEMM_MODE.  Process-wide.
constexpr int kDefaultGemmMode = 2;
int gemm_mode();
void set_gemm_mode(int mode);
// *out = max |P[r][c]| over a [rows x cols] matrix with row stride ld (amax_abs zeroes the word first; _accumulate folds into it)
void amax_abs(hipStream_t st, const float* P, long rows, int cols, int ld, float* out);
void amax_abs_accumulate(hipStream_t st, const float* P, long rows, int cols, int ld, float* out);
// one pass: out_rows[r] = max_c |P[r][c]| and / or out_cols[c] = max_r |P[r][c]| (either may be null; out_cols needs a workspace
// `ws` of kAmaxBlocks * min(16384, pad4(cols)) floats)
constexpr int kAmaxBlocks = 512;
void amax_rows_cols(hipStream_t st, const float* P, long rows, int cols, int ld, float* out_rows, float* out_cols, float* ws);

// Arrival counters of the persistent recurrence kernels (lstm_persistent.hip): per (direction, sequence tile) group 8
// shards (shard = blockIdx.x & 7), one 128-byte line each; a shard counts workgroups-in-shard x completed steps.
#ifndef EESEN_SHARDS
#define EESEN_SHARDS 8
#endif
constexpr int kShards = EESEN_SHARDS, kShardStride = 32;  // words
constexpr int kCtlHalf = 2048 * kShards;                  // words of counters per pass (forward / backward): 64 groups

// "Gated" input->gates GEMM: C = A * B^T + bias where the rows of A are the time-major output [T*S x K] of an LSTM layer
// whose persistent forward kernel is STILL RUNNING on another stream.  A tile of rows waits until the arrival counters
// say that both directions have passed its frames (forward direction: steps >= t_hi + 1, backward: steps >= T - t_lo);
// tiles are visited middle-out in time, which is the order in which a bidirectional layer completes frames.
struct GemmGate {
  const unsigned* cnt;  // counters of the producing kernel
  unsigned* err;        // raised when the bounded spin gives up
  int ndir, nz, nblk;   // counter groups (ndir x nz) and workgroups per group
  int T, S;
  int spin_limit;
};
void gemm_f32_nt_gated(hipStream_t st, int M, int N, int K, const float* A, int lda, const float* B, int ldb, float* C,
                       int ldc, const float* bias, const GemmGate& gate, GemmBound bound_a = {}, GemmBound bound_b = {});

// ---------------------------------------------------------------------------------------- lstm.hip
struct LstmLayerDev {
  // geometry
  int T, S, H, ndir;
  // activations of this layer (see DESIGN.md "data layout")
  float* G;         // [T*S x ndir*4H]   gate-interleaved (col = dir*4H + u*4 + q, q in g,i,f,o)
  float* C;         // [(T+2)*S x ndir*H] cell state, row (t+1)*S+s, boundary row-blocks zero
  float* Y;         // [(T+2)*S x ndir*H] cell output m, same indexing; rows S.. are the layer output
  // Exchange copy of m for the persistent forward kernel (optional, may be null): the SAME values in the order the consumers
  // fetch them -- per (t, dir, 16-sequence tile) a block [chunk of 32 units][k half][k quad][16 sequences][4 floats], so that
  // 16 adjacent lanes of an operand load read 256 contiguous bytes (whole 128-byte lines, each requested once) instead of 16
  // bytes out of 16 different rows of Y.  T * ndir * ceil(S/16) * ceil(H/32) * 512 floats.
  float* X = nullptr;
  // parameters (internal layout)
  const float* Wm;   // [ndir][4H x H]   rows gate-interleaved (u*4+q)
  const float* WmT;  // [ndir][H x 4H]   transpose of the above
  const float* peep; // [ndir][3][H]     p_i, p_f, p_o
  const int* lens;   // [S]
  // recurrent dropout (bilstm-parallel-layer.h:209-377): mask of 0 / 1/(1-p), [(T+2)*S x ndir*H], indexed like C;
  // drop_mode 0 none, 1 no-memory-loss (mask multiplies g*i), 2 RNNDrop (mask multiplies the whole new cell)
  const float* rmask = nullptr;
  int drop_mode = 0;
  // sequence window of one persistent launch: sequences [s_begin, s_begin + s_count) of the S interleaved ones.  Shapes whose
  // whole batch needs more workgroups than can be co-resident (S = 64 at H = 1024: BASELINE config 5) run as several
  // launches, one window each -- the sequences are independent chains.  s_count = 0 means all S.
  int s_begin = 0, s_count = 0;
  // partial-sum exchange space of the K-split backward kernel for wide layers (lstm_bwd_ksplit_px_floats; null: not offered)
  float* PX = nullptr;
  size_t px_floats = 0;
  // first-poll delay of the persistent kernels' hand-off waits in wall-clock ticks of 10 ns; set by the host from the measured
  // increment flight of the device (handoff_flight_ns)
  int poll_delay = 0;
  int poll_raw = 0;   // 1: poll_delay is an experimenter's value (EESEN_POLL_NS): the launchers apply no per-plan factor to it
  // progress milestone of the narrow forward persistent kernel (null: none; two words): when every workgroup of a (direction, sequence
  // tile) group has published step `milestone_step`, the group's first workgroup adds 1 to milestone[0], and the last group sets
  // milestone[1] -- the host starts the GEMM that consumes this pass's rows for the frames both directions have finished by then
  // while the recurrence is still running (net.cpp: "the middle first")
  unsigned* milestone = nullptr;
  int milestone_step = 0;
  // kernel selection switches (tuning.h; all 1 in production): XCD-aware role map, time-multiplexed forward kernel, 4 x 32 and
  // K-split backward tiles
  int xcd_map = 1, fwd_mux = 1, bwd_q4 = 1, bwd_ksplit = 1, bwd_mux = 1;
  int fwd_t16_small = 1; // batches of <= 16 sequences on narrow layers: the 16 x 8 tile (bf16-pipe kernel) instead of the 32 x 4 fp32 tile (tuning.h: EESEN_FWD_T16_SMALL)
  int fwd_narrow2 = 1;  // the narrow bf16-pipe forward tile as up to two workgroups per CU where a census has seen them resident (tuning.h: EESEN_FWD_NARROW2)
  int bwd_q4_st8 = 1;   // the 4 x 32 backward tile with TWO 4-sequence tiles per workgroup where one does not fit (tuning.h: EESEN_BWD_Q4_ST8)
  // eesen_net_set_forward_precision(1): the recurrent product m_{t-1} W_m^T of the persistent forward kernel on bf16 operands with
  // fp32 accumulation (lstm_fwd_persistent_bf_kernel<.., AP = 1, WP = 2>): m_t as ONE bf16 plane in the exchange buffer X, W_m as
  // hi + lo planes
  int fwd_bf16 = 0;
  // fp32-class forward recurrence on the bf16 matrix pipe (3-way split of both operands, six products; tuning.h: EESEN_FWD_SPLIT)
  int fwd_split = 0;
  // fp32-class forward recurrence on TWO fp16 planes per operand, three products (round 6; tuning.h: EESEN_FWD_F16): taken where
  // the 3-way split would be, and on the wide 16 x 16 tile.  wm_amax: device word holding max |W_m| of this layer (both directions).
  int fwd_f16 = 0;
  const float* wm_amax = nullptr;
  // the K-split backward tile of wide layers on two fp16 planes per operand (round 6; tuning.h: EESEN_BWD_F16): DGH = the gate gradients a
  // second time, as planes with a per-(producer, sequence) power of two ([T*S x ndir*4H] words of 4 bytes: the fp32 rows' bytes), EX = the
  // inverse powers ([T][ndir][ceil(S/16)][H/16][4 groups of 4 sequences] words of 16 bytes); both null: not offered (lstm_bwd_planes_floats)
  int bwd_f16 = 0;
  unsigned char* DGH = nullptr;
  float* EX = nullptr;
};
// floats of LstmLayerDev::EX this layer shape needs when its backward pass may take the fp16-plane K-split tile (0: it will not)
size_t lstm_bwd_planes_ex_floats(const LstmLayerDev& L);
float handoff_flight_ns();
// The share of the device's CUs this PROCESS sizes its persistent grids against: 1/n (EESEN_GPU_SHARE, or set by a communicator
// that found n of its ranks on this device: comm.cpp).  Process-wide.
void set_gpu_share(int n);
int gpu_share_value();
// one wave that returns once *word >= target (or when the recurrence kernels' error word is raised; it raises that word itself
// if it ever gives up): puts a stream behind a milestone of a kernel that is still running on another stream
void wait_for_word(hipStream_t st, const unsigned* word, unsigned target, unsigned* err, double limit_s = 2.0);   // gives up after limit_s seconds of wall clock
// One recurrence step of every direction: fw direction handles t = step, bw direction t = T-1-step.
void lstm_fwd_step(hipStream_t st, const LstmLayerDev& L, int step);
// One step of the backward recurrence: fw direction handles t = T-1-step, bw direction t = step.
// dY: [T*S x ndir*H] gradient w.r.t. the layer output, DG: [T*S x ndir*4H] (out), DCF: [S x ndir*H] carry.
void lstm_bwd_step(hipStream_t st, const LstmLayerDev& L, int step, const float* dY, int lddy, float* DG, float* DCF);
// The whole recurrence of one layer in ONE cooperative launch with W_m resident in registers (lstm_persistent.hip).
// cnt: >= ndir * ceil(S/16) zero-initialisable counters, err: one word raised when a bounded spin gives up.
// Return false (nothing launched) when the shape does not fit the resident-workgroup budget: use the step kernels then.
bool lstm_fwd_persistent(hipStream_t st, const LstmLayerDev& L, unsigned* cnt, unsigned* err, int spin_limit,
                         unsigned long long* trace = nullptr, hipEvent_t after_reset = nullptr);
// counter geometry of the forward persistent kernel (for gated consumers): workgroups per group, sequence tiles
// out[r][c] = (u(r,c) > p) ? 1/(1-p) : 0 with u ~ U[0,1) from a counter-based hash of (seed, element) -- the mask recipe of
// bilstm-parallel-layer.h:54-61 (SetRandUniform / SetRandUniformCol, Add(-p), ApplyHeaviside, Scale(1/(1-p))); per_column
// draws one value per column and repeats it down the rows (SetRandUniformCol, cpucompute/matrix.cc:952-965).
void dropout_mask(hipStream_t st, float* out, long rows, int cols, int ld, float p, unsigned long long seed, bool per_column);
// out[r][c] = a[r][c] * m[r][c]   (MulElements, :416 / :895)
// <Sigmoid> / <Tanh> layers: y = f(x) (cuda-kernels.cu:687-697, 713-727) and d *= f'(y) in place (:699-709, :730-740)
void activation_rows(hipStream_t st, bool tanh_, const float* x, int ldx, float* y, int ldy, long rows, int cols);
void activation_diff_rows(hipStream_t st, bool tanh_, const float* y, int ldy, float* d, int ldd, long rows, int cols);
void mul_elements(hipStream_t st, const float* a, int lda, const float* m, int ldm, float* out, int ldo, long rows, int cols);
void lstm_fwd_persistent_geometry(const LstmLayerDev& L, int* nblk, int* nz, int* units_per_wg = nullptr);
// sequence windows (= cooperative launches) the persistent forward pass of this layer takes: 1 for every shape whose
// workgroups are co-resident at once, 2+ for S = 64 at H = 1024, 0 when no persistent tile fits (per-step kernels then)
int lstm_fwd_persistent_windows(const LstmLayerDev& L);
// true when lstm_fwd_persistent would run this layer's recurrence on the bf16 kernel (L.fwd_bf16 set and the shape allows)
bool lstm_fwd_persistent_is_bf16(const LstmLayerDev& L);
// the forward tile this layer takes leaves room on a CU for a 128 x 128 GEMM workgroup (net.cpp: "the middle first")
bool lstm_fwd_persistent_leaves_room(const LstmLayerDev& L);
size_t lstm_bwd_ksplit_px_floats(const LstmLayerDev& L);
// What runs a layer's time recurrence.  ONE selection function per pass (lstm_fwd_plan / lstm_bwd_plan, lstm_persistent.hip) decides
// the instantiation; the launchers execute the plan, and everything else that has to know -- the per-minibatch side-stream rule of
// Net::backpropagate, the exchange-schedule rule of the data-parallel path (comm.cpp), eesen_net_plan_string, the tests -- reads the
// SAME plan, so a rule can no longer drift away from what the launcher picks.
enum { kRecNone = 0, kRecFwdBf = 1, kRecFwdMux = 2, kRecFwdF32 = 3, kRecBwdQ4 = 4, kRecBwdKsplit = 5, kRecBwdKsplitMux = 6, kRecBwdGeneric = 7 };
struct RecPlan {
  int kind = kRecNone;        // kRecNone: no persistent tile fits this shape -- the one-launch-per-step kernels of lstm.hip
  char kernel[80] = "";       // the instantiation, e.g. "lstm_bwd_persistent_q4_kernel<8,4>"
  int seq_tile = 0;           // sequences per workgroup
  int units = 0;              // hidden units per workgroup
  int windows = 0;            // launches per layer pass (sequence windows, one after the other)
  int grid[3] = {0, 0, 0};    // (unit blocks, directions, sequence groups) of ONE launch; launched as a 1-D grid of their product
  int wgs = 0;                // workgroups per launch
  int wgs_per_cu = 0;         // ... per CU of this process's share of the device (EESEN_GPU_SHARE)
  int vgprs = 0, lds = 0;     // registers per lane / static LDS bytes of the instantiation (hipFuncGetAttributes = the code object's
                              // .vgpr_count / .group_segment_fixed_size: pinned by tests/test_kernel_resources.py); 0: not known
  int free_vgprs = -1;        // registers per SIMD lane this grid leaves on a CU it occupies: 512 - waves per SIMD x allocated (-1: not known)
  bool light = false;         // backward: the 4- / 8-sequence tiles, beside which a side-stream GEMM workgroup keeps a third of its rate
  // for the launcher
  int cpw = 0, stq = 0, chunk = 0, xchg = 0;
  const void* fn = nullptr;
};
RecPlan lstm_fwd_plan(const LstmLayerDev& L);
// assume_px: plan as if the caller will hand over the K-split kernels' exchange buffer (LstmLayerDev::PX) -- Net::backpropagate does
// whenever lstm_bwd_ksplit_px_floats asks for one; the launcher itself plans with the buffer it was really given
RecPlan lstm_bwd_plan(const LstmLayerDev& L, bool assume_px);
bool lstm_bwd_persistent(hipStream_t st, const LstmLayerDev& L, const float* dY, int lddy, float* DG, unsigned* cnt,
                         unsigned* err, int spin_limit, unsigned long long* trace = nullptr);
// bias_grad[ndir*4H] = column sums of DG; peep_grad[ndir][3][H] = the diag(D^T C) products of
// bilstm-parallel-layer.h:507-510 / :598-601.  ws: >= red_rows_ws(...) floats.
void lstm_bias_peep_grads(hipStream_t st, const LstmLayerDev& L, const float* DG, float* bias_grad, float* peep_grad,
                          float* ws, size_t ws_floats);
size_t lstm_bias_peep_ws_floats(int T, int S, int H, int ndir);

// out[c] = sum_r M[r][c]  (bias gradient of AffineTransform, affine-trans-layer.h:183)
void col_sums(hipStream_t st, const float* M, int rows, int cols, int ld, float* out, float* ws, size_t ws_floats);
size_t col_sums_ws_floats(int rows, int cols);

// ---------------------------------------------------------------------------------------- ctc.hip
// y = softmax(x) per row (Softmax::PropagateFnc, softmax-layer.h:44-47)
void softmax_rows(hipStream_t st, const float* x, int ldx, float* y, int ldy, int rows, int K);
// out = log(in) elementwise on a [rows x K] matrix (CuMatrixBase::ApplyLog, ctc-loss.cc:132-133)
void log_rows(hipStream_t st, const float* in, int ldi, float* out, int ldo, int rows, int K);
// alpha and beta lattice sweeps for all S sequences (2*S workgroups of ctc_sweep_waves(Lpad) wavefronts: one up to Lpad = 256).
// logp: [T*S x K] (ld), labx: [S x Lpad] expanded labels (-1 padded), lens/lablens: [S]
// alpha/beta: [S][T][Lpad] (utterance-major), pzx: [S].  Lpad in {64, 128, ..., 4096}.  waves: 0 = the default for Lpad.
void ctc_alpha_beta(hipStream_t st, const float* logp, int ld, int T, int S, int Lpad, const int* labx, const int* lens,
                    const int* lablens, float* alpha, float* beta, float* pzx, int waves = 0);
int ctc_sweep_waves(int Lpad, int waves = 0);
// diff[t*S+s][k] = y*rowsum(e) - gamma ... (error kernel + softmax Jacobian, ctc-loss.cc:156-168)
// labx [S x Lpad]: the expanded labels (blank 0 at even positions), lablens [S] = 2 U_s + 1
void ctc_error_diff(hipStream_t st, const float* probs, int ld, int T, int S, int K, int Lpad, int Lmax, const int* lens,
                    const int* lablens, const int* labx, const float* alpha, const float* beta,
                    const float* pzx, float* diff, int ldd);
// in place: m = (apply_log ? log m : m) - prior_scale * log_prior[col]; log_prior may be null (net-output-extract.cc:103-112)
void log_sub_prior(hipStream_t st, float* m, int ld, int rows, int K, bool apply_log, const float* log_prior, float prior_scale);
// ids[r] = argmax_k m[r][k], first maximum wins (CuMatrixBase::FindRowMaxId, cuda-matrix.cc:1038-1095)
void row_argmax(hipStream_t st, const float* m, int ld, int rows, int K, int* ids);

// ---------------------------------------------------------------------------------------- optim.hip
// corr = mmt*corr + fresh; clip to +-max_grad when max_grad > 0; param -= lr_coef*corr
// skip (may be null): device word; when non-zero at execution time the update is a no-op (see optim.hip)
// live (may be null): device float; when ZERO at execution time the update is a no-op (data-parallel closing round, comm.cpp)
void sgd_update(hipStream_t st, float* param, float* corr, const float* fresh, long n, float mmt, float lr_coef,
                float max_grad, const unsigned* skip = nullptr, const float* live = nullptr);
// p[0..n) = 0 when *word != 0 at execution time (the recurrence kernels' error word): see optim.hip / comm.cpp
void zero_if_set(hipStream_t st, float* p, long n, const unsigned* word);
// Adagrad (rmsprop = false) / RMSProp update of one flat parameter block, see optim.hip
void adaptive_update(hipStream_t st, float* param, float* corr, const float* fresh, float* accu, long n, float mmt, float lr,
                     float max_grad, float eps, float rho, float one_minus_rho, bool rmsprop, const unsigned* skip = nullptr,
                     const float* live = nullptr);
// Moments of a [rows x cols] region (leading dimension ld) of a flat buffer, as MomentStatistics prints them
// (/root/reference/src/net/utils-functions.h:50-82): out6 = {min, max, mean, variance, skewness, kurtosis}, doubles on the
// device; two passes (min / max / sum, then the central power sums), fp64 accumulation.  ws: >= 8 doubles of scratch.
void tensor_moments(hipStream_t st, const float* base, long rows, int cols, long ld, double* out6, double* ws,
                    int nb = 0, int hf = 0, int hi = 0);   // (nb, hf, hi): column c of the tensor lives at (c / hf) * hi + c % hf of its row (0: at c)
// dst[c][r] = src[r][c]  (rows x cols -> cols x rows), dense
void transpose2d(hipStream_t st, const float* src, int rows, int cols, float* dst);
// dst[r][0..cols) = src[r][0..cols) with different leading dimensions
void copy2d(hipStream_t st, const float* src, int lds, float* dst, int ldd, int rows, int cols);

}  // namespace eesen
