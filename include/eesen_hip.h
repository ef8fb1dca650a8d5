/* eesen_hip.h -- C-ABI of libeesen_hip.so: the MI355X (gfx950) implementation of the hot path of
 * Eesen's `train-ctc-parallel` (peephole (Bi-)LSTM forward/backward, affine, softmax, CTC
 * forward-backward, SGD update) behind the reference's net/netbin operator interface.
 *
 * Plain C: opaque handles, plain pointers and sizes, int status codes.  No torch / C++ types.
 * Every entry point returns EESEN_OK (0) or a negative status; eesen_last_error() gives the message
 * of the last failure on the calling thread (the reference throws std::runtime_error from KALDI_ERR,
 * /root/reference/src/base/kaldi-error.cc:168-182; no exception crosses this boundary).
 *
 * Threading: one handle per GPU, not thread-safe per handle (the reference is single-threaded with a
 * process-global CuDevice, src/gpucompute/cuda-device.cc:846).  All work of a handle is enqueued on the
 * handle's HIP stream; entry points that return host data synchronise that stream, the others do not.
 *
 * Memory kinds: pointers marked `dev` are device pointers on the handle's GPU, `host` are host
 * pointers.  Matrices are row-major fp32 with an explicit leading dimension `ld` (elements), exactly
 * the reference's MatrixDim {rows, cols, stride} (src/gpucompute/cuda-matrixdim.h:52-56).
 * Utterance batches are time-major interleaved: row t*S + s = frame t of sequence s
 * (src/netbin/train-ctc-parallel.cc:187-193).
 */
#ifndef EESEN_HIP_H_
#define EESEN_HIP_H_

#ifdef __cplusplus
extern "C" {
#endif

#define EESEN_OK 0
#define EESEN_ERR_INVALID -1   /* bad argument / unsupported model feature (e.g. dropout tokens)   */
#define EESEN_ERR_HIP -2       /* a HIP runtime call or kernel launch failed                        */
#define EESEN_ERR_STATE -3     /* call sequence violated (e.g. Backpropagate before Propagate)      */
#define EESEN_ERR_IO -4        /* model file could not be read / written / parsed                   */
#define EESEN_ERR_COMM -5      /* the data-parallel exchange was aborted: a peer died or stalled    */

/* Layer kinds: the markers of src/net/layer.cc:37-46 that are on the hot path. */
#define EESEN_LAYER_AFFINE 1          /* <AffineTransform>  src/net/affine-trans-layer.h           */
#define EESEN_LAYER_SOFTMAX 2         /* <Softmax>          src/net/softmax-layer.h                */
#define EESEN_LAYER_LSTM_PARALLEL 3   /* <LstmParallel>     src/net/lstm-parallel-layer.h          */
#define EESEN_LAYER_BILSTM_PARALLEL 4 /* <BiLstmParallel>   src/net/bilstm-parallel-layer.h        */
#define EESEN_LAYER_SIGMOID 5         /* <Sigmoid>          src/net/sigmoid-layer.h (no parameters) */
#define EESEN_LAYER_TANH 6            /* <Tanh>             src/net/tanh-layer.h    (no parameters) */

typedef struct eesen_net eesen_net_t; /* replaces eesen::Net, src/net/net.h:37-175           */
typedef struct eesen_ctc eesen_ctc_t; /* replaces eesen::Ctc, src/net/ctc-loss.h:31-90       */

/* ---- library / device ------------------------------------------------------------------------ */
const char* eesen_last_error(void);
const char* eesen_version(void);
/* replaces CuDevice::Instantiate().SelectGpuId (src/gpucompute/cuda-device.cc:73-140): number of
 * visible HIP devices (0 is not an error here; creating a handle then fails loudly). */
int eesen_device_count(int* count);

/* Arithmetic of the dense contractions outside the time recurrence (what the reference sends to cublasSgemm through
 * CuMatrixBase::AddMatMat, src/gpucompute/cuda-matrix.cc:604-639), process-wide:
 *   0  f32-input MFMA (v_mfma_f32_32x32x2_f32): bit-for-bit a k-ordered fp32 fmaf chain, at the f32 vector rate;
 *   1  3-way bf16 split: every fp32 operand is EXACTLY hi + mid + lo (three bf16), six of the nine cross products run on
 *      v_mfma_f32_32x32x16_bf16 with fp32 accumulation; error <= 2^-23 |a*b| per product, i.e. fp32-GEMM class, at 2.67x the
 *      f32 matrix rate (gfx950 runs f32 MFMA at 1/16 of the bf16 rate and has no TF32 form);
 *   2  two fp16 planes (round 6, the default): every fp32 operand is hi + lo (fp16, round to nearest at both levels: equal to
 *      the value to within 2^-22), each row of op(A) / column of op(B) scaled by the exact power of two its own largest magnitude
 *      asks for (measured on the device), three of the four cross products on v_mfma_f32_32x32x16_f16 with fp32 accumulation;
 *      error <= 3 * 2^-22 |a*b| per product (the "3xTF32" arithmetic; measured against fp64 equal to modes 0 and 1,
 *      tests/test_gpu_gemm.py) at 1.58x the rate of mode 1;
 *  -1  follow the environment (EESEN_GEMM_MODE=f32|split|half; half when unset), the initial state. */
int eesen_set_gemm_mode(int mode);
int eesen_get_gemm_mode(int* mode);

/* ---- Net: construction, model I/O ------------------------------------------------------------ */
/* New empty net on `device`.  `stream` is a hipStream_t to enqueue on (e.g. the caller framework's
 * current stream) or NULL for the device's default stream.  A Net and the Ctc it feeds must share a
 * stream (the reference has one implicit stream for everything).  A handle is not thread-safe, and the recurrence
 * kernels are cooperative launches that want every CU of the device: two Nets on the same device must not run
 * Propagate / Backpropagate CONCURRENTLY on different streams (each would hold half the CUs and wait for the rest until
 * its spin bound raises EESEN_ERR_HIP); on one stream, or one after the other, any number of handles coexist. */
int eesen_net_create(int device, void* stream, eesen_net_t** out);
int eesen_net_destroy(eesen_net_t* net);
/* Net::AppendLayer (src/net/net.cc:197-205) with the layer header of src/net/layer.cc:138-222:
 * in_dim = <InputDim>, out_dim = <CellDim> (LSTM kinds: 2H for BiLstm, H for Lstm) or <OutputDim>. */
int eesen_net_add_layer(eesen_net_t* net, int kind, int in_dim, int out_dim,
                        float learn_rate_coef, float max_grad);
/* Allocates parameters (zero) + optimiser state; must follow the last add_layer. */
int eesen_net_finalize(eesen_net_t* net);
/* Net::Read (src/net/net.cc:279-309): parse a Kaldi-stream <Nnet> file (text or \0B binary), build
 * the layers and upload the weights (and the Adagrad/RMSProp accumulators when the file carries them; the nine dropout
 * options of a BiLstm layer are kept, see the dropout section below).  Fails (EESEN_ERR_INVALID) on layer kinds outside
 * the list above.  Resets learn_rate to 0 as the reference does (net.cc:294). */
int eesen_net_read(eesen_net_t* net, const char* path);
/* Net::Write (src/net/net.cc:325-334), same byte format as the reference for these layer kinds. */
int eesen_net_write(eesen_net_t* net, const char* path, int binary);

int eesen_net_num_layers(eesen_net_t* net, int* n);
int eesen_net_layer_info(eesen_net_t* net, int idx, int* kind, int* in_dim, int* out_dim,
                         float* learn_rate_coef, float* max_grad);
/* Layer::TypeToMarker (src/net/layer.cc:68-79): the marker token of layer idx as its model file carries it
 * ("<BiLstmParallel>", "<BiLstm>", "<AffineTransform>", ...), NUL-terminated into buf[cap]. */
int eesen_net_layer_marker(eesen_net_t* net, int idx, char* buf, int cap);
/* What Net::Info / Net::InfoGradient print per tensor (src/net/net.cc:336-385 through MomentStatistics,
 * src/net/utils-functions.h:50-82): {min, max, mean, variance, skewness, kurtosis} of each tensor of `layer`, in the order
 * of the layer's own Info() (bilstm-layer.h:496-560: W_x, W_m, bias, p_i, p_f, p_o forward, then backward; lstm-layer.h:175-196;
 * affine-trans-layer.h:145-159: linearity, bias).  which: 0 = the parameters (Info), 1 = the momentum buffers *_corr_
 * (InfoGradient), 2 = the Adagrad / RMSProp accumulators *_corr_accu.  Computed on the device (two reduction passes per
 * tensor, fp64 accumulation).  out6_host: [cap_tensors x 6] doubles or NULL (count only); *n_tensors = tensors of the layer
 * (0 for layers without parameters).  Synchronises. */
int eesen_net_tensor_moments(eesen_net_t* net, int which, int layer, double* out6_host, int cap_tensors, int* n_tensors);
int eesen_net_input_dim(eesen_net_t* net, int* dim);   /* Net::InputDim  net.cc:139-142 */
int eesen_net_output_dim(eesen_net_t* net, int* dim);  /* Net::OutputDim net.cc:134-137 */
int eesen_net_num_params(eesen_net_t* net, long* n);   /* Net::NumParams net.cc:163-172 */
/* Net::GetParams / SetParams order (src/net/net.cc:181-195): per trainable layer its tensors in
 * model-file order, each row-major and dense.  host pointers, n = num_params. */
int eesen_net_get_params(eesen_net_t* net, float* host_flat, long n);
int eesen_net_set_params(eesen_net_t* net, const float* host_flat, long n);

/* ---- Net: training options (src/net/train-opts.h:29-62, net.h:147-161) ------------------------- */
int eesen_net_set_train_options(eesen_net_t* net, float learn_rate, float momentum);
/* Net::SetUpdateAlgorithm (src/net/net.cc:481-497): "SGD" (default), "Adagrad", "RMSProp" (trainable-layer.h:65-114). */
int eesen_net_set_update_algorithm(eesen_net_t* net, const char* name);
/* NetTrainOptions::adagrad_epsilon (1e-6) and rmsprop_rho (0.9) (train-opts.h:33-42). */
int eesen_net_set_adaptive_options(eesen_net_t* net, float adagrad_epsilon, float rmsprop_rho);
/* The squared-gradient accumulators (*_corr_accu, bilstm-layer.h:1098-1124) in Net::GetParams order; they are what
 * <BiLstmAccus>/<LstmAccus>/<AffineAccus> carry in a model file. Zero until an adaptive rule ran or a file held them. */
int eesen_net_get_accumulators(eesen_net_t* net, float* host_flat, long n);
int eesen_net_set_accumulators(eesen_net_t* net, const float* host_flat, long n);
/* Net::SetSeqLengths (net.h:157): frame count of each of the S parallel sequences. host pointer. */
int eesen_net_set_seq_lengths(eesen_net_t* net, const int* lens, int S);

/* ---- Net: forward / backward / update -------------------------------------------------------- */
/* Net::Propagate (src/net/net.cc:67-86).  in: [rows x input_dim], rows = T*S, `in_is_device` selects
 * the memory kind (the reference trainer hands a host Matrix that the CuMatrix ctor uploads,
 * train-ctc-parallel.cc:198).  The output stays on the device and is owned by the handle:
 * *out_dev -> [rows x *out_cols] fp32 with leading dimension *out_ld, valid until the next Propagate.
 * Rows t >= len[s] hold unspecified values (the reference computes on padding too; nothing reads them). */
int eesen_net_propagate(eesen_net_t* net, const float* in, int rows, int in_ld, int in_is_device,
                        const float** out_dev, int* out_cols, int* out_ld);
/* Copy the last Propagate output to the host ([rows x output_dim], dense). Synchronises. */
int eesen_net_get_output(eesen_net_t* net, float* host_out, long n);

/* Net::Backpropagate minus Update (src/net/net.cc:88-108, with :101-104 split out so that a
 * data-parallel gradient exchange can run between the two).  out_diff_dev: [rows x output_dim]
 * device matrix, d(-ln p)/d(logits) as Ctc::EvalParallel returns it.  After the call every
 * parameter has its FRESH gradient (sum over frames, momentum not yet folded in) in the gradient
 * buffer.  in_diff_dev: optional [rows x input_dim] device matrix (NULL in the trainer,
 * train-ctc-parallel.cc:207). */
int eesen_net_backpropagate(eesen_net_t* net, const float* out_diff_dev, int out_diff_ld,
                            float* in_diff_dev, int in_diff_ld);
/* The contiguous device buffer of fresh gradients of ALL parameters (n floats, library-internal
 * tensor layout, identical on every rank): the payload of the data-parallel all-reduce(SUM) that
 * replaces comm_avg_weights (src/net/communicator.h:39-119). */
int eesen_net_grad_buffer(eesen_net_t* net, float** dev_ptr, long* n);
/* Fresh gradients in Net::GetParams order on the host (debug / parity accessor; the reference keeps
 * *_corr_ protected, src/net/bilstm-layer.h:1076-1096). Synchronises. */
int eesen_net_get_grads(eesen_net_t* net, float* host_flat, long n);
/* TrainableLayer::Update for every layer, SGD rule (src/net/bilstm-layer.h:846-883,
 * affine-trans-layer.h:174-195): corr = momentum*corr + fresh; clip corr to +-max_grad if
 * max_grad > 0; param -= learn_rate*learn_rate_coef*corr. */
int eesen_net_update(eesen_net_t* net);
/* BASELINE config 4's "bf16 forward / fp32 CTC accumulate" variant (no counterpart in the reference, whose BaseFloat is float,
 * src/base/kaldi-types.h:26-30).  mode 1: the whole forward pass multiplies in bf16 with fp32 accumulation --
 *   (a) the GEMMs of Propagate (input->gates, affine / projection) round both operands to nearest-even bf16 and run ONE
 *       v_mfma_f32_32x32x16_bf16 product;
 *   (b) the forward TIME RECURRENCE (the loop of src/net/bilstm-parallel-layer.h:112-149,165-204) keeps W_m as TWO bf16 planes
 *       (hi + lo: 17 significant bits), rounds m_t to ONE bf16 plane, at the cell write into the kernel's exchange buffer, and
 *       forms m_{t-1} W_m^T on
 *       v_mfma_f32_16x16x32_bf16 (layers with H a multiple of 256 up to 1024 cells per direction, no recurrent dropout; other
 *       layers keep the fp32 recurrence).  Gate pre-activations, cell state, activations, and everything stored for the
 *       backward pass stay fp32;
 * softmax, CTC, the whole backward pass and the update stay fp32.  mode 2: (a) only.
 * (Up to round 3 of this library value 1 meant (a) only; since round 4 it also selects (b), and (a) alone is value 2.)
 * mode 0 (default): fp32 everywhere.  Distances: tests/test_gpu_gemm.py::test_bf16_forward_variant against this library's fp32
 * path; against THE REFERENCE at BASELINE config 4's full size: tests/test_gpu_reference_fullsize.py, profiles/parity_cfg4.json. */
int eesen_net_set_forward_precision(eesen_net_t* net, int mode);
/* Debug / test accessor: how many LSTM layers of the last Propagate ran their recurrence on the bf16 kernel of (b). */
int eesen_net_bf16_recurrence_layers(eesen_net_t* net, int* layers);
/* Debug / test accessor: out4 = {LSTM layers, layers whose forward time loop ran as ONE cooperative launch per sequence
 * window (lstm_persistent.hip), layers whose backward time loop did} for the last Propagate / Backpropagate (the rest
 * took the one-launch-per-step kernels), and the number of recoveries so far.  A recovery: a cooperative recurrence
 * kernel whose bounded spin gave up (its workgroups were not all resident -- GPU shared or preempted) raises a device
 * word; the update kernels of that step see it and leave the model untouched, and the handle continues on the per-step
 * kernels with a WARNING on stderr.  With a communicator attached the other ranks do apply their step: the failed rank's
 * gradients then enter the all-reduce as zeros (decided on the device) and it applies the same summed update, so the ranks' models
 * stay identical and the run only loses that rank's share of the minibatches in flight. */
int eesen_net_recurrence_info(eesen_net_t* net, int* out4);
/* What WILL run the current minibatch shape (call after eesen_net_set_seq_lengths), as one JSON object: per LSTM layer the forward
 * and backward recurrence plans -- kernel instantiation, sequences / units per workgroup, grid, launches per pass, registers per
 * lane and LDS bytes of the instantiation, registers per SIMD lane its grid leaves free on a CU -- and the schedule decisions that
 * follow from them: weight-gradient GEMMs on the side stream or not, the exchange schedule with a communicator attached.  The
 * launchers execute exactly these plans (one selection function per pass, lstm_persistent.hip: lstm_fwd_plan / lstm_bwd_plan).
 * Diagnostic; the reference has no counterpart (its kernels are chosen at compile time, src/gpucompute/cuda-kernels.cu). */
int eesen_net_plan_string(eesen_net_t* net, char* json, int cap);
/* Test hook: stores `value` into that device word on the handle's stream, as a kernel that gave up would (1: a recurrence
 * kernel's bounded spin, 2: the side stream's wait for a forward milestone), so that the recovery paths can be exercised. */
int eesen_net_debug_set_error_word(eesen_net_t* net, unsigned value);
/* Block until everything enqueued on the handle's stream has finished. */
int eesen_net_synchronize(eesen_net_t* net);
/* Seconds spent (HIP events on the handle's stream) in the phases of the last step, for bench.py:
 * out[0]=input GEMMs, [1]=recurrence fwd, [2]=affine+softmax, [3]=recurrence bwd, [4]=gradient GEMMs
 * and reductions, [5]=update.  Enabled by eesen_net_set_profiling(net, 1); with 2 the timers ACCUMULATE over all steps
 * since the last eesen_net_get_phase_times, so a multi-step region needs no host synchronisation per step. */
int eesen_net_set_profiling(eesen_net_t* net, int on);
int eesen_net_get_phase_times(eesen_net_t* net, float* out6);
/* The individual timed spans behind those sums, in the order they were recorded: phases[i] (index as above) and seconds[i]
 * of span i, up to `cap`; *n = spans recorded since the last eesen_net_get_phase_times.  One span per kernel group of a
 * layer (e.g. one per layer and step for the recurrences, top layer first in the backward pass), so a caller can separate
 * the launch that runs alone on the chip from those that share it with side-stream work.  Call BEFORE get_phase_times.
 * With a communicator attached two more span kinds appear here (they are not part of out6): phase 6 = one gradient bucket's
 * all-reduce on the communicator's stream (from "bucket ready and stream free" to the end of the collective; in the order of
 * eesen_net_bucket_order), phase 7 = the time the compute stream waited for one bucket inside eesen_net_update -- the part of
 * the exchange the lower layers' backward pass did not hide. */
int eesen_net_get_phase_spans(eesen_net_t* net, int* phases, float* seconds, int cap, int* n);

/* ---- data-parallel exchange: one process per GPU, RCCL over xGMI ------------------------------------------------------
 * Replaces the reference's multi-job mode (--num-jobs / --job-id / --utts-per-avg, src/netbin/train-ctc-parallel.cc:208-235):
 * comm_avg_weights (src/net/communicator.h:39-119) averages whole MODELS through files every few hundred utterances and
 * comm_touch_done (:121-170) merges the error counts through "done" files.  Here every rank runs the same Net on its own
 * utterances and the FRESH gradients (sums over frames) are summed over the ranks every minibatch, so that N ranks x S
 * utterances equal one process with --num-sequence = N*S (momentum and clipping act on the summed gradient).
 *
 * Rendezvous: rank 0 draws the 128-byte RCCL unique id (eesen_comm_get_unique_id) and hands it to the other ranks --
 * through any channel the host has (eesen_comm_create), or through the library's own TCP hand-out
 * (eesen_comm_create_tcp: rank 0 listens on addr:port, the others connect, retrying until `timeout_s`).
 * eesen_comm_exchange is that hand-out on its own (any <= 4 KB blob; needs no GPU). */
typedef struct eesen_comm eesen_comm_t;
int eesen_comm_get_unique_id(char* id128);
int eesen_comm_exchange(const char* addr, int port, int rank, int world, char* buf, int nbytes, int timeout_s);
int eesen_comm_create(int device, const char* id128, int rank, int world, eesen_comm_t** out);
int eesen_comm_create_tcp(int device, const char* addr, int port, int rank, int world, int timeout_s, eesen_comm_t** out);
int eesen_comm_destroy(eesen_comm_t* comm);
int eesen_comm_info(eesen_comm_t* comm, int* rank, int* world);
/* What the communicator really is, as one JSON object -- COLLECTIVE: every rank calls it (it gathers one word per rank).
 * {library: path the dynamic linker resolved, rccl_version: ncclGetVersion, stand_in: the tests' stand-in and not RCCL,
 *  rank / world: as created, world_seen / rank_seen / device_seen: ncclCommCount / ncclCommUserRank / ncclCommCuDevice,
 *  devices: [hosthash/PCI bus id of every rank's GPU], distinct_devices, ranks_share_devices}.
 * The reference's multi-job mode has no such thing (its jobs only meet through files, src/net/communicator.h:39-170); a
 * collective-based exchange owes its operator the answer to "did N ranks on N distinct GPUs really meet". */
int eesen_comm_describe(eesen_comm_t* comm, char* json, int cap);
/* Sum (op 0) or max (op 1) of n <= 64 host doubles over the ranks, in place; blocks.  Carries the statistics the
 * reference merges through done-files (sum ln p, error / reference tokens, frames), "does any rank still have a
 * minibatch", and doubles as a barrier. */
int eesen_comm_allreduce_host(eesen_comm_t* comm, double* values, int n, int op);
/* Attach a communicator to a Net (NULL detaches).  From then on eesen_net_backpropagate sums every layer's fresh
 * gradients over the ranks -- one all-reduce(SUM, fp32) per trainable layer, issued on the communicator's own stream as
 * soon as that layer's weight-gradient kernels are enqueued, i.e. where the reference updates the layer
 * (src/net/net.cc:98-104), so the exchange of the upper layers runs under the backward pass of the lower ones -- and
 * eesen_net_update waits bucket by bucket.  Every rank must call the same sequence of backpropagate / update. */
int eesen_net_set_comm(eesen_net_t* net, eesen_comm_t* comm);
/* The same exchange as ONE all-reduce of the whole gradient buffer on the Net's stream, for hosts that keep the
 * communicator detached: call between eesen_net_backpropagate and eesen_net_update. */
int eesen_net_allreduce_grads(eesen_net_t* net, eesen_comm_t* comm);
/* Jobs with different numbers of minibatches (the reference's sub-jobs simply stop when job 1 has finished,
 * src/net/communicator.h:104-112): a rank that has run out of minibatches while others have not keeps stepping with
 *   eesen_net_backpropagate_zero (zero gradient, the attached communicator's per-layer all-reduces in the order a real
 *   Backpropagate issues them) + eesen_net_update + eesen_net_live_ranks
 * until live_ranks reports 0.  One float rides with the top layer's gradient bucket: 1 from every rank whose step was a real
 * eesen_net_backpropagate, 0 from the others; *live is its sum over the ranks for the last step issued (blocks until that
 * bucket has arrived).  A step in which NO rank was live -- the closing round, which all ranks take together -- leaves the
 * model untouched (the update kernels read the word on the device), so N ranks with uneven shards still equal one process
 * on the union of their minibatches.  Ranks that still train never need to ask. */
int eesen_net_backpropagate_zero(eesen_net_t* net);
int eesen_net_live_ranks(eesen_net_t* net, int* live);
/* A peer that dies leaves the others' collectives spinning.  Every communicator runs a watchdog: a collective that has not
 * completed EESEN_COMM_TIMEOUT_S (default 600) seconds after it was issued is aborted (ncclCommAbort), every blocked wait
 * returns, and every later call on the communicator or a net attached to it fails with EESEN_ERR_COMM. */
/* Debug / test accessor: the layer indices whose buckets the last Backpropagate issued, in issue order. */
int eesen_net_bucket_order(eesen_net_t* net, int* layers_out, int cap, int* n);
/* hipDeviceSynchronize of `device` (bench.py brackets its timed region with it). */
int eesen_device_synchronize(int device);

/* ---- Ctc (src/net/ctc-loss.h:31-90) ------------------------------------------------------------ */
int eesen_ctc_create(int device, void* stream, eesen_ctc_t** out);
int eesen_ctc_destroy(eesen_ctc_t* ctc);
/* Ctc::EvalParallel (src/net/ctc-loss.cc:101-194).  net_out_dev: [T*S x K] softmax outputs (device),
 * frame_num_utt: host int[S]; labels in CSR form on the host: label_off int[S+1], label_ids (no
 * blanks, blank id is 0).  Writes diff_dev [T*S x K] (device, ld given) = d(-ln p)/d(logits), zero
 * on rows t >= frame_num_utt[s]; pzx_host (may be NULL: then nothing waits for the device and ln p joins the
 * objective sum once it has arrived): ln p(z|x) per sequence.  Accumulates the
 * objective / frame counters reported by eesen_ctc_report.  Every sequence needs >= 1 label and a
 * feasible alignment is the caller's business (ln p ~ -1e30 otherwise, as in the reference).
 * Limits (EESEN_ERR_INVALID beyond them; the reference has none): at most 2047 labels per sequence -- expanded label sequences
 * of up to 4096 lattice positions -- and at most 20480 classes (5120 before round 6's closing change). */
int eesen_ctc_eval_parallel(eesen_ctc_t* ctc, const int* frame_num_utt, int S, const float* net_out_dev,
                            int rows, int K, int ld, const int* label_ids, const int* label_off,
                            float* diff_dev, int diff_ld, float* pzx_host);
/* Ctc::ErrorRateMSeq (ctc-loss.cc:235-298): greedy decode (argmax on the device, collapse repeats,
 * drop blanks) and Levenshtein distance against the references (host).  Accumulates error / ref
 * token counts; also returns this call's counts.  With num_err == num_ref == NULL the call only ENQUEUES the argmax
 * and the copy of the ids: the host part runs at the next call / eesen_ctc_stats, under the device's backward pass
 * (the reference's call returns void and only accumulates). */
int eesen_ctc_error_rate_mseq(eesen_ctc_t* ctc, const int* frame_num_utt, int S, const float* net_out_dev,
                              int rows, int K, int ld, const int* label_ids, const int* label_off,
                              int* num_err, int* num_ref);
/* --sequence-out-file of the trainer (src/netbin/train-ctc-parallel.cc:53-54,134-137; written by Ctc::ErrorRateMSeq,
 * ctc-loss.cc:247-250,282-291): every later eesen_ctc_error_rate_mseq appends one line per utterance,
 * `utt | <label> <frame> <probability> | ...` for the greedy-decoded sequence.  The file is removed first, as the
 * reference's trainer does; NULL or "" switches the output off. */
int eesen_ctc_set_sequence_out_file(eesen_ctc_t* ctc, const char* path);
/* Guard the statistics against a timed-out forward pass.  A cooperative recurrence kernel whose bounded spin gave up leaves
 * garbage activations and raises a device word (eesen_net_recurrence_info); the Net recovers on its own, but a minibatch the
 * host had already handed to the Ctc would fold a garbage ln p and garbage decodes into the objective and TOKEN_ACCURACY.  With
 * a guard the word's value travels back with every minibatch's results, and a minibatch computed while it was set is dropped
 * from ALL running totals (eesen_ctc_dropped counts them; one WARNING on stderr) and the ln p values eesen_ctc_eval_parallel hands
 * back for it read NaN.  The Ctc and the Net must share device and stream (the word is read in stream order); either may be
 * destroyed first.  net == NULL removes the guard. */
int eesen_ctc_set_guard(eesen_ctc_t* ctc, eesen_net_t* net);
int eesen_ctc_dropped(eesen_ctc_t* ctc, long* minibatches);
/* Running totals: Ctc::NumErrorTokens/NumRefTokens (ctc-loss.h:58-59) and the sums behind
 * Ctc::Report (ctc-loss.cc:300-308): obj = sum of ln p, sequences, frames. */
int eesen_ctc_stats(eesen_ctc_t* ctc, double* obj_sum, long* sequences, long* frames, long* err_tokens,
                    long* ref_tokens);
/* Debug / parity accessor: alpha and beta of the last EvalParallel in the reference's layout
 * [T*S x L'] (row t*S+s, L' = 2*max_U+1), host pointers, either may be NULL. Cells the reference
 * leaves at -1e30 (padding) read -1e30. */
int eesen_ctc_get_alpha_beta(eesen_ctc_t* ctc, float* alpha_host, float* beta_host, int* Lprime);
/* seconds of the last EvalParallel's device work (HIP events): out[0]=log, [1]=alpha/beta sweep, [2]=error+jacobian.
 * eesen_ctc_set_profiling(ctc, 2): the spans of ALL calls since the last read are summed instead, so that a timed
 * multi-step region needs no host synchronisation per step (0 = back to last-call timing). */
int eesen_ctc_set_profiling(eesen_ctc_t* ctc, int mode);
int eesen_ctc_get_phase_times(eesen_ctc_t* ctc, float* out3);

/* ---- dropout variants of BiLstm(Parallel) (src/net/bilstm-parallel-layer.h:46-94,209-377,604-879) ------
 * Options travel in the model file (nine tokens, src/net/bilstm-layer.h:331-373) and through set/get_layer_dropout, in token
 * order: {ForwardDropoutFactor, ForwardTimeStepDropout, ForwardSequenceDropout, RecurrentTimeStepDropout,
 * RecurrentSequenceDropout, RNNDrop, NoMemLossDropout, RecurrentDropoutFactor, TwiddleForward} (booleans as 0 / 1).
 * Net::SetTrainMode / SetTestMode (src/net/net.cc:396-412): dropout is applied in train mode only (the default).
 * Masks (values 0 or 1/(1-p)) are drawn on the device from (seed, draw counter, element) -- the reference draws them with the
 * host RNG and copies them over (:50-62).  For parity tests the masks of the NEXT eesen_net_propagate can be supplied:
 *   fwd_mask  [T*S x 2H] or NULL;  rec_mask [rec_rows x 2H] with rec_rows = (T+2)*S (time-step masks, row t*S+s as in the
 *   reference's buffers) or S (sequence masks), columns = forward-direction H then backward-direction H, or NULL;
 *   twiddle_coin: 0 / 1 = value of the TwiddleForward coin, -1 = draw it.
 * H is the cell count per direction of the MODEL FILE's layer (eesen_net_layer_info's output dimension / ndir): when the library
 * pads a cell count that is not a multiple of 4 inside, these two accessors gather / scatter the file's columns. */
int eesen_net_set_train_mode(eesen_net_t* net, int train);
int eesen_net_set_dropout_seed(eesen_net_t* net, unsigned long long seed);
int eesen_net_set_layer_dropout(eesen_net_t* net, int layer, const float* nine);
int eesen_net_get_layer_dropout(eesen_net_t* net, int layer, float* nine);
int eesen_net_set_dropout_masks(eesen_net_t* net, int layer, const float* fwd_mask_host, long fwd_floats,
                                const float* rec_mask_host, int rec_rows, long rec_floats, int twiddle_coin);
/* masks used by the LAST propagate: fwd_mask_host [T*S x 2H], rec_mask_host [(T+2)*S x 2H] (either may be NULL);
 * info4 = {forward dropout applied, recurrent mode (0 none, 1 no-memory-loss, 2 RNNDrop), twiddle coin, 2H}. */
int eesen_net_get_dropout_masks(eesen_net_t* net, int layer, float* fwd_mask_host, float* rec_mask_host, int* info4);

/* ---- minibatch assembly on the device (row a1) ---------------------------------------------------
 * Replaces the host-side padding + interleave + blocking H2D of src/netbin/train-ctc-parallel.cc:186-195: the S utterance
 * matrices are packed back to back into a pinned staging slot (no padding crosses PCIe), copied on the feeder's own
 * stream, and a kernel writes the zero-padded time-major matrix (row t*S + s, leading dimension ld = D rounded up to 4)
 * in HBM.  `slots` staging slots rotate: submit() of batch n+1 overlaps the training step of batch n.
 *   submit : host pointers utts[s] -> [frames[s] x D] row-major, row stride strides[s] floats (strides == NULL: D).
 *            Asynchronous; blocks only if the slot being reused is still in flight / not yet released.
 *   acquire: makes `compute_stream` wait (on the device) for the slot; returns the assembled device matrix.  The
 *            pointer stays valid until the slot is submitted again.
 *   release: call after the last consumer of the slot has been ENQUEUED on compute_stream (eesen_net_propagate copies
 *            its input, so right after it). */
typedef struct eesen_feeder eesen_feeder_t;
int eesen_feeder_create(int device, void* compute_stream, int slots, eesen_feeder_t** out);
int eesen_feeder_destroy(eesen_feeder_t* f);
int eesen_feeder_submit(eesen_feeder_t* f, const float* const* utts, const int* frames, const int* strides, int S, int D,
                        int* slot);
int eesen_feeder_acquire(eesen_feeder_t* f, int slot, float** feats_dev, int* T, int* S, int* ld);
int eesen_feeder_release(eesen_feeder_t* f, int slot);

/* ---- feature front end fused into the batch assembly (the step in front of the path) --------
 * The recipes hand the trainer its features through a pipe of host filters (asr_egs/wsj/steps/train_ctc_parallel.sh:95-110,
 * decode_ctc_lat.sh:92-95, librispeech/steps/train_ctc_parallel_mult.sh:110-133):
 *   apply-cmvn [--norm-vars] --utt2spk=... scp:cmvn.scp scp:train.scp ark:- | splice-feats --left-context=L --right-context=R
 *   ark:- ark:- | subsample-feats --n=N --offset=O ark:- ark:- | add-deltas ark:- ark:- |
 * With a pipeline set, submit_raw takes the RAW utterance matrices (what scp:train.scp points at) and the stages run on
 * the device, in the given order, between the PCIe copy and the interleave; acquire() then returns [T*S x D_out] with
 * T = the longest utterance AFTER the pipeline.  Stage arithmetic is the reference tool's, operation for operation:
 *   EESEN_FEAT_CMVN       src/featbin/apply-cmvn.cc, ApplyCmvn src/feat/cmvn.cc:110-117: x * scale[d] + offset[d]; the two
 *                         vectors per utterance come with submit_raw (eesen_cmvn_norm computes them from the stats matrix)
 *   EESEN_FEAT_SPLICE     src/featbin/splice-feats.cc, SpliceFrames src/feat/feature-functions.cc:391-412; a = left, b = right
 *   EESEN_FEAT_SUBSAMPLE  src/featbin/subsample-feats.cc:77-108; a = n (n < 0: repeat every frame -n times), b = offset
 *   EESEN_FEAT_DELTAS     src/featbin/add-deltas.cc, DeltaFeatures src/feat/feature-functions.cc:210-267; a = order, b = window
 * An utterance that the pipeline leaves without frames (subsampling offset beyond its end) contributes no rows, like the
 * reference tool, which writes no output for it; the caller drops it from the labels (eesen_feeder_pipeline_shape tells). */
#define EESEN_FEAT_CMVN 1
#define EESEN_FEAT_SPLICE 2
#define EESEN_FEAT_SUBSAMPLE 3
#define EESEN_FEAT_DELTAS 4
typedef struct eesen_feat_stage { int kind, a, b; } eesen_feat_stage_t;
int eesen_feeder_set_pipeline(eesen_feeder_t* f, const eesen_feat_stage_t* stages, int n_stages); /* 0 stages: plain submit */
/* feature dimension / frame count behind the pipeline for an utterance of [frames_in x D_in] (either output may be NULL) */
int eesen_feeder_pipeline_shape(eesen_feeder_t* f, int D_in, int frames_in, int* D_out, int* frames_out);
/* cmvn[s] -> 2 x Dc floats (offsets, then scales), Dc = the dimension in front of the CMVN stage; NULL without such a stage */
int eesen_feeder_submit_raw(eesen_feeder_t* f, const float* const* utts, const int* frames, const int* strides,
                            const float* const* cmvn, int S, int D_in, int* slot);
/* ApplyCmvn's normaliser (src/feat/cmvn.cc:78-108) on the host: stats = [rows x cols] doubles as apply-cmvn reads them
 * (row 0: sums and, last, the count; row 1: sums of squares), cols = dim + 1 -> offset_scale[2 x dim] */
int eesen_cmvn_norm(const double* stats, int rows, int cols, int norm_vars, float* offset_scale);

/* ---- raw device helpers for hosts that own no GPU allocator --------------------------------- */
int eesen_dev_alloc(int device, long bytes, void** dev_ptr);
int eesen_dev_free(int device, void* dev_ptr);
int eesen_dev_copy(int device, void* dst, const void* src, long bytes, int kind /*1 H2D, 2 D2H, 3 D2D*/);

/* ---- single-op entry points (kernel-level parity tests and microbenchmarks) ------------------- */
/* C[MxN] = alpha*op(A)*op(B) + beta*C (+ bias[n]); fp32 MFMA GEMM on device pointers.
 * a_kc: A stored [M x K] (k contiguous) if 1, [K x M] if 0.  b_kc: B stored [N x K] if 1, [K x N] if 0.
 * Replaces CuMatrixBase::AddMatMat (src/gpucompute/cuda-matrix.cc:604-639). */
int eesen_op_gemm(int device, void* stream, int a_kc, int b_kc, int M, int N, int K, float alpha,
                  const float* A, int lda, const float* B, int ldb, float beta, float* C, int ldc,
                  const float* bias);

/* The same GEMM ENQUEUED on `stream` and nothing else (no allocation, no synchronisation): `ws` / `ws_floats` is the caller's split-K
 * workspace (device memory, 0 floats = no split-K), `extra_lds_bytes` the unused dynamic LDS that caps the kernel's workgroups per CU --
 * how Net::Backpropagate runs its weight-gradient GEMMs beside a recurrence (scripts/corun_probe.py measures what that costs either side). */
int eesen_op_gemm_async(int device, void* stream, int a_kc, int b_kc, int M, int N, int K, const float* A, int lda, const float* B, int ldb,
                        float* C, int ldc, float* ws, long ws_floats, int extra_lds_bytes);

/* Post-processing of net-output-extract (src/netbin/net-output-extract.cc:103-112) on a device matrix, in place:
 * CuMatrixBase::ApplyLog when apply_log != 0, then ClassPrior::SubtractOnLogpost (src/net/class-prior.cc:80-91)
 * m[r][k] -= prior_scale * log_priors[k] when log_priors_host (cols floats) is not NULL.  Synchronises. */
int eesen_op_log_sub_prior(int device, void* stream, float* m_dev, int rows, int cols, int ld, int apply_log,
                           const float* log_priors_host, float prior_scale);

/* Average milliseconds (HIP events) of `iters` back-to-back launches of the same GEMM (beta = 0, no bias);
 * microbenchmark entry point used by scripts/gemm_bench.py. */
int eesen_op_gemm_bench(int device, int a_kc, int b_kc, int M, int N, int K, const float* A, int lda,
                        const float* B, int ldb, float* C, int ldc, int iters, float* avg_ms);

/* The operand bounds of the two-plane GEMM arithmetic (eesen_set_gemm_mode(2), csrc/gemm.hip: amax_rows_cols), as the Net measures
 * them: out_rows[r] = max_c |m[r][c]| (rows floats) and / or out_cols[c] = max_r |m[r][c]| (cols floats) of a device matrix with row
 * stride ld, in ONE pass; either output may be NULL.  Device pointers.  Synchronises.
 * No counterpart in the reference (its GEMM is cublasSgemm, src/gpucompute/cuda-matrix.cc:604-639): exported for the tests. */
int eesen_op_amax_rows_cols(int device, const float* m_dev, long rows, int cols, int ld, float* out_rows_dev, float* out_cols_dev);

#ifdef __cplusplus
}
#endif
#endif /* EESEN_HIP_H_ */
