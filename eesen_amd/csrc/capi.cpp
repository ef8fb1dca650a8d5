// capi.cpp -- the extern "C" boundary declared in include/eesen_hip.h.  Every wrapper converts internal
// exceptions into a status code and a thread-local message; no exception leaves the library.
#include <cstdio>
#include <cstring>

#include <algorithm>

#include "guard.h"
#include "net.h"

using namespace eesen;

#include "handles.h"

namespace eesen {
std::string& last_error_slot() {
  thread_local std::string g_err;
  return g_err;
}
}  // namespace eesen

extern "C" {

const char* eesen_last_error(void) { return last_error_slot().c_str(); }
const char* eesen_version(void) { return "eesen_hip 0.1.0 (gfx950)"; }

int eesen_device_count(int* count) {
  return guard([&] {
    REQ_PTR(count);
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) n = 0;
    *count = n;
  });
}

int eesen_set_gemm_mode(int mode) {
  return guard([&] {
    EESEN_REQUIRE(mode >= -1 && mode <= 2, EESEN_ERR_INVALID, "gemm mode must be -1 (environment), 0 (f32 MFMA), 1 (three bf16 planes) or 2 (two fp16 planes)");
    set_gemm_mode(mode);
  });
}
int eesen_get_gemm_mode(int* mode) {
  return guard([&] { REQ_PTR(mode); *mode = gemm_mode(); });
}

int eesen_device_synchronize(int device) {
  return guard([&] {
    EESEN_HIP_CHECK(hipSetDevice(device));
    EESEN_HIP_CHECK(hipDeviceSynchronize());
  });
}

int eesen_net_create(int device, void* stream, eesen_net_t** out) {
  return guard([&] {
    REQ_PTR(out);
    *out = new eesen_net(device, stream);
  });
}
int eesen_net_destroy(eesen_net_t* net) {
  return guard([&] { delete net; });
}
int eesen_net_add_layer(eesen_net_t* net, int kind, int in_dim, int out_dim, float coef, float max_grad) {
  return guard([&] { REQ_PTR(net); net->add_layer(kind, in_dim, out_dim, coef, max_grad); });
}
int eesen_net_set_train_mode(eesen_net_t* net, int train) {
  return guard([&] { REQ_PTR(net); net->set_train_mode(train != 0); });
}
int eesen_net_set_dropout_seed(eesen_net_t* net, unsigned long long seed) {
  return guard([&] { REQ_PTR(net); net->drop_seed = seed; net->drop_counter = 0; });
}
int eesen_net_set_layer_dropout(eesen_net_t* net, int layer, const float* nine) {
  return guard([&] { REQ_PTR(net); REQ_PTR(nine); net->set_layer_dropout(layer, nine); });
}
int eesen_net_get_layer_dropout(eesen_net_t* net, int layer, float* nine) {
  return guard([&] { REQ_PTR(net); REQ_PTR(nine); net->get_layer_dropout(layer, nine); });
}
int eesen_net_set_dropout_masks(eesen_net_t* net, int layer, const float* fwd_mask_host, long fwd_floats,
                                const float* rec_mask_host, int rec_rows, long rec_floats, int twiddle_coin) {
  return guard([&] { REQ_PTR(net); net->set_dropout_masks(layer, fwd_mask_host, fwd_floats, rec_mask_host, rec_rows, rec_floats, twiddle_coin); });
}
int eesen_net_get_dropout_masks(eesen_net_t* net, int layer, float* fwd_mask_host, float* rec_mask_host, int* info4) {
  return guard([&] { REQ_PTR(net); net->get_dropout_masks(layer, fwd_mask_host, rec_mask_host, info4); });
}
int eesen_net_finalize(eesen_net_t* net) {
  return guard([&] { REQ_PTR(net); net->finalize(); });
}
int eesen_net_read(eesen_net_t* net, const char* path) {
  return guard([&] { REQ_PTR(net); REQ_PTR(path); net->read(path); });
}
int eesen_net_write(eesen_net_t* net, const char* path, int binary) {
  return guard([&] { REQ_PTR(net); REQ_PTR(path); net->write(path, binary != 0); });
}
int eesen_net_num_layers(eesen_net_t* net, int* n) {
  return guard([&] { REQ_PTR(net); REQ_PTR(n); *n = (int)net->layers.size(); });
}
int eesen_net_layer_info(eesen_net_t* net, int idx, int* kind, int* in_dim, int* out_dim, float* coef, float* max_grad) {
  return guard([&] {
    REQ_PTR(net);
    EESEN_REQUIRE(idx >= 0 && idx < (int)net->layers.size(), EESEN_ERR_INVALID, "layer index out of range");
    const Layer& L = net->layers[idx];
    if (kind) *kind = L.kind;
    if (in_dim) *in_dim = L.din_f;     // the model's own dimensions (an LSTM layer the library pads to a multiple of 4 cells reports what the file says)
    if (out_dim) *out_dim = L.dout_f;
    if (coef) *coef = L.coef;
    if (max_grad) *max_grad = L.max_grad;
  });
}
int eesen_net_layer_marker(eesen_net_t* net, int idx, char* buf, int cap) {
  return guard([&] {
    REQ_PTR(net); REQ_PTR(buf);
    EESEN_REQUIRE(idx >= 0 && idx < (int)net->layers.size() && cap > 0, EESEN_ERR_INVALID, "layer index out of range");
    std::snprintf(buf, (size_t)cap, "%s", net->layers[idx].marker());
  });
}
int eesen_net_tensor_moments(eesen_net_t* net, int which, int layer, double* out6_host, int cap_tensors, int* n_tensors) {
  return guard([&] {
    REQ_PTR(net); REQ_PTR(n_tensors);
    *n_tensors = net->tensor_moments(which, layer, out6_host, cap_tensors);
  });
}
int eesen_net_input_dim(eesen_net_t* net, int* dim) {
  return guard([&] { REQ_PTR(net); REQ_PTR(dim); EESEN_REQUIRE(!net->layers.empty(), EESEN_ERR_STATE, "empty net"); *dim = net->layers.front().din_f; });
}
int eesen_net_output_dim(eesen_net_t* net, int* dim) {
  return guard([&] { REQ_PTR(net); REQ_PTR(dim); EESEN_REQUIRE(!net->layers.empty(), EESEN_ERR_STATE, "empty net"); *dim = net->layers.back().dout_f; });
}
int eesen_net_num_params(eesen_net_t* net, long* n) {
  return guard([&] { REQ_PTR(net); REQ_PTR(n); *n = net->num_params(); });
}
int eesen_net_get_params(eesen_net_t* net, float* host_flat, long n) {
  return guard([&] { REQ_PTR(net); REQ_PTR(host_flat); net->get_flat(net->params, host_flat, n); });
}
int eesen_net_set_params(eesen_net_t* net, const float* host_flat, long n) {
  return guard([&] { REQ_PTR(net); REQ_PTR(host_flat); net->set_params(host_flat, n); });
}
int eesen_net_set_train_options(eesen_net_t* net, float learn_rate, float momentum) {
  return guard([&] { REQ_PTR(net); net->lr = learn_rate; net->mmt = momentum; });
}
int eesen_net_set_update_algorithm(eesen_net_t* net, const char* name) {
  return guard([&] {
    REQ_PTR(net); REQ_PTR(name);
    const std::string a(name);  // Net::SetUpdateAlgorithm, net.cc:481-497
    if (a == "SGD") net->rule = 0;
    else if (a == "Adagrad") net->rule = 1;
    else if (a == "RMSProp") net->rule = 2;
    else throw Error(EESEN_ERR_INVALID, "unknown optimization algorithm '" + a + "' (SGD|Adagrad|RMSProp)");
  });
}
int eesen_net_set_adaptive_options(eesen_net_t* net, float adagrad_epsilon, float rmsprop_rho) {
  return guard([&] { REQ_PTR(net); net->ada_eps = adagrad_epsilon; net->rms_rho = rmsprop_rho; net->rms_one_minus_rho = 1.0f - rmsprop_rho; });
}
int eesen_net_get_accumulators(eesen_net_t* net, float* host_flat, long n) {
  return guard([&] { REQ_PTR(net); REQ_PTR(host_flat); net->init_accu(); net->get_flat(net->accu, host_flat, n); });
}
int eesen_net_set_accumulators(eesen_net_t* net, const float* host_flat, long n) {
  return guard([&] { REQ_PTR(net); REQ_PTR(host_flat); net->set_accu(host_flat, n); });
}
int eesen_net_set_seq_lengths(eesen_net_t* net, const int* lens, int S) {
  return guard([&] { REQ_PTR(net); REQ_PTR(lens); net->set_seq_lengths(lens, S); });
}
int eesen_net_propagate(eesen_net_t* net, const float* in, int rows, int in_ld, int in_is_device, const float** out_dev,
                        int* out_cols, int* out_ld) {
  return guard([&] {
    REQ_PTR(net); REQ_PTR(in);
    net->propagate(in, rows, in_ld, in_is_device != 0);
    if (out_dev) *out_dev = net->out_ptr;
    if (out_cols) *out_cols = net->out_cols;
    if (out_ld) *out_ld = net->out_ld;
  });
}
int eesen_net_get_output(eesen_net_t* net, float* host_out, long n) {
  return guard([&] {
    REQ_PTR(net); REQ_PTR(host_out);
    EESEN_REQUIRE(net->propagated, EESEN_ERR_STATE, "no Propagate output");
    EESEN_REQUIRE(n == (long)net->rows * net->out_cols, EESEN_ERR_INVALID, "output size mismatch");
    net->sync();
    EESEN_HIP_CHECK(hipMemcpy2D(host_out, (size_t)net->out_cols * 4, net->out_ptr, (size_t)net->out_ld * 4, (size_t)net->out_cols * 4,
                                net->rows, hipMemcpyDeviceToHost));
  });
}
int eesen_net_backpropagate(eesen_net_t* net, const float* out_diff_dev, int out_diff_ld, float* in_diff_dev, int in_diff_ld) {
  return guard([&] { REQ_PTR(net); REQ_PTR(out_diff_dev); net->backpropagate(out_diff_dev, out_diff_ld, in_diff_dev, in_diff_ld); });
}
int eesen_net_grad_buffer(eesen_net_t* net, float** dev_ptr, long* n) {
  return guard([&] { REQ_PTR(net); REQ_PTR(dev_ptr); REQ_PTR(n); *dev_ptr = net->fresh.p; *n = (long)net->P; });
}
int eesen_net_get_grads(eesen_net_t* net, float* host_flat, long n) {
  return guard([&] { REQ_PTR(net); REQ_PTR(host_flat); net->get_flat(net->fresh, host_flat, n); });
}
int eesen_net_update(eesen_net_t* net) {
  return guard([&] { REQ_PTR(net); net->update(); });
}
int eesen_net_set_forward_precision(eesen_net_t* net, int bf16) {
  return guard([&] { REQ_PTR(net); net->fwd_bf16 = bf16 != 0; net->fwd_bf16_rec = bf16 == 1; });
}
int eesen_net_bf16_recurrence_layers(eesen_net_t* net, int* layers) {
  return guard([&] { REQ_PTR(net); REQ_PTR(layers); *layers = net->info_fwd_bf16; });
}
int eesen_net_plan_string(eesen_net_t* net, char* json, int cap) {
  return guard([&] {
    REQ_PTR(net); REQ_PTR(json);
    const std::string s = net->plan_string();
    EESEN_REQUIRE(cap > (int)s.size(), EESEN_ERR_INVALID, "eesen_net_plan_string: buffer too small (" + std::to_string(s.size() + 1) + " bytes needed)");
    std::memcpy(json, s.c_str(), s.size() + 1);
  });
}
int eesen_net_recurrence_info(eesen_net_t* net, int* out3) {  // four ints
  return guard([&] {
    REQ_PTR(net); REQ_PTR(out3);
    out3[0] = net->info_lstm_layers; out3[1] = net->info_fwd_persistent; out3[2] = net->info_bwd_persistent;
    out3[3] = net->recoveries;
  });
}
int eesen_net_debug_set_error_word(eesen_net_t* net, unsigned value) {
  return guard([&] {
    REQ_PTR(net);
    EESEN_HIP_CHECK(hipSetDevice(net->device));
    EESEN_HIP_CHECK(hipMemsetD32Async(reinterpret_cast<hipDeviceptr_t>(net->ctl.p + kCtlWords - 1), (int)value, 1, net->st));
  });
}
int eesen_net_synchronize(eesen_net_t* net) {
  return guard([&] { REQ_PTR(net); net->sync(); });
}
int eesen_net_set_profiling(eesen_net_t* net, int on) {
  return guard([&] { REQ_PTR(net); net->timer.enable(on != 0); net->timer.set_accumulate(on == 2); });
}
int eesen_net_get_phase_spans(eesen_net_t* net, int* phases, float* seconds, int cap, int* n) {
  return guard([&] { REQ_PTR(net); REQ_PTR(n); *n = net->timer.spans(phases, seconds, cap); });
}
int eesen_net_get_phase_times(eesen_net_t* net, float* out6) {
  return guard([&] { REQ_PTR(net); REQ_PTR(out6); net->timer.collect(out6, 6); });
}

int eesen_ctc_create(int device, void* stream, eesen_ctc_t** out) {
  return guard([&] { REQ_PTR(out); *out = new eesen_ctc(device, stream); });
}
int eesen_ctc_destroy(eesen_ctc_t* ctc) {
  return guard([&] { delete ctc; });
}
int eesen_ctc_eval_parallel(eesen_ctc_t* ctc, const int* frame_num_utt, int S, const float* net_out_dev, int rows, int K,
                            int ld, const int* label_ids, const int* label_off, float* diff_dev, int diff_ld,
                            float* pzx_host) {
  return guard([&] {
    REQ_PTR(ctc); REQ_PTR(frame_num_utt); REQ_PTR(net_out_dev); REQ_PTR(label_ids); REQ_PTR(label_off); REQ_PTR(diff_dev);
    ctc->eval_parallel(frame_num_utt, S, net_out_dev, rows, K, ld, label_ids, label_off, diff_dev, diff_ld, pzx_host);
  });
}
int eesen_ctc_error_rate_mseq(eesen_ctc_t* ctc, const int* frame_num_utt, int S, const float* net_out_dev, int rows, int K,
                              int ld, const int* label_ids, const int* label_off, int* num_err, int* num_ref) {
  return guard([&] {
    REQ_PTR(ctc); REQ_PTR(frame_num_utt); REQ_PTR(net_out_dev); REQ_PTR(label_ids); REQ_PTR(label_off);
    ctc->error_rate_mseq(frame_num_utt, S, net_out_dev, rows, K, ld, label_ids, label_off, num_err, num_ref);
  });
}
int eesen_ctc_stats(eesen_ctc_t* ctc, double* obj_sum, long* sequences, long* frames, long* err_tokens, long* ref_tokens) {
  return guard([&] {
    REQ_PTR(ctc);
    ctc->flush();  // deferred results of earlier calls
    if (obj_sum) *obj_sum = ctc->obj_sum;
    if (sequences) *sequences = ctc->sequences;
    if (frames) *frames = ctc->frames;
    if (err_tokens) *err_tokens = ctc->err_tokens;
    if (ref_tokens) *ref_tokens = ctc->ref_tokens;
  });
}
int eesen_ctc_get_alpha_beta(eesen_ctc_t* ctc, float* alpha_host, float* beta_host, int* Lprime) {
  return guard([&] { REQ_PTR(ctc); ctc->get_alpha_beta(alpha_host, beta_host, Lprime); });
}
int eesen_ctc_set_sequence_out_file(eesen_ctc_t* ctc, const char* path) {
  return guard([&] {
    REQ_PTR(ctc);
    ctc->flush();
    ctc->seq_out = path ? path : "";
    if (!ctc->seq_out.empty()) std::remove(ctc->seq_out.c_str());  // train-ctc-parallel.cc:134-137
  });
}
int eesen_ctc_set_guard(eesen_ctc_t* ctc, eesen_net_t* net) {
  return guard([&] {
    REQ_PTR(ctc);
    ctc->flush();
    // the guard word is read with copies enqueued on the Ctc's stream: only in the Net's own stream are they ordered behind the
    // kernels that raise it (a Net and a Ctc created without a stream share the device's default stream, like the reference's
    // single-stream CuDevice)
    if (net) EESEN_REQUIRE(net->device == ctc->device && net->st == ctc->st, EESEN_ERR_INVALID, "eesen_ctc_set_guard: the Ctc and the Net must live on the same device and stream");
    if (ctc->guard_net) {   // unhook from the Net guarded so far
      auto& g = ctc->guard_net->guards;
      g.erase(std::remove(g.begin(), g.end(), static_cast<Ctc*>(ctc)), g.end());
    }
    ctc->guard = net && net->ctl.p ? net->ctl.p + kCtlWords - 1 : nullptr;
    ctc->guard_net = ctc->guard ? net : nullptr;
    if (ctc->guard_net) net->guards.push_back(ctc);
  });
}
int eesen_ctc_dropped(eesen_ctc_t* ctc, long* minibatches) {
  return guard([&] { REQ_PTR(ctc); REQ_PTR(minibatches); ctc->flush(); *minibatches = ctc->dropped; });
}
int eesen_ctc_set_profiling(eesen_ctc_t* ctc, int mode) {
  return guard([&] { REQ_PTR(ctc); ctc->timer.enable(mode == 2); ctc->timer.set_accumulate(mode == 2); });
}
int eesen_ctc_get_phase_times(eesen_ctc_t* ctc, float* out3) {
  return guard([&] { REQ_PTR(ctc); REQ_PTR(out3); ctc->phase_times(out3); });
}

int eesen_dev_alloc(int device, long bytes, void** dev_ptr) {
  return guard([&] {
    REQ_PTR(dev_ptr);
    EESEN_HIP_CHECK(hipSetDevice(device));
    EESEN_HIP_CHECK(hipMalloc(dev_ptr, (size_t)bytes));
  });
}
int eesen_dev_free(int device, void* dev_ptr) {
  return guard([&] {
    EESEN_HIP_CHECK(hipSetDevice(device));
    EESEN_HIP_CHECK(hipFree(dev_ptr));
  });
}
int eesen_dev_copy(int device, void* dst, const void* src, long bytes, int kind) {
  return guard([&] {
    EESEN_HIP_CHECK(hipSetDevice(device));
    const hipMemcpyKind k = kind == 1 ? hipMemcpyHostToDevice : kind == 2 ? hipMemcpyDeviceToHost : hipMemcpyDeviceToDevice;
    EESEN_HIP_CHECK(hipMemcpy(dst, src, (size_t)bytes, k));
  });
}

int eesen_op_gemm(int device, void* stream, int a_kc, int b_kc, int M, int N, int K, float alpha, const float* A, int lda,
                  const float* B, int ldb, float beta, float* C, int ldc, const float* bias) {
  return guard([&] {
    EESEN_HIP_CHECK(hipSetDevice(device));
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    // private split-K workspace for this standalone call
    DevBuf<float> ws;
    ws.reserve((size_t)8 << 20);
    gemm_f32(st, a_kc != 0, b_kc != 0, M, N, K, alpha, A, lda, B, ldb, beta, C, ldc, bias, ws.p, ws.cap);
    EESEN_HIP_CHECK(hipStreamSynchronize(st));
  });
}

int eesen_op_gemm_async(int device, void* stream, int a_kc, int b_kc, int M, int N, int K, const float* A, int lda, const float* B, int ldb,
                        float* C, int ldc, float* ws, long ws_floats, int extra_lds_bytes) {
  return guard([&] {
    REQ_PTR(A); REQ_PTR(B); REQ_PTR(C);
    EESEN_REQUIRE(ws_floats >= 0 && (ws || ws_floats == 0) && extra_lds_bytes >= 0, EESEN_ERR_INVALID, "bad workspace / LDS cap");
    EESEN_HIP_CHECK(hipSetDevice(device));
    gemm_f32(reinterpret_cast<hipStream_t>(stream), a_kc != 0, b_kc != 0, M, N, K, 1.f, A, lda, B, ldb, 0.f, C, ldc, nullptr, ws, (size_t)ws_floats,
             extra_lds_bytes);
  });
}

int eesen_op_log_sub_prior(int device, void* stream, float* m_dev, int rows, int cols, int ld, int apply_log,
                           const float* log_priors_host, float prior_scale) {
  return guard([&] {
    REQ_PTR(m_dev);
    EESEN_HIP_CHECK(hipSetDevice(device));
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    DevBuf<float> pri;
    if (log_priors_host) {
      pri.reserve(cols);
      EESEN_HIP_CHECK(hipMemcpyAsync(pri.p, log_priors_host, sizeof(float) * cols, hipMemcpyHostToDevice, st));
    }
    log_sub_prior(st, m_dev, ld, rows, cols, apply_log != 0, log_priors_host ? pri.p : nullptr, prior_scale);
    EESEN_HIP_CHECK(hipStreamSynchronize(st));
  });
}

int eesen_op_gemm_bench(int device, int a_kc, int b_kc, int M, int N, int K, const float* A, int lda, const float* B, int ldb,
                        float* C, int ldc, int iters, float* avg_ms) {
  return guard([&] {
    REQ_PTR(avg_ms);
    EESEN_HIP_CHECK(hipSetDevice(device));
    DevBuf<float> ws;
    ws.reserve((size_t)16 << 20);
    hipEvent_t e0, e1;
    EESEN_HIP_CHECK(hipEventCreate(&e0));
    EESEN_HIP_CHECK(hipEventCreate(&e1));
    // the operand bounds of the two-plane mode are measured once, outside the timed loop (the Net keeps them per tensor)
    DevBuf<float> am;
    am.reserve((size_t)M + N + (size_t)kAmaxBlocks * 16384);
    GemmBound ba{am.p, 1}, bb{am.p + M, 1};
    float* aws = am.p + M + N;
    if (a_kc) amax_rows_cols(nullptr, A, M, K, lda, am.p, nullptr, nullptr); else amax_rows_cols(nullptr, A, K, M, lda, nullptr, am.p, aws);
    if (b_kc) amax_rows_cols(nullptr, B, N, K, ldb, am.p + M, nullptr, nullptr); else amax_rows_cols(nullptr, B, K, N, ldb, nullptr, am.p + M, aws);
    for (int i = 0; i < 2; ++i) gemm_f32(nullptr, a_kc != 0, b_kc != 0, M, N, K, 1.f, A, lda, B, ldb, 0.f, C, ldc, nullptr, ws.p, ws.cap, 0, false, ba, bb);
    EESEN_HIP_CHECK(hipEventRecord(e0, nullptr));
    for (int i = 0; i < iters; ++i) gemm_f32(nullptr, a_kc != 0, b_kc != 0, M, N, K, 1.f, A, lda, B, ldb, 0.f, C, ldc, nullptr, ws.p, ws.cap, 0, false, ba, bb);
    EESEN_HIP_CHECK(hipEventRecord(e1, nullptr));
    EESEN_HIP_CHECK(hipEventSynchronize(e1));
    float ms = 0.f;
    EESEN_HIP_CHECK(hipEventElapsedTime(&ms, e0, e1));
    *avg_ms = ms / iters;
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
  });
}

int eesen_op_amax_rows_cols(int device, const float* m_dev, long rows, int cols, int ld, float* out_rows_dev, float* out_cols_dev) {
  return guard([&] {
    REQ_PTR(m_dev);
    EESEN_REQUIRE(rows > 0 && cols > 0 && ld >= cols, EESEN_ERR_INVALID, "bad matrix shape");
    EESEN_HIP_CHECK(hipSetDevice(device));
    DevBuf<float> ws;
    if (out_cols_dev) ws.reserve((size_t)kAmaxBlocks * 16384);
    amax_rows_cols(nullptr, m_dev, rows, cols, ld, out_rows_dev, out_cols_dev, ws.p);
    EESEN_HIP_CHECK(hipStreamSynchronize(nullptr));
  });
}

}  // extern "C"
