// oracle/ref_build/ref_cuda_emul.cc -- TEST INFRASTRUCTURE (oracle), not product code.
//
// This file is never compiled on its own: oracle/ref_build/Makefile streams
//     sed 's/<<<...>>>//' /root/reference/src/gpucompute/cuda-kernels.cu ; cat ref_cuda_emul.cc
// into ONE g++ translation unit (with shim/cuda_cpu_shim.h force-included), so that the code
// below can call the reference's own `cudaF_*` launchers.  With the <<<grid,block>>> syntax
// removed a launcher runs the kernel body for exactly ONE emulated CUDA thread, the one whose
// coordinates are in the shim's blockIdx/threadIdx globals; emulate_launch() walks the grid.
// No reference source is copied into this repository: the .cu is read where it lies.
//
// Why: the reference's CTC has NO CPU implementation (every CuMatrixBase::ComputeCtc* has an
// empty CPU branch, src/gpucompute/cuda-matrix.cc:861-864,894-897,927-930,961-964,994-997,
// 1030-1033), so the only way to run the reference's CTC arithmetic in this container is to run
// its CUDA kernels' C++ bodies on the CPU.  The host orchestration restated here follows
// Ctc::EvalParallel, src/net/ctc-loss.cc:101-169, and the launch shapes follow
// src/gpucompute/cuda-matrix.cc:868-898 (alpha), :934-965 (beta), :1001-1034 (error).

#include <vector>

namespace {

template <typename F>
void emulate_launch(dim3 Gr, dim3 Bl, F body) {
  gridDim = Gr;
  blockDim = Bl;
  for (unsigned by = 0; by < Gr.y; by++)
    for (unsigned bx = 0; bx < Gr.x; bx++)
      for (unsigned ty = 0; ty < Bl.y; ty++)
        for (unsigned tx = 0; tx < Bl.x; tx++) {
          blockIdx.x = bx; blockIdx.y = by; blockIdx.z = 0;
          threadIdx.x = tx; threadIdx.y = ty; threadIdx.z = 0;
          body();
        }
}

inline int n_blocks(int size, int block_size) {  // cuda-common.h: n_blocks()
  return size / block_size + ((size % block_size == 0) ? 0 : 1);
}

}  // namespace

extern "C" {

// Returns exp_len_labels (= 2*max_label_len+1).  alpha/beta: [T*S x exp_len] row-major, dense.
// probs: [T*S x K] softmax outputs, row t*S+s.  labels: CSR (off[S+1], ids[]).
// diff: [T*S x K] = d(-ln p)/d(logits) exactly as Ctc::EvalParallel leaves it in *diff.
// ctc_err (nullable): the intermediate error kernel output (before MulElements).
int ref_cuda_ctc_eval_parallel(const float* probs, int T, int S, int K, const int* frame_num_utt,
                               const int* label_ids, const int* label_off,
                               float* alpha, float* beta, float* pzx, float* diff, float* ctc_err_out,
                               int alpha_capacity_cols) {
  const int num_sequence = S, num_frames = T * S, num_classes = K;
  int max_label_len = 0;
  for (int s = 0; s < S; s++) max_label_len = std::max(max_label_len, label_off[s + 1] - label_off[s]);
  const int exp_len_labels = 2 * max_label_len + 1;                       // ctc-loss.cc:118
  if (exp_len_labels > alpha_capacity_cols) return -exp_len_labels;

  // label expansion, ctc-loss.cc:116-129
  std::vector<int> label_expand(num_sequence * exp_len_labels, -1), label_lengths_utt(S);
  for (int s = 0; s < S; s++) {
    const int U = label_off[s + 1] - label_off[s];
    label_lengths_utt[s] = 2 * U + 1;
    for (int l = 0; l < U; l++) {
      label_expand[s * exp_len_labels + 2 * l] = 0;
      label_expand[s * exp_len_labels + 2 * l + 1] = label_ids[label_off[s] + l];
    }
    label_expand[s * exp_len_labels + 2 * U] = 0;
  }

  // log scale, ctc-loss.cc:132-133 (kernel _apply_log, cuda-kernels.cu:218-226: mat[i] = log(mat[i]))
  std::vector<float> log_nnet_out((size_t)num_frames * K);
  for (size_t i = 0; i < log_nnet_out.size(); i++) log_nnet_out[i] = log(probs[i]);

  MatrixDim dim_alpha = {num_frames, exp_len_labels, exp_len_labels};
  MatrixDim dim_prob = {num_frames, K, K};
  for (size_t i = 0; i < (size_t)num_frames * exp_len_labels; i++) {        // ctc-loss.cc:138-139
    alpha[i] = NumericLimits<float>::log_zero_;
    beta[i] = NumericLimits<float>::log_zero_;
  }
  {
    dim3 dimBlock(CU2DBLOCK, CU2DBLOCK);
    dim3 dimGrid(n_blocks(num_sequence, CU2DBLOCK), n_blocks(exp_len_labels, CU2DBLOCK));  // cuda-matrix.cc:878-879
    for (int t = 0; t < T; t++)                                          // ctc-loss.cc:140-142
      emulate_launch(dimGrid, dimBlock, [&] {
        cudaF_compute_ctc_alpha_multiple_sequence(dimGrid, dimBlock, alpha, num_sequence, t, dim_alpha,
                                                  log_nnet_out.data(), dim_prob, label_expand.data(),
                                                  exp_len_labels, frame_num_utt);
      });
    for (int t = T - 1; t >= 0; t--)                                     // ctc-loss.cc:143-145
      emulate_launch(dimGrid, dimBlock, [&] {
        cudaF_compute_ctc_beta_multiple_sequence(dimGrid, dimBlock, beta, num_sequence, t, dim_alpha,
                                                 log_nnet_out.data(), dim_prob, label_expand.data(),
                                                 exp_len_labels, frame_num_utt, label_lengths_utt.data());
      });
  }
  for (int s = 0; s < S; s++) {                                          // ctc-loss.cc:147-153
    int label_len = label_lengths_utt[s];
    int frame_num = frame_num_utt[s];
    float tmp1 = alpha[(size_t)((frame_num - 1) * num_sequence + s) * exp_len_labels + label_len - 1];
    float tmp2 = alpha[(size_t)((frame_num - 1) * num_sequence + s) * exp_len_labels + label_len - 2];
    pzx[s] = tmp1 + log(1 + ExpA(tmp2 - tmp1));
  }

  std::vector<float> ctc_err((size_t)num_frames * K, 0.0f);               // ctc-loss.cc:156
  {
    MatrixDim dim_err = {num_frames, K, K};
    dim3 dimBlock(CU2DBLOCK, CU2DBLOCK);
    dim3 dimGrid(n_blocks(num_frames, CU2DBLOCK), n_blocks(K, CU2DBLOCK));  // cuda-matrix.cc:1010-1011
    emulate_launch(dimGrid, dimBlock, [&] {
      cudaF_compute_ctc_error_multiple_sequence(dimGrid, dimBlock, ctc_err.data(), num_sequence, dim_err, alpha, beta,
                                                dim_alpha, probs, label_expand.data(), exp_len_labels,
                                                frame_num_utt, pzx);
    });
  }
  if (ctc_err_out) memcpy(ctc_err_out, ctc_err.data(), ctc_err.size() * sizeof(float));

  // softmax Jacobian, ctc-loss.cc:160-168
  for (int r = 0; r < num_frames; r++) {
    float* e = &ctc_err[(size_t)r * K];
    const float* y = &probs[(size_t)r * K];
    float row_sum = 0.0f;
    for (int k = 0; k < K; k++) { e[k] *= y[k]; row_sum += e[k]; }
    for (int k = 0; k < K; k++) diff[(size_t)r * K + k] = e[k] - y[k] * row_sum;
  }
  (void)num_classes;
  return exp_len_labels;
}

// The SINGLE-sequence path, Ctc::Eval (src/net/ctc-loss.cc:28-75) with the one-sequence kernels
// (cuda-kernels.cu:1332-1408 alpha, :1448-1544 beta, :1584-1640 error; launch shapes cuda-matrix.cc:850-851, :987-988):
// what the reference's train-ctc (one utterance at a time, <BiLstm> layers) computes.  probs [T x K], labels[U].
// alpha / beta: [T x 2U+1], *pzx = ln p(z|x), diff [T x K] as Ctc::Eval leaves it.
int ref_cuda_ctc_eval(const float* probs, int T, int K, const int* label, int U, float* alpha, float* beta, float* pzx, float* diff) {
  const int num_frames = T, exp_len_labels = 2 * U + 1;                     // :36-37
  std::vector<int> label_expand(exp_len_labels, 0);                          // :39-43
  for (int l = 0; l < U; l++) label_expand[2 * l + 1] = label[l];
  std::vector<float> log_nnet_out((size_t)T * K);                            // :46-47
  for (size_t i = 0; i < log_nnet_out.size(); i++) log_nnet_out[i] = log(probs[i]);
  MatrixDim dim_alpha = {num_frames, exp_len_labels, exp_len_labels};
  MatrixDim dim_prob = {num_frames, K, K};
  for (size_t i = 0; i < (size_t)T * exp_len_labels; i++) alpha[i] = beta[i] = 0.0f;   // Resize(..., kSetZero), :49-50
  {
    dim3 dimBlock(CU1DBLOCK), dimGrid(n_blocks(exp_len_labels, CU1DBLOCK));  // cuda-matrix.cc:850-851
    for (int t = 0; t < T; t++)                                              // :51-53
      emulate_launch(dimGrid, dimBlock, [&] {
        cudaF_compute_ctc_alpha(dimGrid, dimBlock, alpha, t, dim_alpha, log_nnet_out.data(), dim_prob, label_expand.data());
      });
    for (int t = T - 1; t >= 0; t--)                                         // :54-56
      emulate_launch(dimGrid, dimBlock, [&] {
        cudaF_compute_ctc_beta(dimGrid, dimBlock, beta, t, dim_alpha, log_nnet_out.data(), dim_prob, label_expand.data());
      });
  }
  const float tmp1 = alpha[(size_t)(T - 1) * exp_len_labels + exp_len_labels - 1];   // :59-61
  const float tmp2 = alpha[(size_t)(T - 1) * exp_len_labels + exp_len_labels - 2];
  *pzx = tmp1 + log(1 + ExpA(tmp2 - tmp1));
  std::vector<float> ctc_err((size_t)T * K, 0.0f);                           // :64-65
  {
    MatrixDim dim_err = {num_frames, K, K};
    dim3 dimBlock(CU2DBLOCK, CU2DBLOCK);
    dim3 dimGrid(n_blocks(num_frames, CU2DBLOCK), n_blocks(K, CU2DBLOCK));   // cuda-matrix.cc:987-988
    emulate_launch(dimGrid, dimBlock, [&] {
      cudaF_compute_ctc_error(dimGrid, dimBlock, ctc_err.data(), dim_err, alpha, beta, dim_alpha, probs, label_expand.data(), *pzx);
    });
  }
  for (int r = 0; r < T; r++) {                                              // :68-75
    float* e = &ctc_err[(size_t)r * K];
    const float* y = &probs[(size_t)r * K];
    float row_sum = 0.0f;
    for (int k = 0; k < K; k++) { e[k] *= y[k]; row_sum += e[k]; }
    for (int k = 0; k < K; k++) diff[(size_t)r * K + k] = e[k] - y[k] * row_sum;
  }
  return exp_len_labels;
}

// Elementwise activation kernels as the CUDA side computes them (cuda-kernels.cu:686-740); used to
// quantify the CPU-vs-GPU activation-formula difference the survey notes (Appendix A).
void ref_cuda_sigmoid(float* y, const float* x, int rows, int cols) {
  MatrixDim d = {rows, cols, cols};
  dim3 Bl(CU2DBLOCK, CU2DBLOCK), Gr(n_blocks(cols, CU2DBLOCK), n_blocks(rows, CU2DBLOCK));
  emulate_launch(Gr, Bl, [&] { cudaF_sigmoid(Gr, Bl, y, x, d, cols); });
}
void ref_cuda_tanh(float* y, const float* x, int rows, int cols) {
  MatrixDim d = {rows, cols, cols};
  dim3 Bl(CU2DBLOCK, CU2DBLOCK), Gr(n_blocks(cols, CU2DBLOCK), n_blocks(rows, CU2DBLOCK));
  emulate_launch(Gr, Bl, [&] { cudaF_tanh(Gr, Bl, y, x, d, cols); });
}

// Adagrad / RMSProp update of one tensor exactly as TrainableLayer + BiLstm::Update compose it on the GPU
// (src/net/trainable-layer.h:65-114, src/net/bilstm-layer.h:885-955), through the reference's own elementwise kernel
// launchers (cuda-kernels.cu: _mul_elements, _scale, _add_mat, _sqrt_elements, _invert_elements, _add_mat_mat_elements).
// The CPU build of the reference cannot run these rules at all (CuMatrixBase::AddMatMatElements exit(-101),
// cuda-matrix.cc:657-661).  corr already holds momentum*corr + gradient, clipped.  rmsprop != 0 selects RMSProp.
void ref_cuda_adaptive_update(float* param, const float* corr, float* accu, int rows, int cols, float lr, float eps,
                              float rho, float one_minus_rho, int rmsprop) {
  MatrixDim d = {rows, cols, cols};
  dim3 Bl(CU2DBLOCK, CU2DBLOCK), Gr(n_blocks(cols, CU2DBLOCK), n_blocks(rows, CU2DBLOCK));
  std::vector<float> tmp(corr, corr + (size_t)rows * cols);                                     // grad_tmp.CopyFromMat(grad)
  emulate_launch(Gr, Bl, [&] { cudaF_mul_elements(Gr, Bl, tmp.data(), corr, d, cols); });        // grad_tmp.MulElements(grad)
  if (rmsprop) emulate_launch(Gr, Bl, [&] { cudaF_scale(Gr, Bl, accu, rho, d); });               // accu.Scale(rho)
  emulate_launch(Gr, Bl, [&] { cudaF_add_mat(Gr, Bl, rmsprop ? one_minus_rho : 1.0f, tmp.data(), accu, d, cols, 0); });
  std::vector<float> scale(accu, accu + (size_t)rows * cols);                                    // accu_scale.CopyFromMat(accu)
  emulate_launch(Gr, Bl, [&] { cudaF_sqrt_elements(Gr, Bl, scale.data(), eps, d); });             // ApplySqrt(epsilon)
  emulate_launch(Gr, Bl, [&] { cudaF_invert_elements(Gr, Bl, scale.data(), d); });                // InvertElements
  emulate_launch(Gr, Bl, [&] { cudaF_add_mat_mat_elements(Gr, Bl, param, scale.data(), corr, d, cols, cols, -lr, 1.0f); });
}

}  // extern "C"
