// feeder.hip -- minibatch assembly on the device, double-buffered (row a1 of the hot-path table).
//
// The reference builds every minibatch on the host: a zeroed host matrix of max_frame_num * S rows, one row copy per frame
// into row t*S + s (/root/reference/src/netbin/train-ctc-parallel.cc:186-193), then a blocking host-to-device copy of
// the PADDED matrix (:195 `feats_transf = feat_mat_host`).  Here the host only packs the S utterance matrices back to back
// into a pinned staging slot (no padding crosses PCIe); the copy runs on the feeder's own stream and a kernel writes the
// zero-padded, time-major interleaved matrix in HBM.  Slots rotate, so the staging of batch n+1 overlaps the training
// step of batch n; the compute stream only waits for the slot's event.
//
// The interleave is pure byte movement (HBM-bound, a few MB per batch): one thread per output float4 (or float when
// D % 4 != 0), reads coalesced along d within a frame, writes fully coalesced; padded frames are written as zeros by the
// same kernel, so the output needs no memset.
#include <cmath>
#include <cstring>
#include <string>
#include <vector>

#include "guard.h"

namespace eesen {
namespace {

template <int V>  // V = floats per thread (4: D % 4 == 0 and 16-byte aligned rows, else 1)
__global__ __launch_bounds__(256) void interleave_kernel(const float* __restrict__ packed, const long* __restrict__ off,
                                                        const int* __restrict__ frames, float* __restrict__ out, int T,
                                                        int S, int D, int ld) {
  const int dv = D / V;
  const long n = (long)T * S * dv;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const int d = (int)(i % dv) * V;
    const long r = i / dv;  // output row t*S + s
    const int s = (int)(r % S), t = (int)(r / S);
    float* dst = out + r * ld + d;
    if (V == 4) {
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (t < frames[s]) v = *reinterpret_cast<const float4*>(packed + off[s] + (long)t * D + d);
      *reinterpret_cast<float4*>(dst) = v;
    } else {
      *dst = t < frames[s] ? packed[off[s] + (long)t * D + d] : 0.f;
    }
  }
}

// ---- feature front end ----------------------------------------------------------------------------------------------------
// The recipes feed the trainer through a pipe of host filters, e.g. (asr_egs/wsj/steps/train_ctc_parallel.sh:95-110,
// decode_ctc_lat.sh:92-95, librispeech/steps/train_ctc_parallel_mult.sh:110-133)
//   apply-cmvn --norm-vars=true --utt2spk=ark:utt2spk scp:cmvn.scp scp:train.scp ark:- | splice-feats --left-context=1
//   --right-context=1 ark:- ark:- | subsample-feats --n=3 --offset=0 ark:- ark:- | add-deltas ark:- ark:- |
// Every one of these is a per-dimension affine map or a clamped row gather -- bytes, not arithmetic -- so they run here, on
// the packed utterance matrices in HBM between the PCIe copy and the interleave: the raw features cross PCIe once (three
// times fewer bytes than with deltas applied on the host) and no host filter processes sit in front of a GPU that consumes
// hundreds of thousands of frames per second.  One launch per stage, one thread per output element, utterance = blockIdx.y.
// Arithmetic is the reference's, operation for operation and WITHOUT fused multiply-add, so the results equal a scalar
// host evaluation bit for bit (the reference's own AddVec goes through BLAS saxpy, whose rounding depends on the BLAS build).
struct StageDev { int kind, a, b; };

__device__ __forceinline__ int clampi(int v, int hi) { return v < 0 ? 0 : (v > hi ? hi : v); }

__global__ __launch_bounds__(256) void feat_stage_kernel(StageDev st, const float* __restrict__ scales, const float* __restrict__ src,
                                                        const long* __restrict__ off_in, const int* __restrict__ fr_in, int Din,
                                                        float* __restrict__ dst, const long* __restrict__ off_out,
                                                        const int* __restrict__ fr_out, int Dout, const float* __restrict__ cmvn) {
#pragma clang fp contract(off)   // hipcc contracts a * b + c into one fma by default; __fmul_rn / __fadd_rn do not help (header inlines built with that default)
  // the delta windows (at most 9 orders x 33-wide steps: 1161 floats) sit in LDS: the tap loop then has no dependent global load
  // in front of every tap, and the taps' feature loads are independent of each other
  __shared__ float s_sc[1200];
  if (st.kind == EESEN_FEAT_DELTAS) {
    const int nsc = (st.a + 1) + st.b * st.a * (st.a + 1);
    for (int i = threadIdx.x; i < nsc; i += blockDim.x) s_sc[i] = scales[i];
    __syncthreads();
  }
  const int s = blockIdx.y;
  const int Ti = fr_in[s], To = fr_out[s];
  const float* x = src + off_in[s];
  float* y = dst + off_out[s];
  const long n = (long)To * Dout;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const int t = (int)(i / Dout), c = (int)(i % Dout);
    float v;
    if (st.kind == EESEN_FEAT_CMVN) {                      // feat/cmvn.cc:112-116: f = norm(0, d) + f * norm(1, d)
      const float* nm = cmvn + (size_t)s * 2 * Din;
      const float prod = x[(long)t * Din + c] * nm[Din + c];
      v = nm[c] + prod;
    } else if (st.kind == EESEN_FEAT_SPLICE) {             // feat/feature-functions.cc:401-411
      const int j = c / Din, d = c % Din;
      v = x[(long)clampi(t + j - st.a, Ti - 1) * Din + d];
    } else if (st.kind == EESEN_FEAT_SUBSAMPLE) {          // featbin/subsample-feats.cc:93-108
      const int ti = st.a > 0 ? st.b + st.a * t : t / (-st.a);
      v = x[(long)ti * Din + c];
    } else {                                               // DeltaFeatures::Process, feat/feature-functions.cc:252-266
      const int o = c / Din, d = c % Din;
      const int max_off = o * st.b;
      const float* sc = s_sc + o + st.b * o * (o - 1);     // windows of orders 0..o-1 hold 1 + 2kW entries each
      v = 0.f;
      for (int j = -max_off; j <= max_off; ++j) {
        const float w = sc[j + max_off];
        const float prod = w * x[(long)clampi(t + j, Ti - 1) * Din + d];
        v = w != 0.f ? v + prod : v;                         // a zero tap is SKIPPED by the reference (:263), not added
      }
    }
    y[i] = v;
  }
}

// DeltaFeatures::DeltaFeatures, feat/feature-functions.cc:216-241, in the reference's float arithmetic
std::vector<float> delta_scales(int order, int window) {
  std::vector<std::vector<float>> sc(order + 1);
  sc[0] = {1.0f};
  for (int i = 1; i <= order; ++i) {
    const std::vector<float>& prev = sc[i - 1];
    std::vector<float>& cur = sc[i];
    const int prev_offset = ((int)prev.size() - 1) / 2, cur_offset = prev_offset + window;
    cur.assign(prev.size() + 2 * window, 0.f);
    float normalizer = 0.0f;
    for (int j = -window; j <= window; ++j) {
      normalizer += j * j;
      for (int k = -prev_offset; k <= prev_offset; ++k) cur[j + k + cur_offset] += static_cast<float>(j) * prev[k + prev_offset];
    }
    const float alpha = (float)(1.0 / normalizer);           // VectorBase<float>::Scale(1.0 / normalizer)
    for (float& v : cur) v *= alpha;
  }
  std::vector<float> flat;
  for (const auto& v : sc) flat.insert(flat.end(), v.begin(), v.end());
  return flat;
}

}  // namespace

struct Feeder {
  struct Slot {
    float* host = nullptr;  // pinned
    size_t host_cap = 0;    // floats
    long* off_h = nullptr;  // pinned: S offsets (floats) + S frame counts packed behind them as ints
    size_t meta_cap = 0;    // sequences
    DevBuf<float> packed, out;
    DevBuf<long> off_d;
    DevBuf<int> frames_d;
    hipEvent_t ready = nullptr, consumed = nullptr;
    bool in_flight = false, has_consumer = false;
    int T = 0, S = 0, D = 0, ld = 0;
    // feature front end: the second packed buffer of the stage ping-pong, per-boundary offsets / frame counts, CMVN vectors
    DevBuf<float> packed2, cmvn_d;
    long* meta_h = nullptr;   // pinned: (n_stages + 1) x S offsets, then as many frame counts (ints)
    size_t meta2_cap = 0;     // (n_stages + 1) * S
    float* cmvn_h = nullptr;  // pinned
    size_t cmvn_cap = 0;
    DevBuf<long> offs_d;
    DevBuf<int> frs_d;
  };
  struct Stage { int kind, a, b; size_t scale_off; };
  std::vector<Stage> pipe;
  DevBuf<float> scales_d;
  int device;
  hipStream_t compute, copy = nullptr;
  std::vector<Slot> slots;
  int next = 0;

  Feeder(int dev, void* compute_stream, int nslots) : device(dev), compute(static_cast<hipStream_t>(compute_stream)) {
    EESEN_REQUIRE(nslots >= 1 && nslots <= 8, EESEN_ERR_INVALID, "1..8 staging slots");
    int n = 0;
    EESEN_REQUIRE(hipGetDeviceCount(&n) == hipSuccess && n > 0, EESEN_ERR_HIP, "no HIP device visible: the feeder has no CPU fallback");
    EESEN_REQUIRE(dev >= 0 && dev < n, EESEN_ERR_INVALID, "device index out of range");
    EESEN_HIP_CHECK(hipSetDevice(device));
    EESEN_HIP_CHECK(hipStreamCreateWithFlags(&copy, hipStreamNonBlocking));
    slots.resize(nslots);
    for (auto& s : slots) {
      EESEN_HIP_CHECK(hipEventCreateWithFlags(&s.ready, hipEventDisableTiming));
      EESEN_HIP_CHECK(hipEventCreateWithFlags(&s.consumed, hipEventDisableTiming));
    }
  }
  ~Feeder() {
    (void)hipSetDevice(device);
    if (copy) (void)hipStreamSynchronize(copy);
    for (auto& s : slots) {
      if (s.host) (void)hipHostFree(s.host);
      if (s.off_h) (void)hipHostFree(s.off_h);
      if (s.meta_h) (void)hipHostFree(s.meta_h);
      if (s.cmvn_h) (void)hipHostFree(s.cmvn_h);
      if (s.ready) (void)hipEventDestroy(s.ready);
      if (s.consumed) (void)hipEventDestroy(s.consumed);
    }
    if (copy) (void)hipStreamDestroy(copy);
  }

  int submit(const float* const* utts, const int* frames, const int* strides, int S, int D) {
    EESEN_REQUIRE(S > 0 && D > 0, EESEN_ERR_INVALID, "need at least one utterance and one feature column");
    EESEN_HIP_CHECK(hipSetDevice(device));
    long total = 0;
    int T = 0;
    for (int s = 0; s < S; ++s) {
      EESEN_REQUIRE(frames[s] >= 0, EESEN_ERR_INVALID, "negative frame count");
      EESEN_REQUIRE(frames[s] == 0 || utts[s] != nullptr, EESEN_ERR_INVALID, "null utterance matrix");
      EESEN_REQUIRE(!strides || frames[s] == 0 || strides[s] >= D, EESEN_ERR_INVALID, "row stride smaller than the feature dimension");
      total += (long)frames[s] * D;
      T = std::max(T, frames[s]);
    }
    EESEN_REQUIRE(T > 0, EESEN_ERR_INVALID, "every utterance is empty");
    const int id = next;
    next = (next + 1) % (int)slots.size();
    Slot& sl = slots[id];
    // the slot's previous batch: its copy + kernel must have run (we overwrite the pinned staging) and its consumer must
    // have finished reading the assembled matrix (we overwrite `out`)
    if (sl.in_flight) EESEN_HIP_CHECK(hipEventSynchronize(sl.ready));
    if (sl.has_consumer) EESEN_HIP_CHECK(hipEventSynchronize(sl.consumed));
    sl.in_flight = sl.has_consumer = false;
    if ((size_t)total > sl.host_cap) {
      if (sl.host) EESEN_HIP_CHECK(hipHostFree(sl.host));
      sl.host = nullptr;
      sl.host_cap = (size_t)total + (size_t)total / 4;
      EESEN_HIP_CHECK(hipHostMalloc(reinterpret_cast<void**>(&sl.host), sl.host_cap * sizeof(float), hipHostMallocDefault));
    }
    if ((size_t)S > sl.meta_cap) {
      if (sl.off_h) EESEN_HIP_CHECK(hipHostFree(sl.off_h));
      sl.off_h = nullptr;
      sl.meta_cap = (size_t)S;
      EESEN_HIP_CHECK(hipHostMalloc(reinterpret_cast<void**>(&sl.off_h), sl.meta_cap * (sizeof(long) + sizeof(int)), hipHostMallocDefault));
    }
    int* frames_h = reinterpret_cast<int*>(sl.off_h + sl.meta_cap);
    long o = 0;
    for (int s = 0; s < S; ++s) {  // pack: utterance matrices back to back, rows made contiguous
      sl.off_h[s] = o;
      frames_h[s] = frames[s];
      const int st = strides ? strides[s] : D;
      if (st == D) {
        std::memcpy(sl.host + o, utts[s], (size_t)frames[s] * D * sizeof(float));
      } else {
        for (int t = 0; t < frames[s]; ++t) std::memcpy(sl.host + o + (long)t * D, utts[s] + (long)t * st, (size_t)D * sizeof(float));
      }
      o += (long)frames[s] * D;
    }
    const int ld = (D + 3) / 4 * 4;  // rows of the assembled matrix start on 16 bytes (GEMM operand alignment)
    sl.packed.reserve((size_t)total);
    sl.off_d.reserve((size_t)S);
    sl.frames_d.reserve((size_t)S);
    const bool fresh = sl.out.reserve((size_t)T * S * ld);
    if (ld != D && fresh) EESEN_HIP_CHECK(hipMemsetAsync(sl.out.p, 0, sl.out.cap * sizeof(float), copy));  // pad columns stay zero
    else if (ld != D) EESEN_HIP_CHECK(hipMemsetAsync(sl.out.p, 0, (size_t)T * S * ld * sizeof(float), copy));
    if (total) EESEN_HIP_CHECK(hipMemcpyAsync(sl.packed.p, sl.host, (size_t)total * sizeof(float), hipMemcpyHostToDevice, copy));
    EESEN_HIP_CHECK(hipMemcpyAsync(sl.off_d.p, sl.off_h, (size_t)S * sizeof(long), hipMemcpyHostToDevice, copy));
    EESEN_HIP_CHECK(hipMemcpyAsync(sl.frames_d.p, frames_h, (size_t)S * sizeof(int), hipMemcpyHostToDevice, copy));
    const bool vec = D % 4 == 0;  // then every packed frame and every output row is 16-byte aligned
    const long n = (long)T * S * (vec ? D / 4 : D);
    const int blocks = (int)std::min<long>(cdivl(n, 256), 256L * 32);
    if (vec) interleave_kernel<4><<<blocks, 256, 0, copy>>>(sl.packed.p, sl.off_d.p, sl.frames_d.p, sl.out.p, T, S, D, ld);
    else interleave_kernel<1><<<blocks, 256, 0, copy>>>(sl.packed.p, sl.off_d.p, sl.frames_d.p, sl.out.p, T, S, D, ld);
    check_launch("interleave_kernel");
    EESEN_HIP_CHECK(hipEventRecord(sl.ready, copy));
    sl.in_flight = true;
    sl.T = T; sl.S = S; sl.D = D; sl.ld = ld;
    return id;
  }

  // ---- feature front end (see feat_stage_kernel) ----
  void set_pipeline(const eesen_feat_stage_t* st, int n) {
    EESEN_REQUIRE(n >= 0 && n <= 8, EESEN_ERR_INVALID, "0..8 front-end stages");
    EESEN_HIP_CHECK(hipSetDevice(device));
    std::vector<Stage> np;
    std::vector<float> scales;
    int n_cmvn = 0;
    for (int k = 0; k < n; ++k) {
      Stage g{st[k].kind, st[k].a, st[k].b, 0};
      switch (g.kind) {
        case EESEN_FEAT_CMVN: ++n_cmvn; break;
        case EESEN_FEAT_SPLICE:   // KALDI_ASSERT(left_context >= 0 && right_context >= 0), feature-functions.cc:398
          EESEN_REQUIRE(g.a >= 0 && g.b >= 0 && g.a + g.b <= 64, EESEN_ERR_INVALID, "splice-feats: context must be 0..64 frames");
          break;
        case EESEN_FEAT_SUBSAMPLE:  // subsample-feats.cc:52-55
          EESEN_REQUIRE(g.a != 0, EESEN_ERR_INVALID, "subsample-feats: n must not be 0");
          EESEN_REQUIRE(g.a > 0 ? g.b >= 0 : g.b == 0, EESEN_ERR_INVALID, "subsample-feats: --offset cannot be used with negative n (and must be >= 0)");
          break;
        case EESEN_FEAT_DELTAS: {  // feature-functions.cc:211-213
          EESEN_REQUIRE(g.a >= 0 && g.a <= 8 && g.b > 0 && g.b <= 16, EESEN_ERR_INVALID, "add-deltas: order 0..8, window 1..16");
          g.scale_off = scales.size();
          const std::vector<float> sc = delta_scales(g.a, g.b);
          scales.insert(scales.end(), sc.begin(), sc.end());
          break;
        }
        default: throw Error(EESEN_ERR_INVALID, "unknown front-end stage kind " + std::to_string(g.kind));
      }
      np.push_back(g);
    }
    EESEN_REQUIRE(n_cmvn <= 1, EESEN_ERR_INVALID, "at most one CMVN stage");
    for (auto& s : slots)  // batches in flight still read the old scale table
      if (s.in_flight) EESEN_HIP_CHECK(hipEventSynchronize(s.ready));
    if (!scales.empty()) {
      scales_d.reserve(scales.size());
      EESEN_HIP_CHECK(hipMemcpy(scales_d.p, scales.data(), scales.size() * sizeof(float), hipMemcpyHostToDevice));
    }
    pipe.swap(np);
  }
  static int stage_dim(const Stage& g, int D) {
    if (g.kind == EESEN_FEAT_SPLICE) return D * (1 + g.a + g.b);
    if (g.kind == EESEN_FEAT_DELTAS) return D * (1 + g.a);
    return D;
  }
  static int stage_frames(const Stage& g, int T) {
    if (g.kind != EESEN_FEAT_SUBSAMPLE) return T;
    if (g.a < 0) return T * (-g.a);
    return T > g.b ? (T - g.b + g.a - 1) / g.a : 0;   // number of k = offset, offset + n, ... < T (subsample-feats.cc:82-84)
  }
  int pipeline_dim(int D) const { for (const Stage& g : pipe) D = stage_dim(g, D); return D; }
  int pipeline_frames(int T) const { for (const Stage& g : pipe) T = stage_frames(g, T); return T; }
  int cmvn_dim(int D) const {  // feature dimension in front of the CMVN stage (0: no such stage)
    for (const Stage& g : pipe) { if (g.kind == EESEN_FEAT_CMVN) return D; D = stage_dim(g, D); }
    return 0;
  }

  int submit_raw(const float* const* utts, const int* frames, const int* strides, const float* const* cmvn, int S, int D) {
    if (pipe.empty()) return submit(utts, frames, strides, S, D);
    EESEN_REQUIRE(S > 0 && D > 0, EESEN_ERR_INVALID, "need at least one utterance and one feature column");
    EESEN_HIP_CHECK(hipSetDevice(device));
    const int nst = (int)pipe.size(), nb = nst + 1;
    const int Dc = cmvn_dim(D);
    EESEN_REQUIRE(!Dc || cmvn, EESEN_ERR_INVALID, "the pipeline has a CMVN stage: per-utterance offset/scale vectors are required");
    // dimensions and frame counts at every stage boundary
    std::vector<int> dims(nb);
    dims[0] = D;
    for (int k = 0; k < nst; ++k) dims[k + 1] = stage_dim(pipe[k], dims[k]);
    const int id = next;
    next = (next + 1) % (int)slots.size();
    Slot& sl = slots[id];
    if (sl.in_flight) EESEN_HIP_CHECK(hipEventSynchronize(sl.ready));
    if (sl.has_consumer) EESEN_HIP_CHECK(hipEventSynchronize(sl.consumed));
    sl.in_flight = sl.has_consumer = false;
    if ((size_t)nb * S > sl.meta2_cap) {
      if (sl.meta_h) EESEN_HIP_CHECK(hipHostFree(sl.meta_h));
      sl.meta_h = nullptr;
      sl.meta2_cap = (size_t)nb * S;
      EESEN_HIP_CHECK(hipHostMalloc(reinterpret_cast<void**>(&sl.meta_h), sl.meta2_cap * (sizeof(long) + sizeof(int)), hipHostMallocDefault));
    }
    long* off_h = sl.meta_h;
    int* fr_h = reinterpret_cast<int*>(sl.meta_h + sl.meta2_cap);
    std::vector<long> total(nb, 0);
    int T = 0;
    for (int s = 0; s < S; ++s) {
      EESEN_REQUIRE(frames[s] >= 0, EESEN_ERR_INVALID, "negative frame count");
      EESEN_REQUIRE(frames[s] == 0 || utts[s] != nullptr, EESEN_ERR_INVALID, "null utterance matrix");
      EESEN_REQUIRE(!strides || frames[s] == 0 || strides[s] >= D, EESEN_ERR_INVALID, "row stride smaller than the feature dimension");
      EESEN_REQUIRE(!Dc || frames[s] == 0 || cmvn[s] != nullptr, EESEN_ERR_INVALID, "null CMVN vectors");
      int f = frames[s];
      for (int b = 0; b < nb; ++b) {
        if (b) f = stage_frames(pipe[b - 1], f);
        off_h[(size_t)b * S + s] = total[b];
        fr_h[(size_t)b * S + s] = f;
        total[b] += (long)f * dims[b];
      }
      T = std::max(T, f);
    }
    EESEN_REQUIRE(T > 0, EESEN_ERR_INVALID, "every utterance is empty after the front end");
    if ((size_t)total[0] > sl.host_cap) {
      if (sl.host) EESEN_HIP_CHECK(hipHostFree(sl.host));
      sl.host = nullptr;
      sl.host_cap = (size_t)total[0] + (size_t)total[0] / 4;
      EESEN_HIP_CHECK(hipHostMalloc(reinterpret_cast<void**>(&sl.host), sl.host_cap * sizeof(float), hipHostMallocDefault));
    }
    for (int s = 0; s < S; ++s) {  // pack the RAW matrices back to back
      const long o = off_h[s];
      const int st = strides ? strides[s] : D;
      if (st == D) std::memcpy(sl.host + o, utts[s], (size_t)frames[s] * D * sizeof(float));
      else for (int t = 0; t < frames[s]; ++t) std::memcpy(sl.host + o + (long)t * D, utts[s] + (long)t * st, (size_t)D * sizeof(float));
    }
    if (Dc) {
      if ((size_t)S * 2 * Dc > sl.cmvn_cap) {
        if (sl.cmvn_h) EESEN_HIP_CHECK(hipHostFree(sl.cmvn_h));
        sl.cmvn_h = nullptr;
        sl.cmvn_cap = (size_t)S * 2 * Dc;
        EESEN_HIP_CHECK(hipHostMalloc(reinterpret_cast<void**>(&sl.cmvn_h), sl.cmvn_cap * sizeof(float), hipHostMallocDefault));
      }
      for (int s = 0; s < S; ++s) {
        if (frames[s]) std::memcpy(sl.cmvn_h + (size_t)s * 2 * Dc, cmvn[s], (size_t)2 * Dc * sizeof(float));
        else std::memset(sl.cmvn_h + (size_t)s * 2 * Dc, 0, (size_t)2 * Dc * sizeof(float));
      }
      sl.cmvn_d.reserve((size_t)S * 2 * Dc);
      EESEN_HIP_CHECK(hipMemcpyAsync(sl.cmvn_d.p, sl.cmvn_h, (size_t)S * 2 * Dc * sizeof(float), hipMemcpyHostToDevice, copy));
    }
    // even boundaries live in `packed`, odd ones in `packed2`
    size_t cap_even = 0, cap_odd = 0;
    for (int b = 0; b < nb; ++b) (b % 2 ? cap_odd : cap_even) = std::max(b % 2 ? cap_odd : cap_even, (size_t)total[b]);
    sl.packed.reserve(std::max<size_t>(cap_even, 1));
    sl.packed2.reserve(std::max<size_t>(cap_odd, 1));
    sl.offs_d.reserve((size_t)nb * S);
    sl.frs_d.reserve((size_t)nb * S);
    const int Dout = dims[nst], ld = (Dout + 3) / 4 * 4;
    const bool fresh = sl.out.reserve((size_t)T * S * ld);
    if (ld != Dout && fresh) EESEN_HIP_CHECK(hipMemsetAsync(sl.out.p, 0, sl.out.cap * sizeof(float), copy));
    else if (ld != Dout) EESEN_HIP_CHECK(hipMemsetAsync(sl.out.p, 0, (size_t)T * S * ld * sizeof(float), copy));
    if (total[0]) EESEN_HIP_CHECK(hipMemcpyAsync(sl.packed.p, sl.host, (size_t)total[0] * sizeof(float), hipMemcpyHostToDevice, copy));
    EESEN_HIP_CHECK(hipMemcpyAsync(sl.offs_d.p, off_h, (size_t)nb * S * sizeof(long), hipMemcpyHostToDevice, copy));
    EESEN_HIP_CHECK(hipMemcpyAsync(sl.frs_d.p, fr_h, (size_t)nb * S * sizeof(int), hipMemcpyHostToDevice, copy));
    for (int k = 0; k < nst; ++k) {
      if (!total[k + 1]) continue;
      const float* src = k % 2 ? sl.packed2.p : sl.packed.p;
      float* dst = k % 2 ? sl.packed.p : sl.packed2.p;
      long biggest = 0;
      for (int s = 0; s < S; ++s) biggest = std::max(biggest, (long)fr_h[(size_t)(k + 1) * S + s] * dims[k + 1]);
      const int bx = (int)std::min<long>(std::max<long>(cdivl(biggest, 256), 1), 2048);   // one element per thread: the stage is latency-bound otherwise
      const StageDev sd{pipe[k].kind, pipe[k].a, pipe[k].b};
      hipLaunchKernelGGL(feat_stage_kernel, dim3(bx, S), dim3(256), 0, copy, sd, scales_d.p ? scales_d.p + pipe[k].scale_off : nullptr, src,
                         sl.offs_d.p + (size_t)k * S, sl.frs_d.p + (size_t)k * S, dims[k], dst, sl.offs_d.p + (size_t)(k + 1) * S,
                         sl.frs_d.p + (size_t)(k + 1) * S, dims[k + 1], sl.cmvn_d.p);
      check_launch("feat_stage_kernel");
    }
    const float* last = nst % 2 ? sl.packed2.p : sl.packed.p;
    const bool vec = Dout % 4 == 0;
    const long n = (long)T * S * (vec ? Dout / 4 : Dout);
    const int blocks = (int)std::min<long>(cdivl(n, 256), 256L * 32);
    if (vec) interleave_kernel<4><<<blocks, 256, 0, copy>>>(last, sl.offs_d.p + (size_t)nst * S, sl.frs_d.p + (size_t)nst * S, sl.out.p, T, S, Dout, ld);
    else interleave_kernel<1><<<blocks, 256, 0, copy>>>(last, sl.offs_d.p + (size_t)nst * S, sl.frs_d.p + (size_t)nst * S, sl.out.p, T, S, Dout, ld);
    check_launch("interleave_kernel");
    EESEN_HIP_CHECK(hipEventRecord(sl.ready, copy));
    sl.in_flight = true;
    sl.T = T; sl.S = S; sl.D = Dout; sl.ld = ld;
    return id;
  }

  Slot& checked(int id) {
    EESEN_REQUIRE(id >= 0 && id < (int)slots.size(), EESEN_ERR_INVALID, "slot id out of range");
    EESEN_REQUIRE(slots[id].in_flight, EESEN_ERR_STATE, "slot holds no submitted batch");
    return slots[id];
  }
  void acquire(int id, float** feats, int* T, int* S, int* ld) {
    Slot& sl = checked(id);
    EESEN_HIP_CHECK(hipSetDevice(device));
    EESEN_HIP_CHECK(hipStreamWaitEvent(compute, sl.ready, 0));  // device-side wait: the host does not block
    *feats = sl.out.p; *T = sl.T; *S = sl.S; *ld = sl.ld;
  }
  void release(int id) {  // everything enqueued on the compute stream so far may read the slot; later work may not
    Slot& sl = checked(id);
    EESEN_HIP_CHECK(hipSetDevice(device));
    EESEN_HIP_CHECK(hipEventRecord(sl.consumed, compute));
    sl.has_consumer = true;
  }
};

}  // namespace eesen

struct eesen_feeder : public eesen::Feeder { using eesen::Feeder::Feeder; };

using namespace eesen;

extern "C" {

int eesen_feeder_create(int device, void* compute_stream, int slots, eesen_feeder_t** out) {
  return guard([&] {
    REQ_PTR(out);
    *out = new eesen_feeder(device, compute_stream, slots);
  });
}
int eesen_feeder_destroy(eesen_feeder_t* f) {
  return guard([&] { delete f; });
}
int eesen_feeder_submit(eesen_feeder_t* f, const float* const* utts, const int* frames, const int* strides, int S, int D, int* slot) {
  return guard([&] {
    REQ_PTR(f); REQ_PTR(utts); REQ_PTR(frames); REQ_PTR(slot);
    *slot = f->submit(utts, frames, strides, S, D);
  });
}
int eesen_feeder_set_pipeline(eesen_feeder_t* f, const eesen_feat_stage_t* stages, int n_stages) {
  return guard([&] {
    REQ_PTR(f);
    if (n_stages > 0) REQ_PTR(stages);
    f->set_pipeline(stages, n_stages);
  });
}
int eesen_feeder_pipeline_shape(eesen_feeder_t* f, int D_in, int frames_in, int* D_out, int* frames_out) {
  return guard([&] {
    REQ_PTR(f);
    if (D_out) *D_out = f->pipeline_dim(D_in);
    if (frames_out) *frames_out = f->pipeline_frames(frames_in);
  });
}
int eesen_feeder_submit_raw(eesen_feeder_t* f, const float* const* utts, const int* frames, const int* strides,
                            const float* const* cmvn, int S, int D_in, int* slot) {
  return guard([&] {
    REQ_PTR(f); REQ_PTR(utts); REQ_PTR(frames); REQ_PTR(slot);
    *slot = f->submit_raw(utts, frames, strides, cmvn, S, D_in);
  });
}
// the arithmetic of ApplyCmvn, /root/reference/src/feat/cmvn.cc:78-108: host-only (a few dozen doubles per speaker)
int eesen_cmvn_norm(const double* stats, int rows, int cols, int norm_vars, float* offset_scale) {
  return guard([&] {
    REQ_PTR(stats); REQ_PTR(offset_scale);
    const int dim = cols - 1;
    EESEN_REQUIRE(rows >= 1 && rows <= 2 && dim >= 1, EESEN_ERR_INVALID, "Dim mismatch in ApplyCmvn: cmvn stats must be 1 or 2 rows of dim + 1 columns");
    EESEN_REQUIRE(!(rows == 1 && norm_vars), EESEN_ERR_INVALID, "You requested variance normalization but no variance stats are supplied.");
    const double count = stats[dim];
    EESEN_REQUIRE(count >= 1.0, EESEN_ERR_INVALID, "Insufficient stats for cepstral mean and variance normalization: count = " + std::to_string(count));
    for (int d = 0; d < dim; ++d) {
      double mean = stats[d] / count, offset, scale;
      if (!norm_vars) {
        scale = 1.0;
        offset = -mean;
      } else {
        double var = (stats[(size_t)cols + d] / count) - mean * mean;
        const double floor = 1.0e-20;
        if (var < floor) var = floor;
        scale = 1.0 / std::sqrt(var);
        EESEN_REQUIRE(scale == scale && 1 / scale != 0.0, EESEN_ERR_INVALID, "NaN or infinity in cepstral mean/variance computation");
        offset = -(mean * scale);
      }
      offset_scale[d] = (float)offset;
      offset_scale[dim + d] = (float)scale;
    }
  });
}
int eesen_feeder_acquire(eesen_feeder_t* f, int slot, float** feats_dev, int* T, int* S, int* ld) {
  return guard([&] {
    REQ_PTR(f); REQ_PTR(feats_dev); REQ_PTR(T); REQ_PTR(S); REQ_PTR(ld);
    f->acquire(slot, feats_dev, T, S, ld);
  });
}
int eesen_feeder_release(eesen_feeder_t* f, int slot) {
  return guard([&] { REQ_PTR(f); f->release(slot); });
}

}  // extern "C"
