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
#include <cstring>
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
  };
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
