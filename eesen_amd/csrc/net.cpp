// net.cpp -- host orchestration of the hot path: the replacement of eesen::Net
// (/root/reference/src/net/net.{h,cc}) for the layer kinds on the path.
//
//   Net::Propagate      net.cc:67-86     -> Net::propagate
//   Net::Backpropagate  net.cc:88-108    -> Net::backpropagate (+ Net::update for the Update calls :101-104)
//   BiLstmParallel      bilstm-parallel-layer.h:379-420 (fwd), :881-913 (bwd); Lstm twin lstm-parallel-layer.h
//   AffineTransform     affine-trans-layer.h:161-219;  Softmax softmax-layer.h:44-57
//
// Differences by design: one input GEMM for both directions, gate-interleaved G layout,
// everything asynchronous on one HIP stream, fresh gradients kept apart from the momentum buffer until
// update() so the data-parallel all-reduce can sit between backpropagate() and update().
#include "net.h"

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>

namespace eesen {


// ------------------------------------------------------------------------------------------ PhaseTimer
PhaseTimer::~PhaseTimer() {
  for (auto& s : spans_) {
    (void)hipEventDestroy(s.a);
    (void)hipEventDestroy(s.b);
  }
}
int PhaseTimer::begin(hipStream_t st, int phase) {
  if (!on_) return -1;
  if (used_ == spans_.size()) {
    Span s;
    EESEN_HIP_CHECK(hipEventCreate(&s.a));
    EESEN_HIP_CHECK(hipEventCreate(&s.b));
    spans_.push_back(s);
  }
  spans_[used_].phase = phase;
  EESEN_HIP_CHECK(hipEventRecord(spans_[used_].a, st));
  return (int)used_++;
}
void PhaseTimer::end(hipStream_t st, int idx) {
  if (!on_ || idx < 0) return;
  EESEN_HIP_CHECK(hipEventRecord(spans_[idx].b, st));
}
void PhaseTimer::collect(float* out, int nphase) {
  for (int i = 0; i < nphase; ++i) out[i] = 0.f;
  for (size_t i = 0; i < used_; ++i) {
    EESEN_HIP_CHECK(hipEventSynchronize(spans_[i].b));
    float ms = 0.f;
    EESEN_HIP_CHECK(hipEventElapsedTime(&ms, spans_[i].a, spans_[i].b));
    if (spans_[i].phase >= 0 && spans_[i].phase < nphase) out[spans_[i].phase] += ms * 1e-3f;
  }
  if (accumulate_) used_ = 0;
}

int PhaseTimer::spans(int* phases, float* secs, int cap) {
  for (size_t i = 0; i < used_ && (int)i < cap; ++i) {
    EESEN_HIP_CHECK(hipEventSynchronize(spans_[i].b));
    float ms = 0.f;
    EESEN_HIP_CHECK(hipEventElapsedTime(&ms, spans_[i].a, spans_[i].b));
    if (phases) phases[i] = spans_[i].phase;
    if (secs) secs[i] = ms * 1e-3f;
  }
  return (int)used_;
}

// ------------------------------------------------------------------------------------------ Layer
long Layer::file_params() const {
  if (is_lstm()) return (long)ndir * ((long)4 * Hf * din_f + (long)4 * Hf * Hf + 4 * Hf + 3 * Hf);  // bilstm-layer.h:991-998
  if (kind == EESEN_LAYER_AFFINE) return (long)dout_f * din_f + dout_f;
  return 0;
}

const char* Layer::marker() const {
  switch (kind) {
    case EESEN_LAYER_BILSTM_PARALLEL: return nonparallel ? "<BiLstm>" : "<BiLstmParallel>";
    case EESEN_LAYER_LSTM_PARALLEL: return nonparallel ? "<Lstm>" : "<LstmParallel>";
    case EESEN_LAYER_AFFINE: return "<AffineTransform>";
    case EESEN_LAYER_SOFTMAX: return "<Softmax>";
    case EESEN_LAYER_SIGMOID: return "<Sigmoid>";
    case EESEN_LAYER_TANH: return "<Tanh>";
  }
  return "<Unknown>";
}

// Visits every parameter of a layer as (index in Net::GetParams order, offset in the internal block).
// File / GetParams order per LSTM direction: W_x [4H x D], W_m [4H x H], bias [4H], p_i, p_f, p_o [H]
// (bilstm-layer.h:478-492, 1000-1031), gate rows in g,i,f,o order.  Internal: rows gate-interleaved
// (row u*4+q), both directions' W_x stacked, W_x rows padded to a multiple of 4 floats.
template <class F>
static void for_each_param(const Layer& L, F f) {
  long fi = 0;
  if (L.is_lstm()) {   // (file side: Hf cells, din_f input columns; internal: H cells -- the padded ones are the LAST rows u*4+q, u >= Hf, of a direction)
    const int H = L.H, Hf = L.Hf, D = L.din_f, D4 = pad4(L.din);
    for (int dir = 0; dir < L.ndir; ++dir) {
      for (int r = 0; r < 4 * Hf; ++r) {
        const int q = r / Hf, u = r % Hf;
        for (int d = 0; d < D; ++d) f(fi++, L.off_wx + ((size_t)dir * 4 * H + u * 4 + q) * D4 + L.in_col(d));
      }
      for (int r = 0; r < 4 * Hf; ++r) {
        const int q = r / Hf, u = r % Hf;
        for (int k = 0; k < Hf; ++k) f(fi++, L.off_wm + ((size_t)dir * 4 * H + u * 4 + q) * H + k);
      }
      for (int r = 0; r < 4 * Hf; ++r) f(fi++, L.off_bias + (size_t)dir * 4 * H + (r % Hf) * 4 + r / Hf);
      for (int g = 0; g < 3; ++g)
        for (int u = 0; u < Hf; ++u) f(fi++, L.off_peep + ((size_t)dir * 3 + g) * H + u);
    }
  } else if (L.kind == EESEN_LAYER_AFFINE) {
    const int D4 = pad4(L.din);
    for (int r = 0; r < L.dout; ++r)
      for (int d = 0; d < L.din_f; ++d) f(fi++, L.off_w + (size_t)r * D4 + L.in_col(d));
    for (int r = 0; r < L.dout; ++r) f(fi++, L.off_b + r);
  }
}

// ------------------------------------------------------------------------------------------ Net
Net::Net(int dev, void* stream) : device(dev) {
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess || n <= 0)
    throw Error(EESEN_ERR_HIP, "no HIP device available: this library has no CPU fallback (hipGetDeviceCount: " +
                                   std::string(e == hipSuccess ? "0 devices" : hipGetErrorString(e)) + ")");
  EESEN_REQUIRE(dev >= 0 && dev < n, EESEN_ERR_INVALID, "device index out of range");
  EESEN_HIP_CHECK(hipSetDevice(dev));
  // NULL selects the device's default stream, so a Net and a Ctc created without a stream are ordered
  // against each other exactly like the reference's single-stream CuDevice.
  st = reinterpret_cast<hipStream_t>(stream);
  // side stream for work off the critical path (weight-gradient GEMMs), lowest priority so that the latency-bound
  // recurrence kernels of the main stream are dispatched first
  int lo = 0, hi = 0;
  EESEN_HIP_CHECK(hipDeviceGetStreamPriorityRange(&lo, &hi));
  EESEN_HIP_CHECK(hipStreamCreateWithPriority(&st2, hipStreamNonBlocking, lo));
  EESEN_HIP_CHECK(hipEventCreateWithFlags(&ev_rec, hipEventDisableTiming));
  for (auto& e : ev_grad) EESEN_HIP_CHECK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
  EESEN_HIP_CHECK(hipEventCreateWithFlags(&ev_gate_reset, hipEventDisableTiming));
  EESEN_HIP_CHECK(hipEventCreateWithFlags(&ev_gate_done, hipEventDisableTiming));
  tn = Tuning::from_env();   // every switch: tuning.h
  persistent = tn.persistent;
  // The next layer's input GEMM can run UNDER this layer's forward recurrence, gated tile by tile on the recurrence's arrival
  // counters (gemm_f32_nt_gated, bit-identical results).  Measured on cfg2: with the f32-MFMA GEMMs it pays (62.0 -> 60.0 ms per
  // step); with the 1.7x faster bf16-split GEMMs the spinning tiles cost the recurrence more than the 1.5 ms of GEMM per layer
  // they hide (50.5 ms gated, 48.9 not).  Default: gate only in f32 mode.
  gate_fwd = tn.gate_fwd >= 0 ? tn.gate_fwd != 0 : gemm_mode() == 0;
  // Weight-gradient GEMMs under the next layer's recurrence (side stream).  Measured on cfg2: with the one-launch-per-step
  // recurrence it is neutral (105.1 vs 104.9 ms per step); with the persistent recurrence, whose workgroups mostly wait on
  // hand-offs, it pays (78.3 -> 73.0 ms at the time).
  overlap = tn.overlap >= 0 ? tn.overlap != 0 : persistent != 0;
  spin_limit = tn.spin_limit;
  if (tn.trace) {
    trace.reserve(1280);
    EESEN_HIP_CHECK(hipMemset(trace.p, 0, 1280 * sizeof(unsigned long long)));
  }
  ctl.reserve(kCtlWords);
  EESEN_HIP_CHECK(hipMemset(ctl.p, 0, kCtlWords * sizeof(unsigned)));
  // First-poll delays of the recurrence kernels' hand-off waits.  Rounds 2-5 derived them from the MEASURED flight of an agent-scope
  // increment between two CUs of this device (lstm_persistent.hip: handoff_flight_ns) times a factor per wait kind (1.0 forward, 0.7
  // backward: round 2's hand-tuned s_sleep constants restated), so that a chip that clocks or routes differently gets proportionally
  // different delays.  EESEN_POLL_NS="fwd,bwd" overrides (experiments).
  {
    // Round 6 swept the delays again on the final kernels (profiles/r06_poll_sweep.log; cfg2, ms per step): forward 0 / 100 / 200 / 300 /
    // 400 / 600 ns at the best backward value 33.8 / 32.5 / 32.4 / 31.7 / 31.4 / 31.7; backward 0 / 100 / 200 / 280 / 420 / 500 / 600 /
    // 700 / 1000 ns at forward 400: 33.1 / 32.0 / 32.0 / 31.6 / 31.4 / 31.55 / 31.8 / 32.0 / 32.9 -- with the weight-gradient GEMMs of the
    // layer above on the side stream the increments take longer to land: 420 ns there, 280 where the recurrence has the chip to itself
    // (cfg2 at S = 64, cfg4, cfg5: at or within noise of the best of 140-560).  And the MEASUREMENT turned out to have two modes on this
    // part (370-420 and 550-620 ns: where the dispatcher puts the ping-pong's two workgroups; one box read the far mode in every process)
    // while the step does not care which one was read (fixed delays, readings of 398-591 ns: 36.0-36.15 ms): scaling the delays with
    // the reading made a far-mode process 0.3-0.9 ms slower per step for nothing.  So: the delays tuned on the MI355X, in ns, whenever the
    // reading is in the part's own range; scaled by reading / 400 only outside it (another clock, another fabric).
    constexpr float kFwdNs = 400.f, kBwdNs = 280.f, kBwdSideNs = 420.f, kRefFlightNs = 400.f;
    flight_ns = handoff_flight_ns();
    const float fscale = flight_ns >= 300.f && flight_ns <= 700.f ? 1.f : flight_ns / kRefFlightNs;
    auto ticks = [&](float ns) { return std::min(300, std::max(0, (int)std::lround(fscale * ns / 10.f))); };
    delay_fwd = ticks(kFwdNs); delay_bwd = ticks(kBwdNs); delay_bwd_side = ticks(kBwdSideNs);
    if (const char* e = tn.poll_ns) {
      int a = -1, b2 = -1;
      if (sscanf(e, "%d,%d", &a, &b2) == 2) { delay_fwd = a / 10; delay_bwd = delay_bwd_side = b2 / 10; poll_raw = 1; }
    }
    if (tn.print_flight)
      fprintf(stderr, "eesen_hip: increment flight %.0f ns; first-poll delays forward %d0, backward %d0 (beside side-stream GEMMs %d0) ns\n", flight_ns, delay_fwd, delay_bwd, delay_bwd_side);
  }
}

Net::~Net() {
  (void)hipSetDevice(device);
  (void)hipStreamSynchronize(st);
  for (Ctc* c : guards) {   // a Ctc that outlives this Net must not keep reading its (about to be freed) error word
    (void)hipStreamSynchronize(c->st);
    c->guard = nullptr;
    c->guard_net = nullptr;
  }
  if (trace.p) {  // EESEN_TRACE=1: timeline of workgroup 0 of the last persistent launches (shader-clock ticks)
    std::vector<unsigned long long> h(1280);
    if (hipMemcpy(h.data(), trace.p, h.size() * 8, hipMemcpyDeviceToHost) == hipSuccess)
      for (int pass = 0; pass < 2; ++pass) {
        double seg[5] = {0, 0, 0, 0, 0};
        int n = 0;
        for (int s = 8; s < 120; ++s) {
          const unsigned long long* a = h.data() + pass * 640 + s * 5;
          if (!a[0] || !a[5]) continue;
          seg[0] += (double)(a[1] - a[0]); seg[1] += (double)(a[2] - a[1]); seg[2] += (double)(a[3] - a[2]);
          seg[3] += (double)(a[4] - a[3]); seg[4] += (double)(a[5] - a[4]);
          ++n;
        }
        if (n) fprintf(stderr, "EESEN_TRACE %s: wait %.0f | A-load+MFMA+reduce %.0f | epilogue %.0f | drain+barrier %.0f | publish->next %.0f ticks/step (%d steps)\n",
                       pass ? "bwd" : "fwd", seg[0] / n, seg[1] / n, seg[2] / n, seg[3] / n, seg[4] / n, n);
      }
  }
  if (lens_pin) (void)hipHostFree(lens_pin);
  if (err_pin) (void)hipHostFree(err_pin);
  if (live_pin) (void)hipHostFree(live_pin);
  for (auto& h : in_stage) { if (h.p) (void)hipHostFree(h.p); if (h.ev) (void)hipEventDestroy(h.ev); }
  if (lens_ev) (void)hipEventDestroy(lens_ev);
  if (err_ev) (void)hipEventDestroy(err_ev);
  if (st2) { (void)hipStreamSynchronize(st2); (void)hipStreamDestroy(st2); }
  if (ev_rec) (void)hipEventDestroy(ev_rec);
  for (auto& e : ev_grad) if (e) (void)hipEventDestroy(e);
  if (ev_gate_reset) (void)hipEventDestroy(ev_gate_reset);
  if (ev_gate_done) (void)hipEventDestroy(ev_gate_done);
  for (auto& e : ev_ready) if (e) (void)hipEventDestroy(e);
  for (auto& e : ev_bucket) if (e) (void)hipEventDestroy(e);
  if (ev_bwd_done) (void)hipEventDestroy(ev_bwd_done);
  if (own_stream) (void)hipStreamDestroy(st);
}

void Net::sync() {
  EESEN_HIP_CHECK(hipSetDevice(device));
  EESEN_HIP_CHECK(hipStreamSynchronize(st));   // (with a communicator: returns at the latest when its watchdog aborts a lost collective)
  wait_buckets_host();  // gradient buckets still in flight on the communicator's stream
  comm_check_alive(comm);
  check_device_error(/*consumer=*/true);
}

// The error word of the persistent recurrence kernels.  While it is set every persistent kernel leaves at its first poll, the
// update kernels skip (optim.hip) and a guarded Ctc drops the minibatch from its statistics (eesen_ctc_set_guard), so nothing
// computed under it reaches the model or the accuracy: `lost` minibatches -- the one that raised it and those the host had
// already enqueued behind it -- are not applied.  The run continues on the one-launch-per-step kernels.  With a data-parallel
// communicator the other ranks HAVE stepped, so "not applied" would let the ranks diverge: there the failed rank's gradients enter
// the all-reduce as ZEROS (decided on the device, comm.cpp: bucket_allreduce) and it applies the same summed update as the others
// -- the run loses this rank's share of those minibatches and nothing else (round 3 raised an error here, for either value).
// consumer = the caller is about to READ something the last Propagate produced (get_output, Synchronize before a host copy,
// the inference and cross-validation paths): that forward pass is then re-run on the per-step kernels before returning, so a
// timed-out Propagate never hands out garbage with status OK.
void Net::check_device_error(bool consumer) {
  if (!ctl.p) return;
  unsigned e = 0;
  EESEN_HIP_CHECK(hipMemcpy(&e, ctl.p + kCtlWords - 1, sizeof(unsigned), hipMemcpyDeviceToHost));
  if (!e) { steps_since_clean = 0; return; }
  EESEN_HIP_CHECK(hipStreamSynchronize(st));
  EESEN_HIP_CHECK(hipMemset(ctl.p + kCtlWords - 1, 0, sizeof(unsigned)));
  // 2: only the side stream's wait for a forward milestone gave up (wait_for_word: something lets one kernel run at a time and ran the
  // waiter before the recurrence it waits for -- a counter-collecting profiler).  The early GEMM may have read unfinished rows:
  // the step is lost like any other, but the cure is to stop starting GEMMs early, not to give up the persistent kernels.
  const bool only_waiter = e == 2 && tn.fwd_mid;
  if (only_waiter) tn.fwd_mid = 0;
  else { persistent = 0; gate_fwd = false; overlap = false; }
  ++recoveries;
  const int lost = std::max(1, steps_since_clean);
  steps_since_clean = 0;
  const bool rerun = consumer && propagated && input.p && rows > 0;
  // In a data-parallel run the other ranks HAVE stepped: this rank's gradients of those minibatches went into the all-reduce as
  // zeros (comm.cpp: bucket_allreduce) and it applied the same summed update as everybody, so the models stay identical -- the
  // run loses this rank's share of those steps, nothing else.
  const char* what = comm ? "contributed a ZERO gradient to the data-parallel sum (the ranks' models stay identical)" : "were NOT applied";
  if (only_waiter)
    fprintf(stderr, "WARNING (eesen_hip) the side stream's wait for a forward-recurrence milestone gave up (kernels serialised by a tool?): "
                    "%d minibatch(es) in flight %s%s; continuing without the early input GEMM (EESEN_FWD_MID=0)\n", lost, what,
            rerun ? " and the last forward pass is re-run" : "");
  else
    fprintf(stderr, "WARNING (eesen_hip) a persistent recurrence kernel gave up waiting for a peer workgroup (GPU shared or preempted?): "
                    "%d minibatch(es) in flight %s%s; continuing with the one-launch-per-step kernels\n", lost, what,
            rerun ? " and the last forward pass is re-run" : "");
  if (rerun) {
    forward_pass();
    EESEN_HIP_CHECK(hipStreamSynchronize(st));
  }
}

void Net::arm_device_error_poll() {
  if (!ctl.p) return;
  if (!err_pin) {
    EESEN_HIP_CHECK(hipHostMalloc(reinterpret_cast<void**>(&err_pin), sizeof(unsigned), hipHostMallocDefault));
    *err_pin = 0;
    EESEN_HIP_CHECK(hipEventCreateWithFlags(&err_ev, hipEventDisableTiming));
  }
  EESEN_HIP_CHECK(hipMemcpyAsync(err_pin, ctl.p + kCtlWords - 1, sizeof(unsigned), hipMemcpyDeviceToHost, st));
  EESEN_HIP_CHECK(hipEventRecord(err_ev, st));
  err_armed = true;
}

void Net::poll_device_error() {
  if (!err_armed || hipEventQuery(err_ev) != hipSuccess) return;  // not there yet: the next poll or sync() will see it
  err_armed = false;
  if (*err_pin) check_device_error(/*consumer=*/false);  // re-reads the word, resets it and falls back
  else steps_since_clean = 0;
}

void Net::add_layer(int kind, int din, int dout, float coef, float max_grad) {   // din, dout: the FILE's dimensions
  EESEN_REQUIRE(!finalized, EESEN_ERR_STATE, "net already finalized");
  EESEN_REQUIRE(din > 0 && dout > 0, EESEN_ERR_INVALID, "layer dimensions must be positive");
  const Layer* prev = layers.empty() ? nullptr : &layers.back();
  if (prev)  // net.cc:282-286
    EESEN_REQUIRE(prev->dout_f == din, EESEN_ERR_INVALID, "Dimensionality mismatch between consecutive layers");
  Layer L;   // checked as a local: a refused layer leaves the net as it was
  L.kind = kind; L.coef = coef; L.max_grad = max_grad;
  L.din_f = din; L.dout_f = dout;
  L.din = prev ? prev->dout : din;                     // the internal width of what this layer reads ...
  if (prev) { L.in_nb = prev->out_nb; L.in_hf = prev->out_hf; L.in_hi = prev->out_hi; }   // ... and where the file's columns are in it
  L.dout = dout;
  const char* sees_pad = "directly on an LSTM layer whose cell count per direction is not a multiple of 4 is not supported: it would see the "
                         "zero cells the library pads such a layer with";
  switch (kind) {
    case EESEN_LAYER_BILSTM_PARALLEL:
      EESEN_REQUIRE(dout % 2 == 0, EESEN_ERR_INVALID, "<CellDim> of a BiLstm layer must be even");
      L.ndir = 2; L.Hf = dout / 2;
      break;
    case EESEN_LAYER_LSTM_PARALLEL:
      L.ndir = 1; L.Hf = dout;
      break;
    case EESEN_LAYER_AFFINE:
      break;
    case EESEN_LAYER_SOFTMAX:
      EESEN_REQUIRE(din == dout, EESEN_ERR_INVALID, "Softmax needs InputDim == OutputDim");
      if (L.in_nb) throw Error(EESEN_ERR_INVALID, std::string("a <Softmax> ") + sees_pad);
      break;
    case EESEN_LAYER_SIGMOID:
    case EESEN_LAYER_TANH:
      EESEN_REQUIRE(din == dout, EESEN_ERR_INVALID, "an activation layer needs InputDim == OutputDim");
      if (L.in_nb && kind == EESEN_LAYER_SIGMOID) throw Error(EESEN_ERR_INVALID, std::string("a <Sigmoid> ") + sees_pad + " (sigmoid(0) = 1/2)");
      L.dout = L.din;                                  // tanh(0) = 0: the padded columns pass through
      L.out_nb = L.in_nb; L.out_hf = L.in_hf; L.out_hi = L.in_hi;
      break;
    default:
      throw Error(EESEN_ERR_INVALID, "unsupported layer kind " + std::to_string(kind));
  }
  if (L.is_lstm()) {   // the kernels fetch the recurrent state four cells at a time: H = Hf rounded up, the extra cells identically zero
    L.H = pad4(L.Hf);
    L.dout = L.ndir * L.H;
    if (L.H != L.Hf) { L.out_nb = L.ndir; L.out_hf = L.Hf; L.out_hi = L.H; }
  }
  layers.push_back(std::move(L));
}

void Net::finalize() {
  EESEN_REQUIRE(!finalized, EESEN_ERR_STATE, "net already finalized");
  EESEN_REQUIRE(!layers.empty(), EESEN_ERR_INVALID, "empty net");
  EESEN_HIP_CHECK(hipSetDevice(device));
  size_t off = 0;
  for (Layer& L : layers) {
    L.p_off = off;
    size_t n = 0;
    if (L.is_lstm()) {
      const size_t H = L.H, D4 = pad4(L.din), nd = L.ndir;
      L.off_wx = n;   n += nd * 4 * H * D4;
      L.off_bias = n; n += nd * 4 * H;
      L.off_wm = n;   n += nd * 4 * H * H;
      L.off_peep = n; n += nd * 3 * H;
      L.WmT.reserve(nd * H * 4 * H);
    } else if (L.kind == EESEN_LAYER_AFFINE) {
      L.off_w = n; n += (size_t)L.dout * pad4(L.din);
      L.off_b = n; n += pad4(L.dout);
    }
    L.p_n = (n + 3) & ~(size_t)3;
    off += L.p_n;
  }
  P = off;
  if (P) {
    params.reserve(P); corr.reserve(P); fresh.reserve(P + kLiveWords);   // + the data-parallel liveness word (comm.cpp)
    EESEN_HIP_CHECK(hipMemsetAsync(params.p, 0, P * sizeof(float), st));
    EESEN_HIP_CHECK(hipMemsetAsync(corr.p, 0, P * sizeof(float), st));
    EESEN_HIP_CHECK(hipMemsetAsync(fresh.p, 0, (P + kLiveWords) * sizeof(float), st));
  }
  // Side-stream gradient GEMMs pay only while the backward recurrence leaves register room for a GEMM workgroup next to
  // it; the 16x16 tile of wide layers (H > 512: 256 VGPRs x 2 waves per SIMD) does not, and a GEMM that started before the
  // cooperative launch then only delays it (measured on cfg4: 141 ms overlapped, 138 ms not).  EESEN_OVERLAP overrides.
  if (tn.overlap < 0)
    for (const Layer& L : layers)
      if (L.is_lstm() && L.H > 512) overlap = false;
  amax.reserve(1 + layers.size());
  EESEN_HIP_CHECK(hipMemsetAsync(amax.p, 0, amax.cap * sizeof(float), st));
  EESEN_HIP_CHECK(hipMemsetD32Async(reinterpret_cast<hipDeviceptr_t>(amax.p), 0x3f800000, 1, st));   // word 0 = 1.0f
  finalized = true;
  refresh_derived();
  sync();
}

// one pass over a [rows x cols] matrix: the bounds of its rows and of its columns (gemm.hip: amax_rows_cols)
void Net::measure(const float* Pm, long nrows, int cols, int ld, DevBuf<float>& out_rows, DevBuf<float>& out_cols) {
  out_rows.reserve((size_t)nrows);
  out_cols.reserve((size_t)cols);
  amax_ws.reserve((size_t)kAmaxBlocks * std::min(16384, (cols + 3) & ~3));   // (one launch takes <= 16384 columns: amax_rows_cols)
  amax_rows_cols(st, Pm, nrows, cols, ld, out_rows.p, out_cols.p, amax_ws.p);
}
bool Net::x_is_bounded(int li) const {
  if (li == 0) return false;
  const Layer& Pv = layers[li - 1];   // behind an LSTM layer without forward dropout (mask = 1 / (1 - p)), or an activation / softmax layer
  return (Pv.is_lstm() && !Pv.cur_fwd_drop) || Pv.is_activation() || Pv.kind == EESEN_LAYER_SOFTMAX;
}
// the bounds of every weight matrix a GEMM multiplies with (mode 2 of gemm.hip) and max |W_m| (the fp16-plane recurrence), after a
// parameter change
void Net::ensure_weight_amax() {
  const bool mode2 = gemm_mode() == 2;
  const bool need_w = mode2 && !wamax_valid;
  if (wm_valid && !need_w) return;
  if (!wm_valid) EESEN_HIP_CHECK(hipMemsetAsync(am_wm(0), 0, layers.size() * sizeof(float), st));
  for (size_t li = 0; li < layers.size(); ++li) {
    Layer& L = layers[li];
    if (L.is_lstm()) {
      if (need_w) measure(params.p + L.p_off + L.off_wx, (long)L.ndir * 4 * L.H, pad4(L.din), pad4(L.din), L.bw.rows, L.bw.cols);
      if (!wm_valid) amax_abs_accumulate(st, params.p + L.p_off + L.off_wm, (long)L.ndir * 4 * L.H, L.H, L.H, am_wm((int)li));
    } else if (L.kind == EESEN_LAYER_AFFINE && need_w) measure(params.p + L.p_off + L.off_w, L.dout, pad4(L.din), pad4(L.din), L.bw.rows, L.bw.cols);
  }
  wm_valid = true;
  if (mode2) wamax_valid = true;
}

long Net::num_params() const {
  long n = 0;
  for (const Layer& L : layers) n += L.file_params();
  return n;
}

void Net::init_accu() {  // InitAdaBuffers (bilstm-layer.h:66-100, affine-trans-layer.h:74-81): zeros
  if (accu_init) return;
  accu.reserve(P);
  EESEN_HIP_CHECK(hipMemsetAsync(accu.p, 0, P * sizeof(float), st));
  accu_init = true;
}

void Net::refresh_derived() {
  wamax_valid = wm_valid = false;
  for (Layer& L : layers)
    if (L.is_lstm())
      for (int dir = 0; dir < L.ndir; ++dir)
        transpose2d(st, params.p + L.p_off + L.off_wm + (size_t)dir * 4 * L.H * L.H, 4 * L.H, L.H,
                    L.WmT.p + (size_t)dir * L.H * 4 * L.H);
}

int Net::tensor_moments(int which, int layer, double* out6_host, int cap_tensors) {
  EESEN_REQUIRE(finalized, EESEN_ERR_STATE, "net not finalized");
  EESEN_REQUIRE(layer >= 0 && layer < (int)layers.size(), EESEN_ERR_INVALID, "layer index out of range");
  EESEN_REQUIRE(which >= 0 && which <= 2, EESEN_ERR_INVALID, "which: 0 parameters, 1 momentum buffers, 2 accumulators");
  const Layer& L = layers[layer];
  struct Region { size_t off; long rows; int cols; long ld; int nb, hf, hi; };   // (nb, hf, hi): the column map of Layer::in_col
  std::vector<Region> reg;   // the FILE's entries of every tensor: the zero cells of a padded LSTM layer are no statistics
  if (L.is_lstm()) {
    const int H = L.H, Hf = L.Hf, D4 = pad4(L.din);
    for (int dir = 0; dir < L.ndir; ++dir) {
      reg.push_back({L.off_wx + (size_t)dir * 4 * H * D4, 4L * Hf, L.din_f, D4, L.in_nb, L.in_hf, L.in_hi});
      reg.push_back({L.off_wm + (size_t)dir * 4 * H * H, 4L * Hf, Hf, H, 0, 0, 0});
      reg.push_back({L.off_bias + (size_t)dir * 4 * H, 1, 4 * Hf, 4L * H, 0, 0, 0});
      for (int g = 0; g < 3; ++g) reg.push_back({L.off_peep + ((size_t)dir * 3 + g) * H, 1, Hf, H, 0, 0, 0});
    }
  } else if (L.kind == EESEN_LAYER_AFFINE) {
    reg.push_back({L.off_w, L.dout, L.din_f, pad4(L.din), L.in_nb, L.in_hf, L.in_hi});
    reg.push_back({L.off_b, 1, L.dout, L.dout, 0, 0, 0});
  }
  const int n = (int)reg.size();
  if (!out6_host || n == 0) return n;
  EESEN_REQUIRE(cap_tensors >= n, EESEN_ERR_INVALID, "moments buffer too small");
  EESEN_HIP_CHECK(hipSetDevice(device));
  if (which == 2) init_accu();
  sync();
  const float* base = (which == 0 ? params.p : which == 1 ? corr.p : accu.p) + L.p_off;
  DevBuf<double> d;
  d.reserve((size_t)n * 6 + 8);
  for (int i = 0; i < n; ++i)
    eesen::tensor_moments(st, base + reg[i].off, reg[i].rows, reg[i].cols, reg[i].ld, d.p + (size_t)i * 6, d.p + (size_t)n * 6, reg[i].nb, reg[i].hf, reg[i].hi);
  EESEN_HIP_CHECK(hipStreamSynchronize(st));
  EESEN_HIP_CHECK(hipMemcpy(out6_host, d.p, (size_t)n * 6 * sizeof(double), hipMemcpyDeviceToHost));
  return n;
}

static void upload_flat(Net& net, DevBuf<float>& dst, const float* host, long n) {
  EESEN_REQUIRE(net.finalized, EESEN_ERR_STATE, "net not finalized");
  EESEN_REQUIRE(n == net.num_params(), EESEN_ERR_INVALID, "parameter count mismatch");
  EESEN_HIP_CHECK(hipSetDevice(net.device));
  std::vector<float> h(net.P, 0.f);
  long base = 0;
  for (const Layer& L : net.layers) {
    for_each_param(L, [&](long fi, size_t io) { h[L.p_off + io] = host[base + fi]; });
    base += L.file_params();
  }
  net.sync();
  if (net.P) EESEN_HIP_CHECK(hipMemcpy(dst.p, h.data(), net.P * sizeof(float), hipMemcpyHostToDevice));
}

void Net::set_params(const float* host, long n) {
  upload_flat(*this, params, host, n);
  refresh_derived();
  sync();
}

void Net::set_accu(const float* host, long n) {
  init_accu();
  upload_flat(*this, accu, host, n);
}

void Net::get_flat(const DevBuf<float>& buf, float* host, long n) {
  EESEN_REQUIRE(finalized, EESEN_ERR_STATE, "net not finalized");
  EESEN_REQUIRE(n == num_params(), EESEN_ERR_INVALID, "parameter count mismatch");
  sync();
  std::vector<float> h(P);
  if (P) EESEN_HIP_CHECK(hipMemcpy(h.data(), buf.p, P * sizeof(float), hipMemcpyDeviceToHost));
  long base = 0;
  for (const Layer& L : layers) {
    for_each_param(L, [&](long fi, size_t io) { host[base + fi] = h[L.p_off + io]; });
    base += L.file_params();
  }
}

void Net::set_seq_lengths(const int* l, int s) {
  EESEN_REQUIRE(s > 0, EESEN_ERR_INVALID, "need at least one sequence");
  EESEN_HIP_CHECK(hipSetDevice(device));
  lens.assign(l, l + s);
  for (int v : lens) EESEN_REQUIRE(v >= 0, EESEN_ERR_INVALID, "negative sequence length");
  poll_device_error();
  lens_d.reserve(s);
  // Stream-ordered upload: kernels of the previous step that still read the old lengths are ahead of this copy on the
  // stream (the side stream was joined at the end of Backpropagate), so nothing has to be drained.
  if ((size_t)s > lens_pin_cap) {
    if (lens_pin) { EESEN_HIP_CHECK(hipStreamSynchronize(st)); EESEN_HIP_CHECK(hipHostFree(lens_pin)); }
    lens_pin = nullptr;
    lens_pin_cap = std::max<size_t>(64, (size_t)s * 2);
    EESEN_HIP_CHECK(hipHostMalloc(reinterpret_cast<void**>(&lens_pin), lens_pin_cap * sizeof(int), hipHostMallocDefault));
  }
  if (!lens_ev) EESEN_HIP_CHECK(hipEventCreateWithFlags(&lens_ev, hipEventDisableTiming));
  else EESEN_HIP_CHECK(hipEventSynchronize(lens_ev));  // the previous upload has left the staging buffer
  std::copy(lens.begin(), lens.end(), lens_pin);
  EESEN_HIP_CHECK(hipMemcpyAsync(lens_d.p, lens_pin, s * sizeof(int), hipMemcpyHostToDevice, st));
  EESEN_HIP_CHECK(hipEventRecord(lens_ev, st));
  S = s;
  propagated = false;
}

static LstmLayerDev lstm_view(const Net& net, const Layer& L) {
  LstmLayerDev d;
  d.T = net.T; d.S = net.S; d.H = L.H; d.ndir = L.ndir;
  d.G = L.G.p; d.C = L.C.p; d.Y = L.Y.p;
  d.X = L.X.p;
  d.Wm = net.params.p + L.p_off + L.off_wm;
  d.WmT = L.WmT.p;
  d.peep = net.params.p + L.p_off + L.off_peep;
  d.lens = net.lens_d.p;
  d.rmask = L.cur_drop_mode ? L.rmask.p : nullptr;
  d.drop_mode = L.cur_drop_mode;
  d.fwd_bf16 = net.fwd_bf16_rec ? 1 : 0;
  d.fwd_split = net.tn.fwd_split;
  d.fwd_f16 = net.tn.fwd_f16 && net.tn.fwd_split;   // (EESEN_FWD_SPLIT=0 is the master switch: the fp32-input MFMA kernels)
  d.bwd_f16 = net.tn.bwd_f16 && net.tn.fwd_split;
  d.wm_amax = net.amax.p ? const_cast<Net&>(net).am_wm((int)(&L - net.layers.data())) : nullptr;
  d.xcd_map = net.tn.xcd_map; d.fwd_mux = net.tn.fwd_mux; d.bwd_q4 = net.tn.bwd_q4; d.bwd_q4_st8 = net.tn.bwd_q4_st8; d.fwd_narrow2 = net.tn.fwd_narrow2; d.fwd_t16_small = net.tn.fwd_t16_small; d.bwd_ksplit = net.tn.bwd_ksplit; d.bwd_mux = net.tn.bwd_mux;
  return d;
}

// ---- dropout (bilstm-parallel-layer.h:46-94, 385-390) --------------------------------------------------------------------
void Net::set_layer_dropout(int layer, const float* nine) {
  EESEN_REQUIRE(layer >= 0 && layer < (int)layers.size(), EESEN_ERR_INVALID, "layer index out of range");
  Layer& L = layers[layer];
  bool any = false;
  for (int k = 0; k < 9; ++k) any |= nine[k] != 0.f;
  EESEN_REQUIRE(!any || L.kind == EESEN_LAYER_BILSTM_PARALLEL, EESEN_ERR_INVALID, "dropout options exist on BiLstm layers only");
  EESEN_REQUIRE(nine[0] >= 0.f && nine[0] < 1.f && nine[7] >= 0.f && nine[7] < 1.f, EESEN_ERR_INVALID, "dropout factor outside [0, 1)");
  for (int k = 0; k < 9; ++k) L.drop[k] = (k == 0 || k == 7) ? nine[k] : (nine[k] != 0.f ? 1.f : 0.f);
}

void Net::get_layer_dropout(int layer, float* nine) const {
  EESEN_REQUIRE(layer >= 0 && layer < (int)layers.size(), EESEN_ERR_INVALID, "layer index out of range");
  for (int k = 0; k < 9; ++k) nine[k] = layers[layer].drop[k];
}

// The masks cross the boundary at the width of the MODEL FILE's layer, ndir * Hf columns (what eesen_net_layer_info reports and
// include/eesen_hip.h documents), whatever the library pads the cell count to inside (a multiple of 4: Layer::H): the file's columns
// are scattered into / gathered out of the internal rows here, as forward_pass does for the layer output (a C caller
// sizing its buffers by the model's own H overflowed its heap on get and was refused on set).
static void widen_mask(const Layer& L, const float* src, long n, std::vector<float>& dst, const char* what) {
  dst.clear();
  if (!src) return;
  const long wf = (long)L.ndir * L.Hf, wi = (long)L.ndir * L.H;
  EESEN_REQUIRE(n > 0 && n % wf == 0, EESEN_ERR_INVALID, std::string(what) + ": the float count is not a multiple of the layer's ndir * H columns");
  if (wf == wi) { dst.assign(src, src + n); return; }
  const long rows = n / wf;
  dst.assign((size_t)rows * wi, 0.f);     // the padded cells are zero cells (zero weights, zero state): their mask never matters
  for (long r = 0; r < rows; ++r)
    for (int d = 0; d < L.ndir; ++d)
      std::copy(src + r * wf + (long)d * L.Hf, src + r * wf + (long)(d + 1) * L.Hf, dst.begin() + r * wi + (long)d * L.H);
}

void Net::set_dropout_masks(int layer, const float* fwd, long fwd_n, const float* rec, int rec_rows, long rec_n, int coin) {
  EESEN_REQUIRE(layer >= 0 && layer < (int)layers.size() && layers[layer].is_lstm(), EESEN_ERR_INVALID, "not an LSTM layer");
  Layer& L = layers[layer];
  if (rec) EESEN_REQUIRE(rec_rows > 0 && rec_n == (long)rec_rows * L.ndir * L.Hf, EESEN_ERR_INVALID, "recurrent mask: rec_floats must be rec_rows x ndir * H");
  widen_mask(L, fwd, fwd ? fwd_n : 0, L.inj_fmask, "forward mask");
  widen_mask(L, rec, rec ? rec_n : 0, L.inj_rmask, "recurrent mask");
  L.inj_rmask_rows = rec ? rec_rows : 0;
  L.inj_coin = coin;
}

void Net::get_dropout_masks(int layer, float* fwd_host, float* rec_host, int* info4) {
  EESEN_REQUIRE(layer >= 0 && layer < (int)layers.size() && layers[layer].is_lstm(), EESEN_ERR_INVALID, "not an LSTM layer");
  EESEN_REQUIRE(propagated, EESEN_ERR_STATE, "no Propagate yet");
  Layer& L = layers[layer];
  EESEN_HIP_CHECK(hipSetDevice(device));
  sync();
  const size_t ldY = (size_t)L.ndir * L.H, wf = (size_t)L.ndir * L.Hf;
  auto fetch = [&](float* host, const float* dev, size_t nrows) {   // the file's columns of an internal [nrows x ndir * H] mask
    if (wf == ldY) { EESEN_HIP_CHECK(hipMemcpy(host, dev, nrows * ldY * sizeof(float), hipMemcpyDeviceToHost)); return; }
    std::vector<float> tmp(nrows * ldY);
    EESEN_HIP_CHECK(hipMemcpy(tmp.data(), dev, tmp.size() * sizeof(float), hipMemcpyDeviceToHost));
    for (size_t r = 0; r < nrows; ++r)
      for (int d = 0; d < L.ndir; ++d)
        std::copy(tmp.begin() + r * ldY + (size_t)d * L.H, tmp.begin() + r * ldY + (size_t)d * L.H + L.Hf, host + r * wf + (size_t)d * L.Hf);
  };
  if (fwd_host && L.cur_fwd_drop) fetch(fwd_host, L.fmask.p, (size_t)rows);
  if (rec_host && L.cur_drop_mode) fetch(rec_host, L.rmask.p, (size_t)(T + 2) * S);
  if (info4) { info4[0] = L.cur_fwd_drop; info4[1] = L.cur_drop_mode; info4[2] = L.cur_twiddle_coin; info4[3] = (int)wf; }
}

// Decides what this Propagate applies to layer L and puts the masks in HBM: injected ones if the caller supplied them
// (one-shot), otherwise drawn on the device from (drop_seed, draw counter).
static void prepare_dropout(Net& net, Layer& L) {
  L.cur_fwd_drop = false; L.cur_drop_mode = 0; L.cur_twiddle_coin = false;
  const bool want = net.in_train && L.has_dropout();
  if (!want) { L.inj_fmask.clear(); L.inj_rmask.clear(); L.inj_rmask_rows = 0; L.inj_coin = -1; return; }
  const float fwd_p = L.drop[0], rec_p = L.drop[7];
  const bool fw_seq = L.drop[2] != 0.f, rec_step = L.drop[3] != 0.f, rec_seq = L.drop[4] != 0.f;
  const bool rnndrop = L.drop[5] != 0.f, nml = L.drop[6] != 0.f, twiddle = L.drop[8] != 0.f;
  const int T = net.T, S = net.S, ldY = L.ndir * L.H;
  bool coin = false;
  if (twiddle) {  // :385-387: one fair coin per Propagate decides between forward and recurrent dropout
    if (L.inj_coin >= 0) coin = L.inj_coin != 0;
    else {
      unsigned long long z = net.drop_seed + (++net.drop_counter) * 0x9E3779B97F4A7C15ull;
      z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull; z = (z ^ (z >> 27)) * 0x94D049BB133111EBull; z ^= z >> 31;
      coin = (z >> 63) != 0;
    }
  }
  L.cur_twiddle_coin = coin;
  const bool rec = (rnndrop || nml) && (!twiddle || !coin);   // :389
  const bool fwd = fwd_p > 0.f && (!twiddle || coin);         // :390
  if (rec) {
    EESEN_REQUIRE(rec_step != rec_seq, EESEN_ERR_INVALID,
                  "recurrent dropout needs exactly one of RecurrentTimeStepDropout / RecurrentSequenceDropout (bilstm-layer.h:102-113)");
    EESEN_REQUIRE(!(rnndrop && nml), EESEN_ERR_INVALID, "RNNDrop and NoMemLossDropout are mutually exclusive");
    const long mrows = (long)(T + 2) * S;
    L.rmask.reserve((size_t)mrows * ldY);
    if (!L.inj_rmask.empty()) {  // [(T+2)*S x ndir*H] as is, or [S x ndir*H] (sequence mask) repeated for every time step
      EESEN_REQUIRE((L.inj_rmask_rows == mrows || L.inj_rmask_rows == S) && (long)L.inj_rmask.size() == (long)L.inj_rmask_rows * ldY,
                    EESEN_ERR_INVALID, "injected recurrent mask has the wrong shape");
      const size_t chunk = (size_t)L.inj_rmask_rows * ldY;
      for (long r0 = 0; r0 < mrows; r0 += L.inj_rmask_rows)
        EESEN_HIP_CHECK(hipMemcpyAsync(L.rmask.p + (size_t)r0 * ldY, L.inj_rmask.data(), chunk * sizeof(float), hipMemcpyHostToDevice, net.st));
      EESEN_HIP_CHECK(hipStreamSynchronize(net.st));  // the host vector is released below
    } else {
      dropout_mask(net.st, L.rmask.p, mrows, ldY, ldY, rec_p, net.drop_seed + (++net.drop_counter) * 0x632BE59BD9B4E019ull, rec_seq);
    }
    L.cur_drop_mode = rnndrop ? 2 : 1;
  }
  if (fwd) {
    L.fmask.reserve((size_t)net.rows * ldY);
    L.Yd.reserve((size_t)net.rows * ldY);
    if (!L.inj_fmask.empty()) {
      EESEN_REQUIRE((long)L.inj_fmask.size() == (long)net.rows * ldY, EESEN_ERR_INVALID, "injected forward mask has the wrong shape");
      EESEN_HIP_CHECK(hipMemcpyAsync(L.fmask.p, L.inj_fmask.data(), L.inj_fmask.size() * sizeof(float), hipMemcpyHostToDevice, net.st));
      EESEN_HIP_CHECK(hipStreamSynchronize(net.st));
    } else {
      dropout_mask(net.st, L.fmask.p, net.rows, ldY, ldY, fwd_p, net.drop_seed + (++net.drop_counter) * 0x632BE59BD9B4E019ull, fw_seq);
    }
    L.cur_fwd_drop = true;
  }
  L.inj_fmask.clear(); L.inj_rmask.clear(); L.inj_rmask_rows = 0; L.inj_coin = -1;
}

void Net::propagate(const float* in, int nrows, int ld, bool is_device) {
  EESEN_REQUIRE(finalized, EESEN_ERR_STATE, "net not finalized");
  EESEN_REQUIRE(S > 0, EESEN_ERR_STATE, "SetSeqLengths must precede Propagate");
  EESEN_REQUIRE(nrows > 0 && nrows % S == 0, EESEN_ERR_INVALID, "row count must be a positive multiple of the sequence count");
  const int D = layers[0].din, D4 = pad4(D);
  EESEN_REQUIRE(ld >= D, EESEN_ERR_INVALID, "input leading dimension smaller than InputDim");
  EESEN_HIP_CHECK(hipSetDevice(device));
  rows = nrows;
  T = rows / S;
  for (int v : lens) EESEN_REQUIRE(v <= T, EESEN_ERR_INVALID, "sequence length exceeds the number of frames");
  timer.reset();

  // input -> device, rows padded to a multiple of 4 floats (GEMM operand alignment)
  if (input.reserve((size_t)rows * D4) || D4 != D) EESEN_HIP_CHECK(hipMemsetAsync(input.p, 0, (size_t)rows * D4 * sizeof(float), st));
  if (is_device) {
    EESEN_HIP_CHECK(hipMemcpy2DAsync(input.p, (size_t)D4 * sizeof(float), in, (size_t)ld * sizeof(float), (size_t)D * sizeof(float), rows,
                                     hipMemcpyDeviceToDevice, st));
  } else {
    // A HOST matrix is the caller's to free or overwrite the moment this call returns (the reference's CuMatrix constructor copies
    // synchronously, train-ctc-parallel.cc:198, and its trainer rebuilds feat_mat_host every minibatch): it is copied into one of
    // two pinned staging slots here and now, and travels from there on the stream.  (An asynchronous copy straight from pageable
    // memory may still be in flight when the caller reuses the buffer -- seen as run-to-run differences when two jobs share a GPU.)
    HostStage& hs = in_stage[in_stage_idx++ & 1];
    const size_t bytes = (size_t)rows * D * sizeof(float);
    if (!hs.ev) EESEN_HIP_CHECK(hipEventCreateWithFlags(&hs.ev, hipEventDisableTiming));
    if (hs.busy) { EESEN_HIP_CHECK(hipEventSynchronize(hs.ev)); hs.busy = false; }   // two Propagates ago
    if (bytes > hs.cap) {
      if (hs.p) EESEN_HIP_CHECK(hipHostFree(hs.p));
      hs.p = nullptr;
      hs.cap = bytes + bytes / 4;
      EESEN_HIP_CHECK(hipHostMalloc(reinterpret_cast<void**>(&hs.p), hs.cap, hipHostMallocDefault));
    }
    for (int r = 0; r < rows; ++r) std::memcpy(hs.p + (size_t)r * D, in + (size_t)r * ld, (size_t)D * sizeof(float));
    EESEN_HIP_CHECK(hipMemcpy2DAsync(input.p, (size_t)D4 * sizeof(float), hs.p, (size_t)D * sizeof(float), (size_t)D * sizeof(float), rows,
                                     hipMemcpyHostToDevice, st));
    EESEN_HIP_CHECK(hipEventRecord(hs.ev, st));
    hs.busy = true;
  }
  forward_pass();
}

// the layer chain of Net::Propagate on the input already in HBM (also the re-run after a timed-out persistent kernel)
void Net::forward_pass() {
  const float* x = input.p;
  int ldx = pad4(layers[0].din);
  ++steps_since_clean;
  info_fwd_persistent = info_lstm_layers = info_fwd_bf16 = 0;
  bool g_gated = false;  // the current layer's input GEMM was launched gated on the side stream
  int gated_rows = 0;    // ... for its first gated_rows rows (whole 128-row tiles); the rest is a plain GEMM
  int mid_r0 = 0, mid_r1 = 0;   // rows [mid_r0, mid_r1) of the current layer's input GEMM already ran on the side stream (see plan_mid)
  // two-plane fp16 GEMMs: the bounds of the current activation x and of the layer's weights (null in the other modes: unused)
  // (measured also when this pass itself runs on bf16-rounded operands: Backpropagate's GEMMs read them)
  const bool half = gemm_mode() == 2;
  ensure_weight_amax();
  amx_valid = half;
  for (Layer& L : layers) {
    const int li_ = (int)(&L - layers.data());
    GemmBound x_amax, w_amax;
    if (half && L.trainable()) {
      if (x_is_bounded(li_)) x_amax = bound_one();
      else { measure(x, rows, L.din, ldx, L.bx.rows, L.bx.cols); x_amax = GemmBound{L.bx.rows.p, 1}; }
      w_amax = GemmBound{L.bw.rows.p, 1};
    }
    if (L.is_lstm()) {
      const int H = L.H, nd = L.ndir, ldY = nd * H, ldG = nd * 4 * H;
      L.G.reserve((size_t)rows * ldG);
      const size_t state = (size_t)(T + 2) * S * ldY;
      L.C.reserve(state);
      L.Y.reserve(state);
      // exchange copy of Y in the persistent forward kernel's fetch order (LstmLayerDev::X)
      if (persistent && H % 32 == 0) L.X.reserve((size_t)T * nd * ((S + 15) / 16) * (size_t)(H / 32) * 768);   // (room for three bf16 planes)
      // boundary row blocks t = -1 and t = T (bilstm-parallel-layer.h:393-394)
      const size_t blk = (size_t)S * ldY * sizeof(float);
      EESEN_HIP_CHECK(hipMemsetAsync(L.C.p, 0, blk, st));
      EESEN_HIP_CHECK(hipMemsetAsync(L.Y.p, 0, blk, st));
      EESEN_HIP_CHECK(hipMemsetAsync(L.C.p + (size_t)(T + 1) * S * ldY, 0, blk, st));
      EESEN_HIP_CHECK(hipMemsetAsync(L.Y.p + (size_t)(T + 1) * S * ldY, 0, blk, st));
      prepare_dropout(*this, L);
      // all gate pre-activations of both directions in one GEMM: G = x * Wx^T + bias  (:109-110, :163-164)
      if (g_gated) {  // already computed on the side stream, gated on the previous layer's progress (see below)
        if (gated_rows < rows) {  // the last, partial row tile (T*S not a multiple of 128): its frames are the last to complete anyway
          const int ti_ = timer.begin(st, 0);
          gemm_f32(st, true, true, rows - gated_rows, ldG, L.din, 1.f, x + (size_t)gated_rows * ldx, ldx, params.p + L.p_off + L.off_wx,
                   pad4(L.din), 0.f, L.G.p + (size_t)gated_rows * ldG, ldG, params.p + L.p_off + L.off_bias, nullptr, 0, 0, fwd_bf16, x_amax, w_amax);
          timer.end(st, ti_);
        }
        EESEN_HIP_CHECK(hipStreamWaitEvent(st, ev_gate_done, 0));
        g_gated = false;
      } else if (mid_r1 > mid_r0) {  // the middle frames are done (or under way) on the side stream: the two ends here
        const int ti_ = timer.begin(st, 0);
        const int parts[2][2] = {{0, mid_r0}, {mid_r1, rows}};
        for (const auto& pr : parts)
          if (pr[1] > pr[0])
            gemm_f32(st, true, true, pr[1] - pr[0], ldG, L.din, 1.f, x + (size_t)pr[0] * ldx, ldx, params.p + L.p_off + L.off_wx, pad4(L.din),
                     0.f, L.G.p + (size_t)pr[0] * ldG, ldG, params.p + L.p_off + L.off_bias, nullptr, 0, 0, fwd_bf16, x_amax, w_amax);
        timer.end(st, ti_);
        EESEN_HIP_CHECK(hipStreamWaitEvent(st, ev_gate_done, 0));
        mid_r0 = mid_r1 = 0;
      } else {
        const int ti_ = timer.begin(st, 0);
        gemm_f32(st, true, true, rows, ldG, L.din, 1.f, x, ldx, params.p + L.p_off + L.off_wx, pad4(L.din), 0.f, L.G.p, ldG,
                 params.p + L.p_off + L.off_bias, nullptr, 0, 0, fwd_bf16, x_amax, w_amax);
        timer.end(st, ti_);
      }
      // The NEXT LSTM layer's input GEMM can run on the side stream WHILE this layer's persistent kernel is running:
      // its row tiles wait on the kernel's arrival counters and are visited middle-out in time (gemm_f32_nt_gated).
      Layer* nxt = (&L - layers.data()) + 1 < (long)layers.size() ? &layers[(&L - layers.data()) + 1] : nullptr;
      int gate_nblk = 0, nz = 1, gate_units = 0;
      lstm_fwd_persistent_geometry(lstm_view(*this, L), &gate_nblk, &nz, &gate_units);
      // (with forward dropout the next layer reads the MASKED output, which exists only after the recurrence: no gating)
      // (the 16-unit tile of wide layers fills the register file -- 2 x 206 VGPRs per SIMD -- so spinning GEMM workgroups
      // could keep its cooperative kernel from becoming resident: no gating there)
      const bool plan_gate = persistent && overlap && gate_fwd && !fwd_bf16 && !L.cur_fwd_drop && gate_units <= 8 && nxt && nxt->is_lstm() && T >= 2 && rows >= 128 &&
                             lstm_fwd_persistent_windows(lstm_view(*this, L)) == 1 &&
                             (nxt->ndir * 4 * nxt->H) % 128 == 0 && ldY % 16 == 0 && nd * nz * kShards <= 64;
      // The MIDDLE of the next layer's input GEMM under the END of this layer's recurrence.  A bidirectional layer has finished frame
      // t when its forward chain has passed step t and its backward chain step T-1-t: once both have published step m = 3T/4, the
      // frames [T-1-m, m] -- half of them -- are final, and the next layer's input GEMM for those rows runs on the side stream
      // while the last quarter of the recurrence (latency-bound: the matrix pipes are ~28 % busy) is still stepping.  The kernel
      // reports the milestone through a word in HBM (LstmLayerDev::milestone); a one-wave kernel on the side stream waits for it
      // (wait_for_word), and the main stream sets the word itself behind the recurrence, so the side stream is released even if
      // the kernel never reports (per-step fallback, a kernel that gave up).  Narrow tiles only: beside the wide tiles (H = 1024)
      // no GEMM workgroup fits on a CU (section 9), the early part would only queue.  Same GEMM, same rows: results are
      // bit-identical to the one-launch GEMM (every output row is its own dot products).
      const int mile_step = (3 * T) / 4;   // measured at cfg2, same box: 60 % 39.4, 67 % 38.7, 75 % 38.2, 82 % 38.85, 88 % 38.8, off 39.0 ms
      const bool plan_mid = persistent && overlap && tn.fwd_mid && !plan_gate && !L.cur_fwd_drop && lstm_fwd_persistent_leaves_room(lstm_view(*this, L)) && nd == 2 && nxt && nxt->is_lstm() &&
                            T >= 32 && mile_step + 1 < T && lstm_fwd_persistent_windows(lstm_view(*this, L)) == 1;
      // (Round 3 built, and round 4 re-tried as a profiling arm, a COMMAND-PROCESSOR wait instead of the spinning waiter --
      // hipStreamWaitValue64 on signal memory.  It costs 2.2 ms per cfg2 step, and it is NOT immune to kernel-serialising tools, as
      // had been assumed: under `rocprofv3 --pmc` the run dead-locked for the whole 40 minutes of a GPU call (the tool holds the
      // recurrence's dispatch back behind the side queue's pending wait packet), with no bound to give up at.  The spinning waiter
      // below gives up after a wall-clock bound and the run continues; counter passes set EESEN_FWD_MID=0.)
      { const int ti_ = timer.begin(st, 1);
      LstmLayerDev v = lstm_view(*this, L);
      v.poll_delay = delay_fwd;
      v.poll_raw = poll_raw;
      if (plan_mid) {
        mile.reserve(32);
        EESEN_HIP_CHECK(hipMemsetAsync(mile.p, 0, 2 * sizeof(unsigned), st));
        EESEN_HIP_CHECK(hipEventRecord(ev_gate_reset, st));
        v.milestone = mile.p;
        v.milestone_step = mile_step;
      }
      const bool pers = persistent && lstm_fwd_persistent(st, v, ctl.p, ctl.p + kCtlWords - 1, spin_limit, trace.p,
                                                          plan_gate ? ev_gate_reset : nullptr);
      if (!pers)
        for (int step = 0; step < T; ++step) lstm_fwd_step(st, v, step);
      info_fwd_persistent += pers ? 1 : 0;
      info_fwd_bf16 += pers && lstm_fwd_persistent_is_bf16(v) ? 1 : 0;
      ++info_lstm_layers;
      check_launch("lstm_fwd");
      timer.end(st, ti_);
      if (pers && plan_gate) {
        const int ldG2 = nxt->ndir * 4 * nxt->H;
        nxt->G.reserve((size_t)rows * ldG2);
        EESEN_HIP_CHECK(hipStreamWaitEvent(st2, ev_gate_reset, 0));
        const int tj_ = timer.begin(st2, 0);
        GemmGate gate{ctl.p, ctl.p + kCtlWords - 1, nd, nz, gate_nblk, T, S, spin_limit};
        gated_rows = rows / 128 * 128;
        // (this layer's output is the operand: no forward dropout here, so it is bounded by 1; the weights' word was written on `st`
        // before ev_gate_reset, which the side stream has just waited for)
        gemm_f32_nt_gated(st2, gated_rows, ldG2, ldY, L.Y.p + (size_t)S * ldY, ldY, params.p + nxt->p_off + nxt->off_wx, pad4(nxt->din),
                          nxt->G.p, ldG2, params.p + nxt->p_off + nxt->off_bias, gate, half ? bound_one() : GemmBound{}, half ? GemmBound{nxt->bw.rows.p, 1} : GemmBound{});
        timer.end(st2, tj_);
        EESEN_HIP_CHECK(hipEventRecord(ev_gate_done, st2));
        g_gated = true;
      }
      if (plan_mid) {
        EESEN_HIP_CHECK(hipMemsetD32Async(reinterpret_cast<hipDeviceptr_t>(mile.p + 1), 1, 1, st));   // released at the latest here
        const int ldG2 = nxt->ndir * 4 * nxt->H;
        nxt->G.reserve((size_t)rows * ldG2);
        // whole 256-row tiles: the two ends (main stream, critical path) keep the GEMM's 256 x 256 flavour; the middle part takes the
        // 128 x 128 flavour (256 threads), which shares a CU with the recurrence far better.  Measured at cfg2, same box, three runs
        // each: off 38.45; unaligned split (all parts small) 37.6-38.0; aligned, middle big 38.1-38.6; aligned, middle small 37.55-37.7 ms
        mid_r0 = ((T - 1 - mile_step) * S + 255) / 256 * 256;
        mid_r1 = (mile_step + 1) * S / 256 * 256;
        if (mid_r1 <= mid_r0) mid_r0 = mid_r1 = 0;
      }
      if (plan_mid && mid_r1 > mid_r0) {
        const int ldG2 = nxt->ndir * 4 * nxt->H;
        EESEN_HIP_CHECK(hipStreamWaitEvent(st2, ev_gate_reset, 0));
        // bounded on the wall clock: 2 s for a recurrence of up to 1000 steps, longer for longer ones, tenfold under a communicator
        // (whose collectives may delay the recurrence's residency: set_comm raises spin_limit the same way)
        wait_for_word(st2, mile.p + 1, 1u, ctl.p + kCtlWords - 1, 2.0 * std::max(1.0, T / 1000.0) * std::max(1.0, spin_limit / 400000.0));
        const int tj_ = timer.begin(st2, 0);
        gemm_f32(st2, true, true, mid_r1 - mid_r0, ldG2, nxt->din, 1.f, L.Y.p + (size_t)S * ldY + (size_t)mid_r0 * ldY, ldY,
                 params.p + nxt->p_off + nxt->off_wx, pad4(nxt->din), 0.f, nxt->G.p + (size_t)mid_r0 * ldG2, ldG2,
                 params.p + nxt->p_off + nxt->off_bias, nullptr, 0, /* a token of extra LDS: the 128 x 128 flavour */ 64, fwd_bf16,
                 half ? bound_one() : GemmBound{}, half ? GemmBound{nxt->bw.rows.p, 1} : GemmBound{});
        timer.end(st2, tj_);
        EESEN_HIP_CHECK(hipEventRecord(ev_gate_done, st2));
      } }
      if (L.cur_fwd_drop)  // :414-417: the layer's output, not its recurrent state, is masked
        mul_elements(st, L.Y.p + (size_t)S * ldY, ldY, L.fmask.p, ldY, L.Yd.p, ldY, rows, ldY);
      x = L.output(S);
      ldx = ldY;
    } else if (L.kind == EESEN_LAYER_AFFINE) {
      const int ldo = pad4(L.dout);
      if (L.out.reserve((size_t)rows * ldo)) EESEN_HIP_CHECK(hipMemsetAsync(L.out.p, 0, L.out.cap * sizeof(float), st));   // (the whole allocation: pad columns of rows a later, longer minibatch uses)
      { const int ti_ = timer.begin(st, 2);
      gemm_f32(st, true, true, rows, L.dout, L.din, 1.f, x, ldx, params.p + L.p_off + L.off_w, pad4(L.din), 0.f, L.out.p, ldo,
               params.p + L.p_off + L.off_b, nullptr, 0, 0, fwd_bf16, x_amax, w_amax);
      timer.end(st, ti_); }
      x = L.out.p;
      ldx = ldo;
    } else if (L.is_activation()) {  // sigmoid-layer.h:44-46, tanh-layer.h:44-46
      const int ldo = pad4(L.dout);
      if (L.out.reserve((size_t)rows * ldo)) EESEN_HIP_CHECK(hipMemsetAsync(L.out.p, 0, L.out.cap * sizeof(float), st));   // (the whole allocation: pad columns of rows a later, longer minibatch uses)
      { const int ti_ = timer.begin(st, 2);
      activation_rows(st, L.kind == EESEN_LAYER_TANH, x, ldx, L.out.p, ldo, rows, L.dout);
      timer.end(st, ti_); }
      x = L.out.p;
      ldx = ldo;
    } else {  // Softmax
      const int ldo = pad4(L.dout);
      if (L.out.reserve((size_t)rows * ldo)) EESEN_HIP_CHECK(hipMemsetAsync(L.out.p, 0, L.out.cap * sizeof(float), st));   // (the whole allocation: pad columns of rows a later, longer minibatch uses)
      { const int ti_ = timer.begin(st, 2);
      softmax_rows(st, x, ldx, L.out.p, ldo, rows, L.dout);
      timer.end(st, ti_); }
      x = L.out.p;
      ldx = ldo;
    }
  }
  out_ptr = x;
  out_cols = layers.back().dout;
  out_ld = ldx;
  if (const Layer& Lb = layers.back(); Lb.out_nb) {   // a padded LSTM layer (or a Tanh over one) is the net's last layer (Seam 2: the only
    const int ldo = pad4(Lb.dout_f);                  // one): the caller gets the file's columns, run by run
    if (out_f.reserve((size_t)rows * ldo) && ldo != Lb.dout_f) EESEN_HIP_CHECK(hipMemsetAsync(out_f.p, 0, out_f.cap * sizeof(float), st));
    for (int b = 0; b < Lb.out_nb; ++b) copy2d(st, x + (size_t)b * Lb.out_hi, ldx, out_f.p + (size_t)b * Lb.out_hf, ldo, rows, Lb.out_hf);
    out_ptr = out_f.p; out_cols = Lb.dout_f; out_ld = ldo;
  }
  propagated = true;
  if (persistent) arm_device_error_poll();
}

void Net::backpropagate(const float* out_diff, int ldd, float* in_diff, int ldi) {
  bucket_log.clear();
  deferred_buckets.clear();
  try {
    backpropagate_impl(out_diff, ldd, in_diff, ldi);
  } catch (...) {
    fail_step_buckets();   // with a communicator: the peers are in this step and wait for every one of its buckets (comm.cpp)
    throw;
  }
}

void Net::backpropagate_impl(const float* out_diff, int ldd, float* in_diff, int ldi) {
  EESEN_REQUIRE(propagated, EESEN_ERR_STATE, "Backpropagate needs a preceding Propagate");
  EESEN_HIP_CHECK(hipSetDevice(device));
  int maxdim = 0, max_g = 0, max_y = 0;
  size_t need_ws = 0;
  for (const Layer& L : layers) {
    maxdim = std::max(maxdim, std::max(pad4(L.din), pad4(L.dout)));
    if (L.is_lstm()) {
      max_g = std::max(max_g, L.ndir * 4 * L.H);
      max_y = std::max(max_y, L.ndir * L.H);
      need_ws = std::max(need_ws, lstm_bias_peep_ws_floats(T, S, L.H, L.ndir));
    }
    need_ws = std::max(need_ws, col_sums_ws_floats(rows, L.dout));
  }
  need_ws = std::max(need_ws, (size_t)16 << 20);  // 64 MB of split-K slabs
  ws.reserve(need_ws);
  ws_floats = need_ws;   // (what this minibatch asked for, not the allocation: the GEMMs' split-K factor follows the workspace size, and
                         // an allocation that grew by half for an earlier, longer minibatch must not change this one's summation order)
  dA.reserve((size_t)rows * maxdim);
  dB.reserve((size_t)rows * maxdim);
  if (max_g) {
    DGb[0].reserve((size_t)rows * max_g);
    DGb[1].reserve((size_t)rows * max_g);
  }
  if (max_y) DCF.reserve((size_t)S * max_y);
  ws2.reserve(need_ws);
  int dg_slot = 0;
  bool side_pending[2] = {false, false};
  deferred_buckets.clear();
  // Weight-gradient GEMMs on the side stream, under the next-lower layer's recurrence: decided per minibatch.  It pays beside the
  // small backward tiles (cfg2 at S = 32: 1.6 of 22.8 ms) and COSTS beside the 16-sequence tile a narrow layer takes at
  // --num-sequence 64 -- measured (round 5, cfg2 shape at S = 64, `profiles/r05_s64_step_timeline.txt`): the W_x-gradient GEMM ran
  // 14.1 ms beside the recurrence (3.8 alone) and, through the gate-gradient buffer it holds, stalled the main stream 3.5 ms per
  // layer: 79.8 ms per step overlapped, 71.4 not (with the two-tile 4 x 32 kernel there: 77.2 / 67.2).  At the recipes' 320 cells the
  // gradient GEMMs are 0.39 x the work and overlapping still pays at S = 64 (47.7 against 49.8 ms per minibatch): the rule stops at
  // H > 320.  EESEN_OVERLAP=1 forces it.
  const bool overlap = overlap_for_minibatch();
  exchange_deferred = comm && exchange_deferred_for_minibatch();
  bucket_log.clear();
  info_bwd_persistent = 0;
  live_valid = comm != nullptr;
  if (comm) {  // this rank has a minibatch: liveness 1 rides with the top layer's gradient bucket (comm.cpp)
    comm_check_alive(comm);
    EESEN_HIP_CHECK(hipMemsetD32Async(reinterpret_cast<hipDeviceptr_t>(fresh.p + P), 0x3f800000, 1, st));
  }

  // backpropagate_buf_[L] = out_diff (net.cc:96), into a buffer whose rows are 16-byte aligned
  const int Kout = layers.back().dout;
  float* d = dA.p;
  int ld_d = pad4(Kout);
  float* dn = dB.p;
  // two-plane fp16 GEMMs: operand bounds (weights: ensure_weight_amax; layer inputs: what Propagate measured; gradients: measured here)
  const bool half = gemm_mode() == 2;
  ensure_weight_amax();
  auto x_cols = [&](int li) -> GemmBound {   // the layer's input as the B operand of its weight-gradient GEMM: per input column
    if (!half || !amx_valid) return GemmBound{};   // (the mode changed between Propagate and here: measured at the call, gemm.hip)
    return x_is_bounded(li) ? bound_one() : GemmBound{layers[li].bx.cols.p, 1};
  };
  if (const Layer& Lb = layers.back(); Lb.out_nb) {   // the caller's out_diff has the file's columns: the padded cells get zeros
    EESEN_HIP_CHECK(hipMemsetAsync(d, 0, (size_t)rows * ld_d * sizeof(float), st));
    for (int b = 0; b < Lb.out_nb; ++b) copy2d(st, out_diff + (size_t)b * Lb.out_hf, ldd, d + (size_t)b * Lb.out_hi, ld_d, rows, Lb.out_hf);
  } else {
    if (ld_d != Kout) EESEN_HIP_CHECK(hipMemsetAsync(d, 0, (size_t)rows * ld_d * sizeof(float), st));
    copy2d(st, out_diff, ldd, d, ld_d, rows, Kout);
  }

  for (int li = (int)layers.size() - 1; li >= 0; --li) {
    Layer& L = layers[li];
    // this layer's input activation
    const float* x;
    int ldx;
    if (li == 0) { x = input.p; ldx = pad4(L.din); }
    else {
      const Layer& Pv = layers[li - 1];
      if (Pv.is_lstm()) { ldx = Pv.ndir * Pv.H; x = Pv.output(S); }
      else { ldx = pad4(Pv.dout); x = Pv.out.p; }
    }
    const bool want_in = li > 0 || in_diff != nullptr;
    const int ld_n = pad4(L.din);
    float* fr = fresh.p + L.p_off;
    if (L.kind == EESEN_LAYER_SOFTMAX) {
      continue;  // softmax-layer.h:49-57: CTC already delivers d/d(logits)
    } else if (L.is_activation()) {  // in_diff = out_diff * f'(y), from the layer's OUTPUT (sigmoid-layer.h:48-51, tanh-layer.h:48-51); same shape: in place
      { const int ti_ = timer.begin(st, 4);
      activation_diff_rows(st, L.kind == EESEN_LAYER_TANH, L.out.p, pad4(L.dout), d, ld_d, rows, L.dout);
      timer.end(st, ti_); }
      continue;
    } else if (L.kind == EESEN_LAYER_AFFINE) {
      { const int ti_ = timer.begin(st, 4);
      if (half) measure(d, rows, L.dout, ld_d, L.bd.rows, L.bd.cols);
      if (want_in) {  // in_diff = out_diff * W  (affine-trans-layer.h:171)
        if (ld_n != L.din) EESEN_HIP_CHECK(hipMemsetAsync(dn, 0, (size_t)rows * ld_n * sizeof(float), st));
        gemm_f32(st, true, false, rows, L.din, L.dout, 1.f, d, ld_d, params.p + L.p_off + L.off_w, pad4(L.din), 0.f, dn, ld_n,
                 nullptr, nullptr, 0, 0, false, half ? GemmBound{L.bd.rows.p, 1} : GemmBound{}, half ? GemmBound{L.bw.cols.p, 1} : GemmBound{});
      }
      // gradients (computed inside Update in the reference, affine-trans-layer.h:182-183)
      gemm_f32(st, false, false, L.dout, L.din, rows, 1.f, d, ld_d, x, ldx, 0.f, fr + L.off_w, pad4(L.din), nullptr, ws.p, ws_floats, 0, false,
               half ? GemmBound{L.bd.cols.p, 1} : GemmBound{}, x_cols(li));
      col_sums(st, d, rows, L.dout, ld_d, fr + L.off_b, ws.p, ws_floats);
      timer.end(st, ti_); }
      bucket_allreduce(li, st);
    } else {
      const int H = L.H, nd = L.ndir, ldG = nd * 4 * H, ldY = nd * H;
      LstmLayerDev v = lstm_view(*this, L);
      v.poll_delay = overlap ? delay_bwd_side : delay_bwd;   // (beside side-stream GEMMs the hand-off's increments land later: see the constructor)
      v.poll_raw = poll_raw;
      if (persistent) {  // wide layers: partial-sum exchange space of the K-split backward kernel (shared by the layers: their passes are serial)
        const size_t need = lstm_bwd_ksplit_px_floats(v);
        if (need) { bwd_px.reserve(need); v.PX = bwd_px.p; v.px_floats = bwd_px.cap; }
        if (const size_t ex = lstm_bwd_planes_ex_floats(v)) {   // the fp16-plane form: planes of the gate gradients + inverse powers (shared by the layers)
          bwd_dgh.reserve((size_t)rows * ldG);
          bwd_ex.reserve(ex);
          v.DGH = reinterpret_cast<unsigned char*>(bwd_dgh.p);
          v.EX = bwd_ex.p;
        }
      }
      EESEN_REQUIRE(in_train || !L.has_dropout(), EESEN_ERR_STATE, "Can't backpropagate a dropout layer in test mode (bilstm-parallel-layer.h:425)");
      if (L.cur_fwd_drop) mul_elements(st, d, ld_d, L.fmask.p, ldY, d, ld_d, rows, ldY);  // out_diff_drop, :892-896
      // Gate-gradient buffers alternate between LSTM layers: while this layer's weight-gradient GEMMs (side stream)
      // still read DGb[slot], the next-lower layer's recurrence (main stream) already fills the other one.
      float* DGl = DGb[dg_slot].p;
      if (side_pending[dg_slot]) {  // the layer two LSTM layers up used this buffer: its gradient GEMMs must be done
        EESEN_HIP_CHECK(hipStreamWaitEvent(st, ev_grad[dg_slot], 0));
        side_pending[dg_slot] = false;
      }
      // ("The middle first" of the forward pass does not pay here: with the input-gradient GEMM's middle rows on a third stream from
      // step 3T/4 of this recurrence, beside the weight-gradient GEMMs already co-running on the side stream, the cfg2 step went
      // 38.4 -> 39.4-39.5 ms on the same box -- the recurrence loses more to the third contender than the GEMM's head start gains.)
      { const int ti_ = timer.begin(st, 3);
      if (persistent && lstm_bwd_persistent(st, v, d, ld_d, DGl, ctl.p + kCtlHalf, ctl.p + kCtlWords - 1, spin_limit, trace.p ? trace.p + 640 : nullptr))
        ++info_bwd_persistent;
      else
        for (int step = 0; step < T; ++step) lstm_bwd_step(st, v, step, d, ld_d, DGl, DCF.p);
      check_launch("lstm_bwd");
      timer.end(st, ti_); }
      // one pass over the gate gradients for the (up to four) GEMMs that multiply them: per frame (input gradient) and per gate column
      if (half) { const int ti_ = timer.begin(st, 4); measure(DGl, rows, ldG, ldG, L.bd.rows, L.bd.cols); timer.end(st, ti_); }
      EESEN_HIP_CHECK(hipEventRecord(ev_rec, st));
      if (want_in) {  // in_diff = DGIFO_fw * Wx_fw + DGIFO_bw * Wx_bw  (:502, :593) as one K = ndir*4H contraction
        const int ti_ = timer.begin(st, 4);
        if (ld_n != L.din) EESEN_HIP_CHECK(hipMemsetAsync(dn, 0, (size_t)rows * ld_n * sizeof(float), st));
        gemm_f32(st, true, false, rows, L.din, ldG, 1.f, DGl, ldG, params.p + L.p_off + L.off_wx, pad4(L.din), 0.f, dn, ld_n,
                 nullptr, nullptr, 0, 0, false, half ? GemmBound{L.bd.rows.p, 1} : GemmBound{}, half ? GemmBound{L.bw.cols.p, 1} : GemmBound{});
        timer.end(st, ti_);
        // The input-gradient GEMM is on the critical path (the next-lower recurrence waits for it), the weight-gradient GEMMs
        // are not: they start behind it instead of beside it (measured: step 44.15 -> 42.97 ms)
        EESEN_HIP_CHECK(hipEventRecord(ev_rec, st));
      }
      // Everything that only feeds the parameter gradients leaves the critical path: it runs on the side stream,
      // under the next layer's (latency-bound, mostly idle-chip) recurrence.
      hipStream_t sg = overlap ? st2 : st;
      // Unused dynamic LDS caps the side-stream GEMM's occupancy, so that the next layer's cooperative recurrence kernel (one
      // 512-thread workgroup on EVERY CU) never waits for long-running GEMM tiles to retire.  32 KB = two GEMM workgroups per
      // CU.  Measured on cfg2 (3 runs of 20 steps each, same box): uncapped 58.0, 48 KB (one per CU) 55.3, 32 KB 54.7 ms/step --
      // with one per CU the side stream itself became the critical path (12 ms of gradient GEMMs per layer against 9 ms of
      // recurrence + input-gradient GEMM); before the recurrence kernels overlapped fetch and MFMA the ranking was the reverse.
      // (with the bf16-split GEMM: one per CU -- 47.6 vs 48.9 ms/step; the gradient GEMMs are short enough not to become the critical path)
      const int side_lds_env = (tn.side_lds_kb >= 0 ? tn.side_lds_kb : (gemm_mode() >= 1 ? 48 : 32)) * 1024;
      bool lstm_below = false;   // the cap protects the NEXT-LOWER recurrence's cooperative launch: the lowest LSTM layer's
      for (int lj = 0; lj < li; ++lj) lstm_below |= layers[lj].is_lstm();   // gradient GEMMs have the chip to themselves
      const int side_lds = overlap && lstm_below ? side_lds_env : 0;  // occupancy cap of the side-stream GEMMs
      if (overlap) EESEN_HIP_CHECK(hipStreamWaitEvent(st2, ev_rec, 0));
      { const int ti_ = timer.begin(sg, 4);
      // W_x gradient, both directions stacked: DGIFO^T * x  (:505, :596)
      gemm_f32(sg, false, false, ldG, L.din, rows, 1.f, DGl, ldG, x, ldx, 0.f, fr + L.off_wx, pad4(L.din), nullptr, ws2.p, need_ws, side_lds, false,
               half ? GemmBound{L.bd.cols.p, 1} : GemmBound{}, x_cols(li));
      // W_m gradient per direction: DGIFO^T * m shifted one step toward the recurrence source (:506, :597)
      for (int dir = 0; dir < nd; ++dir)
        gemm_f32(sg, false, false, 4 * H, H, rows, 1.f, DGl + (size_t)dir * 4 * H, ldG,
                 L.Y.p + (size_t)(dir == 0 ? 0 : 2 * S) * ldY + (size_t)dir * H, ldY, 0.f,
                 fr + L.off_wm + (size_t)dir * 4 * H * H, H, nullptr, ws2.p, need_ws, side_lds, false,
                 half ? GemmBound{L.bd.cols.p + (size_t)dir * 4 * H, 1} : GemmBound{}, half ? bound_one() : GemmBound{});
      lstm_bias_peep_grads(sg, v, DGl, fr + L.off_bias, fr + L.off_peep, ws2.p, need_ws);
      timer.end(sg, ti_); }
      bucket_allreduce(li, sg);  // this layer's gradients are complete: sum them over the ranks under the lower layers' backward pass
      if (overlap) {
        EESEN_HIP_CHECK(hipEventRecord(ev_grad[dg_slot], st2));
        side_pending[dg_slot] = true;
      }
      dg_slot ^= 1;
    }
    if (want_in) {
      std::swap(d, dn);
      ld_d = ld_n;
    }
  }
  if (in_diff) copy2d(st, d, ld_d, in_diff, ldi, rows, layers[0].din);
  flush_deferred_buckets();   // EESEN_COMM_DEFER=1: the exchange starts here, behind the last recurrence (comm.cpp)
  // the gradient buffer is complete only when the side stream has drained: make the caller's stream wait for it
  for (int k = 0; k < 2; ++k)
    if (side_pending[k]) EESEN_HIP_CHECK(hipStreamWaitEvent(st, ev_grad[k], 0));
}

// The two per-minibatch schedule decisions of the backward pass, both read from the plan the launcher itself will execute
// (lstm_bwd_plan: one selection function).
//   * side stream: the weight-gradient GEMMs under the next-lower recurrence, unless a narrow layer (320 < H <= 512) at more than 32
//     sequences takes a tile that is not "light" (see the measurements at the call site);
//   * exchange (comm.cpp): deferred behind the last recurrence whenever some layer's persistent backward grid leaves fewer than 256
//     registers per SIMD lane free on the CUs it occupies -- an RCCL all-reduce workgroup (ncclDevKernel_Generic_*: 512 threads,
//     248-256 registers per lane, profiles/r05_rccl_kernel_descriptors.md) then cannot become resident beside it, would run in the
//     gaps between recurrences and, while it waits for a late peer, hold CUs the next recurrence needs.  Every BASELINE shape is in
//     that class; narrow layers of <= 256 cells are not.  EESEN_COMM_DEFER=0|1 overrides.
bool Net::overlap_for_minibatch() const {
  bool ov = overlap;
  if (ov && tn.overlap < 0 && persistent)
    for (const Layer& L : layers)
      if (L.is_lstm() && L.H > 320 && L.H <= 512 && S > 32 && !lstm_bwd_plan(lstm_view(*this, L), true).light) ov = false;
  return ov;
}
bool Net::exchange_deferred_for_minibatch() const {
  if (tn.comm_defer >= 0) return tn.comm_defer != 0;
  if (!persistent) return false;    // no persistent grid: an all-reduce kernel only ever meets finite kernels
  for (const Layer& L : layers)
    if (L.is_lstm()) {
      const RecPlan bp = lstm_bwd_plan(lstm_view(*this, L), true);
      if (bp.kind != kRecNone && bp.free_vgprs < kRcclVgprsPerSimdLane) return true;   // (unknown register count: -1, deferred)
    }
  return false;
}

// eesen_net_plan_string: what runs the CURRENT minibatch shape (after eesen_net_set_seq_lengths; T from the last Propagate or the
// longest sequence), as JSON -- per LSTM layer the forward and backward recurrence plans (instantiation, tile, grid, windows,
// registers, LDS, what the grid leaves free on a CU), and the schedule decisions that follow from them.
std::string Net::plan_string() const {
  EESEN_REQUIRE(finalized, EESEN_ERR_STATE, "net not finalized");
  EESEN_REQUIRE(S > 0, EESEN_ERR_STATE, "call eesen_net_set_seq_lengths first: the plan depends on the minibatch shape");
  EESEN_HIP_CHECK(hipSetDevice(device));
  Net& self = const_cast<Net&>(*this);   // lstm_view reads T: plan for the shape the next Propagate will see
  const int T_saved = T;
  if (!propagated) { int tm = 0; for (int v : lens) tm = std::max(tm, v); self.T = tm; }
  auto one = [](const RecPlan& P) {
    char b[512];
    if (P.kind == kRecNone) { snprintf(b, sizeof(b), "{\"kernel\": \"per-step kernels (lstm.hip)\", \"persistent\": false}"); return std::string(b); }
    snprintf(b, sizeof(b), "{\"kernel\": \"%s\", \"persistent\": true, \"sequences_per_workgroup\": %d, \"units_per_workgroup\": %d, \"grid\": [%d, %d, %d], "
             "\"workgroups\": %d, \"workgroups_per_cu\": %d, \"launches\": %d, \"vgprs\": %d, \"lds_bytes\": %d, \"free_vgprs_per_simd_lane\": %d}",
             P.kernel, P.seq_tile, P.units, P.grid[0], P.grid[1], P.grid[2], P.wgs, P.wgs_per_cu, P.windows, P.vgprs, P.lds, P.free_vgprs);
    return std::string(b);
  };
  std::string o = "{\"T\": " + std::to_string(T) + ", \"S\": " + std::to_string(S) + ", \"persistent\": " + (persistent ? "true" : "false") + ", \"layers\": [";
  bool first = true;
  for (size_t li = 0; li < layers.size(); ++li) {
    const Layer& L = layers[li];
    if (!L.is_lstm()) continue;
    LstmLayerDev v = lstm_view(*this, L);
    const RecPlan fp = persistent && T >= 1 ? lstm_fwd_plan(v) : RecPlan{}, bp = persistent && T >= 1 ? lstm_bwd_plan(v, true) : RecPlan{};
    o += std::string(first ? "" : ", ") + "{\"layer\": " + std::to_string(li) + ", \"cells\": " + std::to_string(L.H) + ", \"directions\": " + std::to_string(L.ndir) +
         ", \"forward\": " + one(fp) + ", \"backward\": " + one(bp) + "}";
    first = false;
  }
  const bool ov = overlap_for_minibatch();
  o += std::string("], \"weight_gradient_gemms\": \"") + (ov ? "side stream, under the next-lower recurrence" : "main stream, serial") + "\"";
  o += std::string(", \"gemm_arithmetic\": \"") + (gemm_mode() == 2 ? "two fp16 planes, three products (fp32-class)" : gemm_mode() == 1 ? "3-way bf16 split (fp32-class)" : "f32-input MFMA") + "\"";
  o += std::string(", \"exchange\": ") + (comm ? (exchange_deferred_for_minibatch() ? "\"deferred: every bucket behind the backward pass's last recurrence\""
                                                                                      : "\"overlapped: each bucket as soon as its layer's gradients are enqueued\"") : "null");
  o += std::string(", \"exchange_rule\": \"") + (tn.comm_defer >= 0 ? "EESEN_COMM_DEFER" : "auto: deferred when a persistent backward grid leaves < 256 registers per SIMD lane") + "\"";
  o += ", \"gpu_share\": " + std::to_string(gpu_share_value()) + "}";
  self.T = T_saved;
  return o;
}

void Net::update() {
  EESEN_REQUIRE(finalized, EESEN_ERR_STATE, "net not finalized");
  EESEN_HIP_CHECK(hipSetDevice(device));
  // A persistent recurrence kernel that gave up waiting for a peer (error word raised) leaves garbage gradients: the update
  // kernels read the word ON THE DEVICE and do nothing then, so a failed step never reaches the parameters; the host notices
  // at its next poll and continues on the per-step kernels (check_device_error).
  // (After a data-parallel exchange the word is NOT looked at here: a failed rank has contributed zeros to the sum instead -- on the
  // device, before the all-reduce -- and must apply the same summed gradient as the others, or the ranks would diverge.)
  const unsigned* skip = persistent && !grads_sanitized ? ctl.p + kCtlWords - 1 : nullptr;
  // 0 after the all-reduce: no rank had a minibatch -- the closing round, a no-op.  Only looked at when THIS step wrote it
  // (Backpropagate / BackpropagateZero): an Update whose gradients came another way must not be gated on a stale word.
  const float* live = comm && live_valid ? fresh.p + P : nullptr;
  comm_check_alive(comm);
  { const int ti_ = timer.begin(st, 5);
  // top-down, the order in which Backpropagate completed (and all-reduced) the layers' gradients
  for (int li = (int)layers.size() - 1; li >= 0; --li) {
    Layer& L = layers[li];
    if (L.p_n) {
      if (comm && li < (int)bucket_pending.size() && bucket_pending[li]) {
        // phase 7 of the profiling spans: what the compute stream WAITS for this bucket -- the part of the exchange that the
        // backward pass of the lower layers did not hide (first event: the stream has reached the update; second: the bucket is in)
        const int tw_ = timer.begin(st, 7);
        EESEN_HIP_CHECK(hipStreamWaitEvent(st, ev_bucket[li], 0));
        timer.end(st, tw_);
        bucket_pending[li] = 0;
      }
      if (rule == 0) {
        sgd_update(st, params.p + L.p_off, corr.p + L.p_off, fresh.p + L.p_off, (long)L.p_n, mmt, lr * L.coef, L.max_grad, skip, live);
      } else {  // the adaptive rules do not apply learn_rate_coef (bilstm-layer.h:865-869 multiplies only in the SGD branch)
        init_accu();
        adaptive_update(st, params.p + L.p_off, corr.p + L.p_off, fresh.p + L.p_off, accu.p + L.p_off, (long)L.p_n, mmt, lr,
                        L.max_grad, ada_eps, rms_rho, rms_one_minus_rho, rule == 2, skip, live);
      }
    }
  }
  refresh_derived();
  timer.end(st, ti_); }
  grads_sanitized = false;
  live_valid = false;
  if (persistent) arm_device_error_poll();
}

}  // namespace eesen
