// ctc_host.cpp -- host side of the Ctc object: label expansion, launch sequence, statistics, greedy
// decoding.  Replaces eesen::Ctc::EvalParallel / ErrorRateMSeq (/root/reference/src/net/ctc-loss.cc:101-194,
// :235-298).  The lattice arithmetic itself is in ctc.hip.
#include <algorithm>
#include <cmath>
#include <fstream>

#include <algorithm>
#include <limits>

#include "net.h"

namespace eesen {

Ctc::Ctc(int dev, void* stream) : device(dev) {
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess || n <= 0)
    throw Error(EESEN_ERR_HIP, "no HIP device available: this library has no CPU fallback");
  EESEN_REQUIRE(dev >= 0 && dev < n, EESEN_ERR_INVALID, "device index out of range");
  EESEN_HIP_CHECK(hipSetDevice(dev));
  st = reinterpret_cast<hipStream_t>(stream);  // NULL = the device's default stream (shared with the Net)
  if (const char* e = getenv("EESEN_CTC_WAVES"); e && *e) sweep_waves = atoi(e);
  for (auto& x : ev) EESEN_HIP_CHECK(hipEventCreate(&x));
}

Ctc::~Ctc() {
  (void)hipSetDevice(device);
  (void)hipStreamSynchronize(st);
  if (guard_net) {
    auto& g = guard_net->guards;
    g.erase(std::remove(g.begin(), g.end(), this), g.end());
  }
  for (auto& x : ev)
    if (x) (void)hipEventDestroy(x);
  auto drop = [](Pin& pin) {
    if (pin.p) (void)hipHostFree(pin.p);
    if (pin.ev) (void)hipEventDestroy(pin.ev);
  };
  for (auto& x : stage) drop(x);
  for (auto& x : ppzx) drop(x.pin);
  for (auto& x : perr) { drop(x.pin); drop(x.probs); }
  if (own_stream) (void)hipStreamDestroy(st);
}

void* Ctc::pin_reserve(Pin& pin, size_t bytes) {
  if (!pin.ev) EESEN_HIP_CHECK(hipEventCreateWithFlags(&pin.ev, hipEventDisableTiming));
  if (pin.busy) { EESEN_HIP_CHECK(hipEventSynchronize(pin.ev)); pin.busy = false; }  // two calls ago: long done
  if (bytes > pin.cap) {
    if (pin.p) EESEN_HIP_CHECK(hipHostFree(pin.p));
    pin.p = nullptr;
    pin.cap = std::max<size_t>(bytes * 2, 4096);
    EESEN_HIP_CHECK(hipHostMalloc(&pin.p, pin.cap, hipHostMallocDefault));
  }
  return pin.p;
}

void Ctc::flush_pzx(PendingPzx& q) {
  if (!q.active) return;
  EESEN_HIP_CHECK(hipEventSynchronize(q.pin.ev));
  q.pin.busy = false;
  const float* pz = static_cast<const float*>(q.pin.p);
  q.active = false;
  if (reinterpret_cast<const unsigned*>(pz)[q.S] != 0) {  // computed from a timed-out forward pass (see Ctc::guard): not a statistic
    sequences -= q.S;
    frames -= q.nframes;
    if (dropped++ == 0 || dropped % 100 == 0)
      fprintf(stderr, "WARNING (eesen_hip) CTC statistics of a minibatch computed from a timed-out forward pass were dropped (%ld so far)\n", dropped);
    return;
  }
  double sum = 0;
  for (int s = 0; s < q.S; ++s) sum += pz[s];
  obj_sum += sum;  // ctc-loss.cc:171-177
}

void Ctc::flush() {
  EESEN_HIP_CHECK(hipSetDevice(device));
  for (unsigned k = 0; k < 2; ++k) flush_pzx(ppzx[(ppzx_idx + k) & 1]);   // oldest first
  for (unsigned k = 0; k < 2; ++k) flush_err(perr[(perr_idx + k) & 1], nullptr, nullptr);
}

static void check_batch(const int* frame_num_utt, int S, int rows, int K, const int* label_ids, const int* label_off) {
  EESEN_REQUIRE(S > 0 && rows > 0 && rows % S == 0, EESEN_ERR_INVALID, "rows must be a positive multiple of the sequence count");
  EESEN_REQUIRE(K > 0, EESEN_ERR_INVALID, "no classes");
  const int T = rows / S;
  for (int s = 0; s < S; ++s) {
    EESEN_REQUIRE(frame_num_utt[s] >= 0 && frame_num_utt[s] <= T, EESEN_ERR_INVALID, "frame_num_utt out of range");
    EESEN_REQUIRE(label_off[s + 1] >= label_off[s], EESEN_ERR_INVALID, "label offsets must be non-decreasing");
  }
  for (int i = label_off[0]; i < label_off[S]; ++i)
    EESEN_REQUIRE(label_ids[i] >= 0 && label_ids[i] < K, EESEN_ERR_INVALID, "label id outside [0, K)");
}

void Ctc::eval_parallel(const int* frame_num_utt, int S, const float* net_out, int rows, int K, int ld, const int* label_ids,
                        const int* label_off, float* diff, int ldd, float* pzx_host) {
  check_batch(frame_num_utt, S, rows, K, label_ids, label_off);
  EESEN_REQUIRE(ld >= K && ldd >= K, EESEN_ERR_INVALID, "leading dimension smaller than the class count");
  EESEN_HIP_CHECK(hipSetDevice(device));
  const int T = rows / S;
  int maxU = 0;
  for (int s = 0; s < S; ++s) {
    const int U = label_off[s + 1] - label_off[s];
    // an empty label sequence reads alpha column -1 in the reference (ctc-loss.cc:151); refuse it
    EESEN_REQUIRE(U >= 1, EESEN_ERR_INVALID, "every sequence needs at least one label");
    maxU = std::max(maxU, U);
  }
  const int Lprime = 2 * maxU + 1;  // ctc-loss.cc:118
  int PL = 1;
  while (64 * PL < Lprime) PL *= 2;
  EESEN_REQUIRE(PL <= 64, EESEN_ERR_INVALID, "expanded label length above 4096 (2047 labels per utterance) is not supported");
  const int Lpad = 64 * PL;

  // label expansion (ctc-loss.cc:116-129), sequence and expanded-label lengths: one staging vector
  const size_t n_labx = (size_t)S * Lpad;
  std::vector<int> h(n_labx + 2 * (size_t)S);
  int* labx_h = h.data();
  int* lens_h = labx_h + n_labx;
  int* ll_h = lens_h + S;
  std::fill(labx_h, labx_h + n_labx, -1);
  for (int s = 0; s < S; ++s) {
    const int U = label_off[s + 1] - label_off[s];
    const int* lab = label_ids + label_off[s];
    int* lx = labx_h + (size_t)s * Lpad;
    for (int l = 0; l < U; ++l) { lx[2 * l] = 0; lx[2 * l + 1] = lab[l]; }
    lx[2 * U] = 0;
    lens_h[s] = frame_num_utt[s];
    ll_h[s] = 2 * U + 1;
  }
  // Stream-ordered upload through a pinned slot: kernels of the previous call that still read labx are ahead of this copy on
  // the stream, and the slot written here was last read by the copy of two calls ago -- nothing drains the stream.
  if (labx.cap < h.size()) EESEN_HIP_CHECK(hipStreamSynchronize(st));  // reallocation frees what queued kernels may still read
  labx.reserve(h.size());
  Pin& sp = stage[stage_idx++ & 1];
  int* pinned = static_cast<int*>(pin_reserve(sp, h.size() * sizeof(int)));
  std::copy(h.begin(), h.end(), pinned);
  EESEN_HIP_CHECK(hipMemcpyAsync(labx.p, pinned, h.size() * sizeof(int), hipMemcpyHostToDevice, st));
  EESEN_HIP_CHECK(hipEventRecord(sp.ev, st));
  sp.busy = true;
  const int* labx_d = labx.p;
  const int* lens_dd = labx_d + n_labx;
  const int* ll_d = lens_dd + S;

  if (logp.cap < (size_t)rows * K || alpha.cap < (size_t)S * T * Lpad || pzx_d.cap < (size_t)S) EESEN_HIP_CHECK(hipStreamSynchronize(st));
  logp.reserve((size_t)rows * K);
  alpha.reserve((size_t)S * T * Lpad);
  beta.reserve((size_t)S * T * Lpad);
  pzx_d.reserve(S);

  const bool acc = timer.enabled();
  int sp0 = -1, sp1 = -1, sp2 = -1;
  if (acc) sp0 = timer.begin(st, 0); else EESEN_HIP_CHECK(hipEventRecord(ev[0], st));
  log_rows(st, net_out, ld, logp.p, K, rows, K);                                               // ctc-loss.cc:132-133
  if (acc) { timer.end(st, sp0); sp1 = timer.begin(st, 1); } else EESEN_HIP_CHECK(hipEventRecord(ev[1], st));
  ctc_alpha_beta(st, logp.p, K, T, S, Lpad, labx_d, lens_dd, ll_d, alpha.p, beta.p, pzx_d.p, sweep_waves);  // :136-153
  if (acc) { timer.end(st, sp1); sp2 = timer.begin(st, 2); } else EESEN_HIP_CHECK(hipEventRecord(ev[2], st));
  ctc_error_diff(st, net_out, ld, T, S, K, Lpad, Lprime, lens_dd, ll_d, labx_d, alpha.p, beta.p, pzx_d.p, diff, ldd);  // :156-168
  if (acc) timer.end(st, sp2); else EESEN_HIP_CHECK(hipEventRecord(ev[3], st));

  // ln p(z|x) per sequence (ctc-loss.cc:146-153 reads it element by element): back through a pinned slot.  A caller that wants
  // the values now waits for them; otherwise they join the objective sum when the next call needs the slot or the statistics.
  PendingPzx& q = ppzx[ppzx_idx++ & 1];
  flush_pzx(q);
  float* pz = static_cast<float*>(pin_reserve(q.pin, (size_t)(S + 1) * sizeof(float)));
  EESEN_HIP_CHECK(hipMemcpyAsync(pz, pzx_d.p, S * sizeof(float), hipMemcpyDeviceToHost, st));
  reinterpret_cast<unsigned*>(pz)[S] = 0;   // the guard word's value when these ln p were computed
  if (guard) EESEN_HIP_CHECK(hipMemcpyAsync(pz + S, guard, sizeof(unsigned), hipMemcpyDeviceToHost, st));
  EESEN_HIP_CHECK(hipEventRecord(q.pin.ev, st));
  q.pin.busy = true; q.S = S; q.active = true;
  q.nframes = 0;
  for (int s = 0; s < S; ++s) q.nframes += frame_num_utt[s];
  frames += q.nframes;
  sequences += S;
  if (pzx_host) {
    flush();  // keeps the accumulation order of the calls
    // a minibatch computed from a timed-out forward pass (guard word set) has no ln p: NaN, never garbage with status OK
    const bool bad = reinterpret_cast<const unsigned*>(pz)[S] != 0;
    for (int s = 0; s < S; ++s) pzx_host[s] = bad ? std::numeric_limits<float>::quiet_NaN() : pz[s];
  }
  last_lens.assign(frame_num_utt, frame_num_utt + S);
  last_T = T; last_S = S; last_Lpad = Lpad; last_Lprime = Lprime;
}

void Ctc::phase_times(float* out3) {
  if (timer.enabled()) { timer.collect(out3, 3); return; }   // sums since the last read
  EESEN_HIP_CHECK(hipEventSynchronize(ev[3]));
  for (int i = 0; i < 3; ++i) {
    float ms = 0.f;
    EESEN_HIP_CHECK(hipEventElapsedTime(&ms, ev[i], ev[i + 1]));
    out3[i] = ms * 1e-3f;
  }
}

void Ctc::get_alpha_beta(float* alpha_host, float* beta_host, int* Lprime) {
  EESEN_REQUIRE(last_T > 0, EESEN_ERR_STATE, "no EvalParallel yet");
  EESEN_HIP_CHECK(hipSetDevice(device));
  EESEN_HIP_CHECK(hipStreamSynchronize(st));
  const int T = last_T, S = last_S, Lpad = last_Lpad, Lp = last_Lprime;
  if (Lprime) *Lprime = Lp;
  std::vector<float> h((size_t)S * T * Lpad);
  for (int which = 0; which < 2; ++which) {
    float* dst = which == 0 ? alpha_host : beta_host;
    if (!dst) continue;
    EESEN_HIP_CHECK(hipMemcpy(h.data(), which == 0 ? alpha.p : beta.p, h.size() * sizeof(float), hipMemcpyDeviceToHost));
    for (int t = 0; t < T; ++t)
      for (int s = 0; s < S; ++s) {
        float* o = dst + ((size_t)t * S + s) * Lp;
        if (t >= last_lens[s]) {  // rows the sweep never visits; the reference leaves -1e30 there (ctc-loss.cc:138-139)
          for (int j = 0; j < Lp; ++j) o[j] = -1e30f;
        } else {
          const float* src = h.data() + ((size_t)s * T + t) * Lpad;
          for (int j = 0; j < Lp; ++j) o[j] = src[j];
        }
      }
  }
}

// LevenshteinEditDistance (src/util/edit-distance-inl.h), total errors only
static int levenshtein(const int* ref, int nr, const std::vector<int>& hyp) {
  const int nh = (int)hyp.size();
  std::vector<int> prev(nh + 1), cur(nh + 1);
  for (int j = 0; j <= nh; ++j) prev[j] = j;
  for (int i = 1; i <= nr; ++i) {
    cur[0] = i;
    for (int j = 1; j <= nh; ++j) {
      const int sub = prev[j - 1] + (ref[i - 1] == hyp[j - 1] ? 0 : 1);
      cur[j] = std::min(sub, std::min(prev[j] + 1, cur[j - 1] + 1));
    }
    std::swap(prev, cur);
  }
  return prev[nh];
}

void Ctc::flush_err(PendingErr& q, int* num_err, int* num_ref) {
  if (!q.active) return;
  EESEN_HIP_CHECK(hipEventSynchronize(q.pin.ev));
  q.pin.busy = false;
  const int* ids = static_cast<const int*>(q.pin.p);
  const int S = q.S;
  int err = 0, ref = 0;
  if (q.guarded && ids[(size_t)q.rows] != 0) {  // decoded from a timed-out forward pass (see Ctc::guard): not a statistic
    if (q.with_probs) { EESEN_HIP_CHECK(hipEventSynchronize(q.probs.ev)); q.probs.busy = false; }
    if (num_err) *num_err = 0;
    if (num_ref) *num_ref = 0;
    q.active = false;
    return;
  }
  std::vector<int> hyp, frm;
  std::ofstream output;
  if (q.with_probs) {
    EESEN_HIP_CHECK(hipEventSynchronize(q.probs.ev));
    q.probs.busy = false;
    output.open(seq_out, std::ofstream::out | std::ofstream::app);  // ctc-loss.cc:247-250
  }
  const float* probs = static_cast<const float*>(q.probs.p);
  for (int s = 0; s < S; ++s) {
    hyp.clear(); frm.clear();
    int last = -1;
    for (int f = 0; f < q.frames[s]; ++f) {  // collapse repeats, drop blanks (ctc-loss.cc:252-275)
      const int id = ids[(size_t)f * S + s];
      if (f == 0 || id != last) {
        if (id != 0) { hyp.push_back(id); frm.push_back(f); }
      }
      last = id;
    }
    const int U = q.off[s + 1] - q.off[s];
    err += levenshtein(q.ids.data() + q.off[s], U, hyp);
    ref += U;
    if (q.with_probs) {  // :282-291: the index of the phone, the frame, and the probability
      output << "utt";
      for (size_t i = 0; i < hyp.size(); ++i)
        output << " | " << hyp[i] << " " << frm[i] << " " << probs[((size_t)frm[i] * S + s) * q.K + hyp[i]];
      output << "\n";
    }
  }
  err_tokens += err;
  ref_tokens += ref;
  if (num_err) *num_err = err;
  if (num_ref) *num_ref = ref;
  q.active = false;
}

// With num_err == num_ref == NULL the call only ENQUEUES the argmax and the copy of the ids; collapsing and the edit
// distances (host work, ctc-loss.cc:252-296) run when the next call needs the slot or the statistics are read -- by then
// the device is busy with the backward pass, so neither side waits for the other.
void Ctc::error_rate_mseq(const int* frame_num_utt, int S, const float* net_out, int rows, int K, int ld,
                          const int* label_ids, const int* label_off, int* num_err, int* num_ref) {
  check_batch(frame_num_utt, S, rows, K, label_ids, label_off);
  EESEN_HIP_CHECK(hipSetDevice(device));
  if (ids_d.cap < (size_t)rows) EESEN_HIP_CHECK(hipStreamSynchronize(st));
  ids_d.reserve(rows);
  PendingErr& q = perr[perr_idx++ & 1];
  flush_err(q, nullptr, nullptr);
  row_argmax(st, net_out, ld, rows, K, ids_d.p);  // FindRowMaxId, ctc-loss.cc:238-239
  int* pinned = static_cast<int*>(pin_reserve(q.pin, ((size_t)rows + 1) * sizeof(int)));
  EESEN_HIP_CHECK(hipMemcpyAsync(pinned, ids_d.p, (size_t)rows * sizeof(int), hipMemcpyDeviceToHost, st));
  pinned[rows] = 0;
  q.rows = rows; q.guarded = guard != nullptr;
  if (guard) EESEN_HIP_CHECK(hipMemcpyAsync(pinned + rows, guard, sizeof(unsigned), hipMemcpyDeviceToHost, st));
  EESEN_HIP_CHECK(hipEventRecord(q.pin.ev, st));
  q.pin.busy = true; q.S = S; q.K = K; q.active = true;
  q.with_probs = !seq_out.empty();
  if (q.with_probs) {  // "This is inefficient, but ok for now" (ctc-loss.cc:283): the whole posterior matrix comes back
    float* pp = static_cast<float*>(pin_reserve(q.probs, (size_t)rows * K * sizeof(float)));
    EESEN_HIP_CHECK(hipMemcpy2DAsync(pp, (size_t)K * sizeof(float), net_out, (size_t)ld * sizeof(float), (size_t)K * sizeof(float), rows,
                                     hipMemcpyDeviceToHost, st));
    EESEN_HIP_CHECK(hipEventRecord(q.probs.ev, st));
    q.probs.busy = true;
  }
  q.frames.assign(frame_num_utt, frame_num_utt + S);
  q.off.assign(label_off, label_off + S + 1);
  for (int& o : q.off) o -= label_off[0];
  q.ids.assign(label_ids + label_off[0], label_ids + label_off[S]);
  if (num_err || num_ref) {
    flush_err(perr[perr_idx & 1], nullptr, nullptr);  // the older one first: totals accumulate in call order
    flush_err(q, num_err, num_ref);
  }
}

}  // namespace eesen
