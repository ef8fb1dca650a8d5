// common.h -- error plumbing, device buffers and launch helpers shared by the HIP sources.
#pragma once
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <stdexcept>
#include <string>

#include "../../include/eesen_hip.h"

namespace eesen {

// Internal failures travel as exceptions up to the C-ABI wrappers (capi.cpp), which turn them into a
// status code + thread-local message; nothing throws across the `extern "C"` boundary.
struct Error : public std::runtime_error {
  int code;
  Error(int c, const std::string& m) : std::runtime_error(m), code(c) {}
};

#define EESEN_HIP_CHECK(expr)                                                                       \
  do {                                                                                              \
    hipError_t e_ = (expr);                                                                         \
    if (e_ != hipSuccess)                                                                           \
      throw ::eesen::Error(EESEN_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e_) + " (" + \
                                              __FILE__ + ":" + std::to_string(__LINE__) + ")");     \
  } while (0)

#define EESEN_REQUIRE(cond, code, msg)                                       \
  do {                                                                       \
    if (!(cond)) throw ::eesen::Error((code), std::string(msg) + " [" #cond "]"); \
  } while (0)

inline void check_launch(const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) throw Error(EESEN_ERR_HIP, std::string(what) + " launch: " + hipGetErrorString(e));
}

// Owning device allocation (fp32 / int32 elements), grow-only resize so steady-state steps allocate nothing.
template <typename T>
struct DevBuf {
  T* p = nullptr;
  size_t cap = 0;  // elements
  DevBuf() = default;
  DevBuf(const DevBuf&) = delete;
  DevBuf& operator=(const DevBuf&) = delete;
  DevBuf(DevBuf&& o) noexcept : p(o.p), cap(o.cap) { o.p = nullptr; o.cap = 0; }
  DevBuf& operator=(DevBuf&& o) noexcept {
    if (this != &o) { release(); p = o.p; cap = o.cap; o.p = nullptr; o.cap = 0; }
    return *this;
  }
  ~DevBuf() { release(); }
  void release() {
    if (p) (void)hipFree(p);
    p = nullptr;
    cap = 0;
  }
  // returns true when a new allocation was made (contents undefined then; `cap` elements, possibly more than n).
  // The FIRST allocation is exact (a fixed-shape workload pays for what it uses); a buffer that has to GROW takes half as much again:
  // the recipes sort their lists by length (train_ctc_parallel.sh:84-89), so T grows from minibatch to minibatch through a whole
  // epoch, and exact growth meant a hipFree -- which drains the device -- and a hipMalloc of every activation buffer on EVERY
  // minibatch (round 5, measured through the trainer binaries: 331 k padded frames/s end to end where the same loop on warm buffers
  // does 1.1 M).
  bool reserve(size_t n) {
    if (n <= cap) return false;
    // The margin is bounded (round 6): at most 1 GiB per buffer, and none when the device could not hold the margin twice
    // over -- a near-capacity shape that fits with exact sizing must not fail because EARLIER buffers kept slack.  Peak footprint on a
    // length-sorted list: the exact footprint of the longest minibatch + at most min(50 %, 1 GiB) per buffer.
    size_t want = cap ? std::max(n, cap + cap / 2) : n;
    if (want > n) {
      want = std::min(want, n + ((size_t)1 << 30) / sizeof(T));
      size_t fr = 0, tot = 0;
      if (hipMemGetInfo(&fr, &tot) != hipSuccess) { (void)hipGetLastError(); want = n; }
      else if (fr + cap * sizeof(T) < n * sizeof(T) + 2 * (want - n) * sizeof(T)) want = n;
    }
    release();
    if (hipMalloc(reinterpret_cast<void**>(&p), want * sizeof(T)) == hipSuccess) { cap = want; return true; }
    p = nullptr;
    (void)hipGetLastError();
    EESEN_HIP_CHECK(hipMalloc(reinterpret_cast<void**>(&p), n * sizeof(T)));   // (no room for the margin: exactly what is needed)
    cap = n;
    return true;
  }
};

inline int cdiv(int a, int b) { return (a + b - 1) / b; }
inline long cdivl(long a, long b) { return (a + b - 1) / b; }

}  // namespace eesen
