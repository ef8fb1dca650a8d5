// fake_rccl.hip -- TEST INFRASTRUCTURE: a stand-in for librccl.so that lets the library's own data-parallel path
// (eesen_amd/csrc/comm.cpp: dlopen(EESEN_RCCL_LIBRARY) -> ncclGetUniqueId / ncclCommInitRank / ncclAllReduce / ncclCommAbort /
// ncclCommDestroy / ncclGetErrorString) run with MORE THAN ONE RANK ON ONE GPU, which real RCCL refuses (it wants one device
// per rank) and which the 1-GPU boxes of this pool cannot offer otherwise.  Not part of the product; never loaded unless a
// test sets EESEN_RCCL_LIBRARY to this file.
//
// What it keeps of the real thing, because the tests are about exactly these properties:
//   * the all-reduce is a KERNEL enqueued on the stream it is given (FAKE_RCCL_BLOCKS x 512 threads, default 32 workgroups),
//     so it competes for CUs with whatever else runs on the device -- in particular with the persistent recurrence grids that
//     need every one of their workgroups co-resident -- and it is long-running: it moves its payload through host memory over
//     PCIe (25 MB bucket ~ 1 ms), the order of an xGMI ring all-reduce of the same bucket;
//   * ranks are separate processes; a rank's kernel SPINS until its peers' kernels have arrived (as RCCL's do), so a peer
//     that never issues the collective leaves the kernel spinning until ncclCommAbort raises the communicator's abort flag
//     (as RCCL's kernels poll theirs); a hard bound (FAKE_RCCL_SPIN_SECONDS, default 60) keeps a bug from hanging the box;
//   * every rank ends with bit-identical sums (fixed rank order).
//   * FAKE_RCCL_SHAPE=rccl (round 6): the kernel takes the FOOTPRINT of the real ncclDevKernel_Generic_* of RCCL 2.27.7's gfx950 code
//     object (profiles/r05_rccl_kernel_descriptors.md): 512-thread launch bound, 256 VGPRs per lane (the whole register file of
//     every SIMD of the CU it lands on), 37 664 bytes of static LDS -- so that what becomes resident beside which persistent
//     recurrence tile, and what waits for what when a peer is late, is what it will be under RCCL; the plain flavour (~30 VGPRs,
//     a few bytes of LDS) fits beside every tile and exercised a residency real RCCL will not have.
//
// Transport: one POSIX shared-memory segment named by the 128-byte unique id, mapped by every rank and registered with HIP
// (hipHostRegister: fine-grained, system-coherent).  Per collective chunk `seq` (<= FAKE_RCCL_CHUNK_MB, default 8):
//   block b: wait until every rank has released parity slot seq & 1 (done[r][b] >= seq - 2)
//            copy its range of the send buffer into slot[seq & 1][rank]; system-scope release; arrive[rank][b] = seq
//            wait arrive[r][b] >= seq for every r (system-scope acquire); sum slot[.][0..W) in rank order -> recv
//            done[rank][b] = seq
// Blocks own disjoint ranges, so each block is an independent pipeline and no grid-wide synchronisation is needed.
#include <fcntl.h>
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <atomic>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>

namespace {

constexpr unsigned kMagic = 0x46524343u;  // "FRCC"
constexpr int kThreads = 512;
constexpr int kMaxBlocks = 256, kMaxWorld = 16;
constexpr size_t kHeaderBytes = 4096;

struct Header {
  std::atomic<unsigned> magic, joined, left;
  unsigned world, nblocks;
  unsigned long long chunk_bytes;
};

struct Layout {
  size_t flags_off, slots_off, total;
  size_t slot_stride;  // bytes per (parity, rank)
};
Layout layout(int world, int nblocks, size_t chunk_bytes) {
  Layout l;
  l.flags_off = kHeaderBytes;
  const size_t flags = (size_t)2 * kMaxWorld * kMaxBlocks * sizeof(unsigned);  // arrive, done
  l.slots_off = (l.flags_off + flags + 4095) & ~(size_t)4095;
  l.slot_stride = chunk_bytes;
  l.total = l.slots_off + (size_t)2 * world * chunk_bytes;
  (void)nblocks;
  return l;
}

int env_int(const char* name, int dflt) {
  const char* e = getenv(name);
  return e && *e ? atoi(e) : dflt;
}

struct FakeComm {
  int rank = 0, world = 1, nblocks = 32;
  size_t chunk_bytes = 0;
  Layout lay;
  void* shm = nullptr;      // host mapping
  char* shm_dev = nullptr;  // the same bytes as the device sees them
  unsigned seq = 0;         // chunks issued so far (identical on every rank: same collectives, same sizes, same order)
  unsigned* abort_h = nullptr;  // pinned, device-visible: [0] abort flag, [1] error word raised by a kernel
  unsigned* abort_d = nullptr;
  unsigned spin_seconds = 60;
  bool rccl_shape = false;  // FAKE_RCCL_SHAPE=rccl
  int threads = kThreads;   // FAKE_RCCL_THREADS: 512 (the kernels' launch bound, default) or 256
  int device = 0;
};

__device__ __forceinline__ unsigned ld_acquire_sys(const unsigned* p) {
  return __hip_atomic_load(p, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM);
}
__device__ __forceinline__ void st_release_sys(unsigned* p, unsigned v) {
  __hip_atomic_store(p, v, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

// waits until flag[r * kMaxBlocks + b] >= target for every r < world (thread r polls rank r's flag); false = give up
__device__ bool wait_all(const unsigned* flag, int world, int b, unsigned target, const unsigned* abort_flag, unsigned* err,
                         unsigned long long deadline, int* s_bail) {
  if (threadIdx.x == 0) *s_bail = 0;
  __syncthreads();
  if ((int)threadIdx.x < world) {
    const unsigned* f = flag + (size_t)threadIdx.x * kMaxBlocks + b;
    unsigned spins = 0;
    while ((int)(ld_acquire_sys(f) - target) < 0) {
      if ((++spins & 63u) == 0) {
        if (__hip_atomic_load(abort_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != 0) { *s_bail = 1; break; }
        if (wall_clock64() > deadline) {
          __hip_atomic_store(err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
          *s_bail = 1;
          break;
        }
      }
      __builtin_amdgcn_s_sleep(8);
    }
  }
  __syncthreads();
  const bool ok = *s_bail == 0;
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");  // every wave: what the peers released is visible to the loads below
  return ok;
}

template <typename T, int OP>  // OP 0 sum, 1 max
__device__ __forceinline__ void fake_allreduce_body(const T* __restrict__ send, T* __restrict__ recv, size_t n,
                                                    char* slots, unsigned* arrive, unsigned* done,
                                                    const unsigned* abort_flag, unsigned* err, int rank, int world,
                                                    unsigned seq, size_t slot_stride, unsigned long long spin_ticks, int* s_bail_p) {
  int& s_bail = *s_bail_p;
  const int b = blockIdx.x, nb = gridDim.x;
  const unsigned long long deadline = wall_clock64() + spin_ticks;
  // this block's contiguous range of 16-byte vectors (the tail elements go to the last block)
  constexpr size_t V = 16 / sizeof(T);
  const size_t nvec = n / V, per = (nvec + nb - 1) / nb;
  const size_t v0 = (size_t)b * per < nvec ? (size_t)b * per : nvec, v1 = v0 + per < nvec ? v0 + per : nvec;
  const size_t e0 = v0 * V, e1 = b == nb - 1 ? n : v1 * V;
  char* par = slots + (size_t)(seq & 1u) * world * slot_stride;
  T* mine = reinterpret_cast<T*>(par + (size_t)rank * slot_stride);

  if (seq >= 3 && !wait_all(done, world, b, seq - 2, abort_flag, err, deadline, &s_bail)) return;
  for (size_t i = e0 + threadIdx.x; i < e1; i += blockDim.x) mine[i] = send[i];
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");  // every wave: its stores are visible at system scope ...
  __syncthreads();
  if (threadIdx.x == 0) st_release_sys(arrive + (size_t)rank * kMaxBlocks + b, seq);  // ... before the flag is

  if (!wait_all(arrive, world, b, seq, abort_flag, err, deadline, &s_bail)) return;
  for (size_t i = e0 + threadIdx.x; i < e1; i += blockDim.x) {
    T acc = reinterpret_cast<const T*>(par)[i];
    for (int r = 1; r < world; ++r) {
      const T v = reinterpret_cast<const T*>(par + (size_t)r * slot_stride)[i];
      acc = OP == 0 ? acc + v : (v > acc ? v : acc);
    }
    recv[i] = acc;
  }
  __syncthreads();
  if (threadIdx.x == 0) st_release_sys(done + (size_t)rank * kMaxBlocks + b, seq);
}

template <typename T, int OP>
__global__ __launch_bounds__(kThreads) void fake_allreduce_kernel(const T* __restrict__ send, T* __restrict__ recv, size_t n,
                                                                  char* slots, unsigned* arrive, unsigned* done,
                                                                  const unsigned* abort_flag, unsigned* err, int rank, int world,
                                                                  unsigned seq, size_t slot_stride, unsigned long long spin_ticks) {
  __shared__ int s_bail;
  fake_allreduce_body<T, OP>(send, recv, n, slots, arrive, done, abort_flag, err, rank, world, seq, slot_stride, spin_ticks, &s_bail);
}

// The same collective with RCCL's residency footprint (FAKE_RCCL_SHAPE=rccl): ncclDevKernel_Generic_4 of librccl 2.27.7 for gfx950
// reports .max_flat_workgroup_size 512, .vgpr_count 256, .group_segment_fixed_size 37664.  The registers are claimed by naming the
// last one of the budget in an asm clobber (the allocation is the highest register a kernel touches), the LDS by a static array
// that is really written and read; tests/test_kernel_resources.py reads both back from the built code object.
constexpr int kRcclLdsBytes = 37664;
template <typename T, int OP>
__global__ __launch_bounds__(kThreads) void fake_allreduce_rccl_shaped_kernel(const T* __restrict__ send, T* __restrict__ recv, size_t n,
                                                                              char* slots, unsigned* arrive, unsigned* done,
                                                                              const unsigned* abort_flag, unsigned* err, int rank, int world,
                                                                              unsigned seq, size_t slot_stride, unsigned long long spin_ticks) {
  __shared__ int lds[kRcclLdsBytes / sizeof(int)];
  asm volatile("v_mov_b32 v255, 0" ::: "v255");
  for (int i = threadIdx.x + 1; i < (int)(kRcclLdsBytes / sizeof(int)); i += blockDim.x) lds[i] = (int)seq;
  __syncthreads();
  fake_allreduce_body<T, OP>(send, recv, n, slots, arrive, done, abort_flag, err, rank, world, seq, slot_stride, spin_ticks, &lds[0]);
  if (lds[1 + (seq % 1024u)] != (int)seq) __hip_atomic_store(err, 2u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);   // (never: keeps the array alive)
}

const char* kErr[] = {"no error", "unhandled HIP error", "unhandled system error", "internal error", "invalid argument",
                      "invalid usage", "remote error", "in progress"};

double now_s() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

}  // namespace

extern "C" {

ncclResult_t ncclGetUniqueId(ncclUniqueId* id) {
  if (!id) return ncclInvalidArgument;
  std::memset(id->internal, 0, sizeof(id->internal));
  static std::atomic<unsigned> counter{0};
  std::snprintf(id->internal, sizeof(id->internal), "/eesen_fake_rccl_%d_%llx_%u", (int)getpid(),
                (unsigned long long)std::chrono::steady_clock::now().time_since_epoch().count(), counter.fetch_add(1));
  return ncclSuccess;
}

ncclResult_t ncclCommInitRank(ncclComm_t* comm, int nranks, ncclUniqueId id, int rank) {
  if (!comm || nranks < 1 || nranks > kMaxWorld || rank < 0 || rank >= nranks || id.internal[0] != '/') return ncclInvalidArgument;
  FakeComm* c = new FakeComm;
  c->rank = rank; c->world = nranks;
  c->nblocks = std::min(kMaxBlocks, std::max(1, env_int("FAKE_RCCL_BLOCKS", 32)));
  c->chunk_bytes = (size_t)std::max(1, env_int("FAKE_RCCL_CHUNK_MB", 8)) << 20;
  c->spin_seconds = (unsigned)std::max(1, env_int("FAKE_RCCL_SPIN_SECONDS", 60));
  { const char* sh = getenv("FAKE_RCCL_SHAPE"); c->rccl_shape = sh && std::strcmp(sh, "rccl") == 0; }
  (void)hipGetDevice(&c->device);
  c->threads = env_int("FAKE_RCCL_THREADS", kThreads) <= 256 ? 256 : kThreads;
  c->lay = layout(nranks, c->nblocks, c->chunk_bytes);
  id.internal[sizeof(id.internal) - 1] = 0;
  const int fd = shm_open(id.internal, O_CREAT | O_RDWR, 0600);
  if (fd < 0) { perror("fake_rccl: shm_open"); delete c; return ncclSystemError; }
  if (ftruncate(fd, (off_t)c->lay.total) != 0) { perror("fake_rccl: ftruncate"); close(fd); delete c; return ncclSystemError; }
  c->shm = mmap(nullptr, c->lay.total, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
  close(fd);
  if (c->shm == MAP_FAILED) { perror("fake_rccl: mmap"); delete c; return ncclSystemError; }
  Header* h = static_cast<Header*>(c->shm);
  // a fresh segment is zero-filled: flags start at 0, the first chunk is seq 1
  unsigned expect = 0;
  if (h->magic.compare_exchange_strong(expect, kMagic)) { h->world = (unsigned)nranks; h->nblocks = (unsigned)c->nblocks; h->chunk_bytes = c->chunk_bytes; }
  h->joined.fetch_add(1);
  const double deadline = now_s() + env_int("FAKE_RCCL_INIT_SECONDS", 120);
  while (h->joined.load() < (unsigned)nranks) {  // the bootstrap barrier of the real thing
    if (now_s() > deadline) { fprintf(stderr, "fake_rccl: rank %d: only %u of %d ranks joined\n", rank, h->joined.load(), nranks); shm_unlink(id.internal); munmap(c->shm, c->lay.total); delete c; return ncclRemoteError; }
    std::this_thread::sleep_for(std::chrono::milliseconds(2));
  }
  if (h->world != (unsigned)nranks || h->nblocks != (unsigned)c->nblocks || h->chunk_bytes != c->chunk_bytes) {
    fprintf(stderr, "fake_rccl: ranks disagree on world size / geometry\n");
    return ncclInvalidUsage;
  }
  if (h->left.fetch_add(1) + 1 == (unsigned)nranks) shm_unlink(id.internal);  // every rank has it mapped: the name can go
  if (hipHostRegister(c->shm, c->lay.total, hipHostRegisterMapped | hipHostRegisterPortable) != hipSuccess) {
    fprintf(stderr, "fake_rccl: hipHostRegister failed: %s\n", hipGetErrorString(hipGetLastError()));
    return ncclUnhandledCudaError;
  }
  void* d = nullptr;
  if (hipHostGetDevicePointer(&d, c->shm, 0) != hipSuccess) return ncclUnhandledCudaError;
  c->shm_dev = static_cast<char*>(d);
  if (hipHostMalloc(reinterpret_cast<void**>(&c->abort_h), 64, hipHostMallocMapped) != hipSuccess) return ncclUnhandledCudaError;
  c->abort_h[0] = c->abort_h[1] = 0;
  if (hipHostGetDevicePointer(&d, c->abort_h, 0) != hipSuccess) return ncclUnhandledCudaError;
  c->abort_d = static_cast<unsigned*>(d);
  *comm = reinterpret_cast<ncclComm_t>(c);
  if (rank == 0 && !getenv("FAKE_RCCL_QUIET"))
    fprintf(stderr, "fake_rccl (TEST STAND-IN, not RCCL): %d rank(s), %d x %d-thread workgroups per all-reduce%s, %zu MB chunks through host memory\n",
            nranks, c->nblocks, c->threads, c->rccl_shape ? " with RCCL's footprint (256 VGPRs, 37.7 KB LDS)" : "", c->chunk_bytes >> 20);
  return ncclSuccess;
}

ncclResult_t ncclAllReduce(const void* sendbuff, void* recvbuff, size_t count, ncclDataType_t datatype, ncclRedOp_t op,
                           ncclComm_t comm, hipStream_t stream) {
  FakeComm* c = reinterpret_cast<FakeComm*>(comm);
  if (!c || !sendbuff || !recvbuff) return ncclInvalidArgument;
  if (c->abort_h[0]) return ncclInvalidUsage;  // aborted communicator
  if (!((datatype == ncclFloat32 && op == ncclSum) || (datatype == ncclFloat64 && (op == ncclSum || op == ncclMax)))) return ncclInvalidArgument;
  const size_t esz = datatype == ncclFloat32 ? 4 : 8;
  const size_t per_chunk = c->chunk_bytes / esz;
  unsigned* arrive = reinterpret_cast<unsigned*>(c->shm_dev + c->lay.flags_off);
  unsigned* done = arrive + (size_t)kMaxWorld * kMaxBlocks;
  char* slots = c->shm_dev + c->lay.slots_off;
  const unsigned long long ticks = (unsigned long long)c->spin_seconds * 100000000ull;  // wall_clock64: 100 MHz
  for (size_t off = 0; off < count; off += per_chunk) {
    const size_t n = std::min(per_chunk, count - off);
    const unsigned seq = ++c->seq;
    const dim3 grid(c->nblocks), block(c->threads);
#define FAKE_LAUNCH(KERN, TYPE, OPV)                                                                                              \
  hipLaunchKernelGGL((KERN<TYPE, OPV>), grid, block, 0, stream, static_cast<const TYPE*>(sendbuff) + off, static_cast<TYPE*>(recvbuff) + off, \
                     n, slots, arrive, done, c->abort_d, c->abort_d + 1, c->rank, c->world, seq, c->lay.slot_stride, ticks)
    if (c->rccl_shape) {
      if (datatype == ncclFloat32) FAKE_LAUNCH(fake_allreduce_rccl_shaped_kernel, float, 0);
      else if (op == ncclSum) FAKE_LAUNCH(fake_allreduce_rccl_shaped_kernel, double, 0);
      else FAKE_LAUNCH(fake_allreduce_rccl_shaped_kernel, double, 1);
    } else {
      if (datatype == ncclFloat32) FAKE_LAUNCH(fake_allreduce_kernel, float, 0);
      else if (op == ncclSum) FAKE_LAUNCH(fake_allreduce_kernel, double, 0);
      else FAKE_LAUNCH(fake_allreduce_kernel, double, 1);
    }
#undef FAKE_LAUNCH
    if (hipGetLastError() != hipSuccess) return ncclUnhandledCudaError;
  }
  return ncclSuccess;
}

// As the real one: kernels in flight see the flag and leave; the communicator is unusable afterwards.  Resources stay mapped
// (kernels may still be draining) -- a test process ends soon after.
ncclResult_t ncclCommAbort(ncclComm_t comm) {
  FakeComm* c = reinterpret_cast<FakeComm*>(comm);
  if (!c) return ncclInvalidArgument;
  __atomic_store_n(&c->abort_h[0], 1u, __ATOMIC_SEQ_CST);
  return ncclSuccess;
}

ncclResult_t ncclCommDestroy(ncclComm_t comm) {
  FakeComm* c = reinterpret_cast<FakeComm*>(comm);
  if (!c) return ncclSuccess;
  (void)hipDeviceSynchronize();
  if (c->abort_h && c->abort_h[1]) fprintf(stderr, "fake_rccl: rank %d: an all-reduce kernel gave up waiting for a peer\n", c->rank);
  if (c->shm) { (void)hipHostUnregister(c->shm); munmap(c->shm, c->lay.total); }
  if (c->abort_h) (void)hipHostFree(c->abort_h);
  delete c;
  return ncclSuccess;
}

// what eesen_comm_describe asks the library about itself: version 0 = "not RCCL"
ncclResult_t ncclGetVersion(int* v) { if (!v) return ncclInvalidArgument; *v = 0; return ncclSuccess; }
ncclResult_t ncclCommCount(const ncclComm_t comm, int* n) {
  const FakeComm* c = reinterpret_cast<const FakeComm*>(comm);
  if (!c || !n) return ncclInvalidArgument;
  *n = c->world; return ncclSuccess;
}
ncclResult_t ncclCommUserRank(const ncclComm_t comm, int* r) {
  const FakeComm* c = reinterpret_cast<const FakeComm*>(comm);
  if (!c || !r) return ncclInvalidArgument;
  *r = c->rank; return ncclSuccess;
}
ncclResult_t ncclCommCuDevice(const ncclComm_t comm, int* d) {
  const FakeComm* c = reinterpret_cast<const FakeComm*>(comm);
  if (!c || !d) return ncclInvalidArgument;
  *d = c->device; return ncclSuccess;
}

const char* ncclGetErrorString(ncclResult_t r) { return (int)r >= 0 && (int)r < 8 ? kErr[(int)r] : "unknown result code"; }

// test hook: 1 if a kernel of this communicator hit its spin bound
int fake_rccl_error_word(ncclComm_t comm) {
  FakeComm* c = reinterpret_cast<FakeComm*>(comm);
  return c && c->abort_h ? (int)c->abort_h[1] : -1;
}

}  // extern "C"
