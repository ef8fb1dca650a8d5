// comm.cpp -- the data-parallel exchange of the hot path: one process per GPU, RCCL over xGMI.
//
// Replaces the reference's multi-job mode, which is asynchronous MODEL AVERAGING THROUGH FILES
// (/root/reference/src/net/communicator.h:39-119: every --utts-per-avg utterances each job writes nnet.avgK.jobJ, job 1
// polls with usleep, sums, rescales and writes nnet.avgK, the others re-read it; :121-170 merge the error counts through
// "done" files).  Here the FRESH gradients (sums over frames, bilstm-parallel-layer.h:504-510) are summed over the ranks
// every minibatch, per layer, as soon as that layer's weight-gradient GEMMs have finished -- the point where the reference
// calls Update on the layer (/root/reference/src/net/net.cc:98-104) -- on a separate stream, under the lower layers'
// backward pass; momentum, clipping and the update then run identically on every rank (N ranks x S
// utterances == one process with --num-sequence = N*S).  The scalar statistics travel through the same communicator.
//
// RCCL is loaded with dlopen on first use: a single-GPU process never maps it, and the library keeps loading on boxes
// without it.  Rendezvous of the 128-byte ncclUniqueId is a plain TCP hand-out by rank 0 (no MPI, no torch).
//
// Two things the reference's file protocol had and a collective does not get for free:
//   * jobs with DIFFERENT numbers of minibatches (communicator.h:104-112: a sub-job "gives up if the main job finishes first").
//     Here a rank that is out of data keeps stepping with a zero gradient (Net::backpropagate_zero) and one float rides with
//     the top layer's gradient bucket: 1 from every rank that had a minibatch, 0 from the others.  Its sum tells a draining
//     rank whether anybody is still training (Net::live_ranks), and the update kernels read it ON THE DEVICE: a round in
//     which no rank was live -- the closing round all ranks take together -- does not touch the model.  No per-step host
//     round trip is needed for any of this.
//   * a peer that DIES.  The reference's job 1 would poll for a file forever; a collective kernel would spin forever.  Every
//     collective is followed by an event that a watchdog thread of the communicator watches: one that has not completed
//     EESEN_COMM_TIMEOUT_S (default 600) seconds after it was issued makes the watchdog call ncclCommAbort -- the kernels see
//     the abort flag and leave, every host wait returns -- and every later call on the communicator or a net attached to it
//     fails with EESEN_ERR_COMM.
#include <arpa/inet.h>
#include <dlfcn.h>
#include <netdb.h>
#include <netinet/in.h>
#include <netinet/tcp.h>
#include <sys/socket.h>
#include <unistd.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstring>
#include <deque>
#include <mutex>
#include <thread>

#include <rccl/rccl.h>

#include "guard.h"
#include "handles.h"
#include "net.h"

namespace eesen {

namespace {

struct RcclApi {
  void* dl = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*CommAbort)(ncclComm_t) = nullptr;  // optional
  ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
  const char* (*GetErrorString)(ncclResult_t) = nullptr;
  // optional: only for eesen_comm_describe (what the bench line says about the library that really ran)
  ncclResult_t (*GetVersion)(int*) = nullptr;
  ncclResult_t (*CommCount)(const ncclComm_t, int*) = nullptr;
  ncclResult_t (*CommCuDevice)(const ncclComm_t, int*) = nullptr;
  ncclResult_t (*CommUserRank)(const ncclComm_t, int*) = nullptr;
  std::string path;        // where the dynamic linker found the library (dladdr of ncclAllReduce)
  bool stand_in = false;   // it exports fake_rccl_error_word: tests/native/libfake_rccl.so, not RCCL
};

RcclApi& rccl() {
  static RcclApi api;
  if (api.dl) return api;
  const char* names[] = {getenv("EESEN_RCCL_LIBRARY"), "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
  for (const char* n : names) {
    if (!n || !*n) continue;
    api.dl = dlopen(n, RTLD_NOW | RTLD_LOCAL);
    if (api.dl) break;
  }
  if (!api.dl) throw Error(EESEN_ERR_HIP, std::string("RCCL is not loadable (librccl.so.1): ") + (dlerror() ? dlerror() : "not found"));
  auto sym = [&](const char* s) {
    void* p = dlsym(api.dl, s);
    if (!p) throw Error(EESEN_ERR_HIP, std::string("RCCL symbol missing: ") + s);
    return p;
  };
  api.GetUniqueId = reinterpret_cast<decltype(api.GetUniqueId)>(sym("ncclGetUniqueId"));
  api.CommInitRank = reinterpret_cast<decltype(api.CommInitRank)>(sym("ncclCommInitRank"));
  api.CommDestroy = reinterpret_cast<decltype(api.CommDestroy)>(sym("ncclCommDestroy"));
  api.CommAbort = reinterpret_cast<decltype(api.CommAbort)>(dlsym(api.dl, "ncclCommAbort"));
  api.AllReduce = reinterpret_cast<decltype(api.AllReduce)>(sym("ncclAllReduce"));
  api.GetErrorString = reinterpret_cast<decltype(api.GetErrorString)>(sym("ncclGetErrorString"));
  api.GetVersion = reinterpret_cast<decltype(api.GetVersion)>(dlsym(api.dl, "ncclGetVersion"));
  api.CommCount = reinterpret_cast<decltype(api.CommCount)>(dlsym(api.dl, "ncclCommCount"));
  api.CommCuDevice = reinterpret_cast<decltype(api.CommCuDevice)>(dlsym(api.dl, "ncclCommCuDevice"));
  api.CommUserRank = reinterpret_cast<decltype(api.CommUserRank)>(dlsym(api.dl, "ncclCommUserRank"));
  api.stand_in = dlsym(api.dl, "fake_rccl_error_word") != nullptr;
  Dl_info di{};
  if (dladdr(reinterpret_cast<void*>(api.AllReduce), &di) && di.dli_fname) api.path = di.dli_fname;
  return api;
}

#define EESEN_NCCL_CHECK(expr)                                                                                   \
  do {                                                                                                            \
    ncclResult_t r_ = (expr);                                                                                     \
    if (r_ != ncclSuccess)                                                                                        \
      throw ::eesen::Error(EESEN_ERR_HIP, std::string(#expr) + ": " + rccl().GetErrorString(r_) + " (" + __FILE__ + \
                                              ":" + std::to_string(__LINE__) + ")");                              \
  } while (0)

// ---- TCP hand-out of the unique id --------------------------------------------------------------------------------
void send_all(int fd, const char* p, size_t n) {
  while (n) {
    const ssize_t k = ::send(fd, p, n, MSG_NOSIGNAL);
    if (k <= 0) throw Error(EESEN_ERR_IO, std::string("rendezvous send failed: ") + strerror(errno));
    p += k; n -= (size_t)k;
  }
}
void recv_all(int fd, char* p, size_t n) {
  while (n) {
    const ssize_t k = ::recv(fd, p, n, 0);
    if (k <= 0) throw Error(EESEN_ERR_IO, std::string("rendezvous receive failed: ") + (k == 0 ? "peer closed" : strerror(errno)));
    p += k; n -= (size_t)k;
  }
}
struct Fd {
  int fd = -1;
  ~Fd() { if (fd >= 0) ::close(fd); }
};
double now_s() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

}  // namespace

// Rank 0 listens on addr:port and hands `buf` (n bytes) to the world-1 ranks that connect and announce their rank;
// every other rank connects (retrying until rank 0 is up) and receives into `buf`.  Each peer is served exactly once.
void comm_exchange(const char* addr, int port, int rank, int world, char* buf, int n, int timeout_s) {
  EESEN_REQUIRE(world >= 1 && rank >= 0 && rank < world, EESEN_ERR_INVALID, "rendezvous: bad rank / world size");
  EESEN_REQUIRE(port > 0 && port < 65536, EESEN_ERR_INVALID, "rendezvous: bad port");
  if (world == 1) return;
  const double deadline = now_s() + timeout_s;
  const unsigned magic = 0x45534e31u;  // "ESN1"
  if (rank == 0) {
    Fd ls;
    ls.fd = ::socket(AF_INET, SOCK_STREAM, 0);
    EESEN_REQUIRE(ls.fd >= 0, EESEN_ERR_IO, "rendezvous: socket() failed");
    int one = 1;
    (void)setsockopt(ls.fd, SOL_SOCKET, SO_REUSEADDR, &one, sizeof(one));
    sockaddr_in sa{};
    sa.sin_family = AF_INET;
    sa.sin_port = htons((uint16_t)port);
    // Listen on ONE interface only when the caller named it as a numeric address (--comm-addr=10.0.0.5 / MASTER_ADDR=127.0.0.1): that
    // is an explicit choice.  A host NAME is what the peers were told to reach, not a statement about local interfaces -- on hosts
    // whose /etc/hosts maps their own name to loopback (Debian / Ubuntu: 127.0.1.1) binding to its local resolution would succeed on
    // loopback and remote peers, who resolve the real address, could never connect: names listen on every interface.
    sa.sin_addr.s_addr = htonl(INADDR_ANY);
    bool bound = false;
    if (addr && *addr) {
      sockaddr_in sb = sa;
      if (inet_pton(AF_INET, addr, &sb.sin_addr) == 1 && sb.sin_addr.s_addr != htonl(INADDR_ANY))
        bound = ::bind(ls.fd, reinterpret_cast<sockaddr*>(&sb), sizeof(sb)) == 0;
    }
    if (!bound && ::bind(ls.fd, reinterpret_cast<sockaddr*>(&sa), sizeof(sa)) != 0)
      throw Error(EESEN_ERR_IO, "rendezvous: cannot bind port " + std::to_string(port) + ": " + strerror(errno));
    EESEN_REQUIRE(::listen(ls.fd, world) == 0, EESEN_ERR_IO, "rendezvous: listen() failed");
    std::vector<char> served(world, 0);
    int left = world - 1;
    while (left > 0) {
      const double remain = deadline - now_s();
      if (remain <= 0) throw Error(EESEN_ERR_IO, "rendezvous: timed out waiting for " + std::to_string(left) + " rank(s)");
      timeval tv{(time_t)remain, (suseconds_t)((remain - (time_t)remain) * 1e6)};
      fd_set rf;
      FD_ZERO(&rf);
      FD_SET(ls.fd, &rf);
      if (::select(ls.fd + 1, &rf, nullptr, nullptr, &tv) <= 0) continue;
      Fd c;
      c.fd = ::accept(ls.fd, nullptr, nullptr);
      if (c.fd < 0) continue;
      timeval rt{10, 0};
      (void)setsockopt(c.fd, SOL_SOCKET, SO_RCVTIMEO, &rt, sizeof(rt));
      unsigned hello[2] = {0, 0};
      try {
        recv_all(c.fd, reinterpret_cast<char*>(hello), sizeof(hello));
      } catch (const Error&) {
        continue;  // a stray connection: ignore it
      }
      if (hello[0] != magic || hello[1] == 0 || hello[1] >= (unsigned)world || served[hello[1]]) continue;
      try {
        send_all(c.fd, buf, (size_t)n);
        unsigned ack = 0;   // the peer confirms it HAS the blob: a connection that died on the way is not counted as served
        recv_all(c.fd, reinterpret_cast<char*>(&ack), sizeof(ack));
        if (ack != magic) continue;
      } catch (const Error&) {
        continue;  // a half-open or stray peer: it may connect again
      }
      served[hello[1]] = 1;
      --left;
    }
  } else {
    addrinfo hints{}, *res = nullptr;
    hints.ai_family = AF_INET;
    hints.ai_socktype = SOCK_STREAM;
    const std::string ps = std::to_string(port);
    if (getaddrinfo(addr && *addr ? addr : "127.0.0.1", ps.c_str(), &hints, &res) != 0 || !res)
      throw Error(EESEN_ERR_IO, std::string("rendezvous: cannot resolve ") + (addr ? addr : "(null)"));
    std::string last = "no attempt";
    for (;;) {
      Fd c;
      c.fd = ::socket(AF_INET, SOCK_STREAM, 0);
      if (c.fd >= 0 && ::connect(c.fd, res->ai_addr, res->ai_addrlen) == 0) {
        timeval rt{(time_t)std::max(1.0, deadline - now_s()), 0};
        (void)setsockopt(c.fd, SOL_SOCKET, SO_RCVTIMEO, &rt, sizeof(rt));
        const unsigned hello[2] = {magic, (unsigned)rank};
        try {
          send_all(c.fd, reinterpret_cast<const char*>(hello), sizeof(hello));
          recv_all(c.fd, buf, (size_t)n);
          send_all(c.fd, reinterpret_cast<const char*>(&magic), sizeof(magic));
          freeaddrinfo(res);
          return;
        } catch (const Error& e) {
          last = e.what();
        }
      } else {
        last = strerror(errno);
      }
      if (now_s() > deadline) {
        freeaddrinfo(res);
        throw Error(EESEN_ERR_IO, "rendezvous: rank " + std::to_string(rank) + " could not reach rank 0 at " + (addr ? addr : "") + ":" + ps + " (" + last + ")");
      }
      std::this_thread::sleep_for(std::chrono::milliseconds(50));
    }
  }
}

// ---- Comm ---------------------------------------------------------------------------------------------------------
struct Comm {
  int device = 0, rank = 0, world = 1;
  ncclComm_t comm = nullptr;
  hipStream_t st = nullptr;  // the exchange runs here, beside the compute stream(s)
  double* scratch_d = nullptr;
  double* scratch_h = nullptr;  // pinned
  static constexpr int kScratch = 64;

  // watchdog (see the head of this file)
  struct Pending { hipEvent_t ev; double t0; };
  std::thread wd;
  std::mutex mu;
  std::condition_variable cv;
  std::deque<Pending> pending;
  std::vector<hipEvent_t> free_ev;
  bool stop = false;
  std::atomic<bool> dead{false};
  std::string dead_msg;
  double timeout_s = 600.0;

  Comm(int dev, const char* id128, int rank_, int world_) : device(dev), rank(rank_), world(world_) {
    EESEN_REQUIRE(world >= 1 && rank >= 0 && rank < world, EESEN_ERR_INVALID, "communicator: bad rank / world size");
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) throw Error(EESEN_ERR_HIP, "no HIP device available for the communicator");
    EESEN_REQUIRE(dev >= 0 && dev < n, EESEN_ERR_INVALID, "device index out of range");
    EESEN_HIP_CHECK(hipSetDevice(dev));
    ncclUniqueId id;
    std::memcpy(id.internal, id128, NCCL_UNIQUE_ID_BYTES);
    EESEN_NCCL_CHECK(rccl().CommInitRank(&comm, world, id, rank));
    int lo = 0, hi = 0;
    EESEN_HIP_CHECK(hipDeviceGetStreamPriorityRange(&lo, &hi));
    EESEN_HIP_CHECK(hipStreamCreateWithPriority(&st, hipStreamNonBlocking, hi));
    EESEN_HIP_CHECK(hipMalloc(reinterpret_cast<void**>(&scratch_d), kScratch * sizeof(double)));
    EESEN_HIP_CHECK(hipHostMalloc(reinterpret_cast<void**>(&scratch_h), kScratch * sizeof(double), hipHostMallocDefault));
    if (const char* e = getenv("EESEN_COMM_TIMEOUT_S")) timeout_s = std::max(0.05, atof(e));
    wd = std::thread([this] { watch(); });
    share_colocated_device();
  }
  // Jobs that were given the SAME device (the reference's own two-jobs-one-GPU test mode; `--num-jobs 2` with one GPU in the box)
  // must size their persistent grids against their share of it, all of them alike -- or each plans for every CU and they find out
  // through spin time-outs.  The ranks' device ids are gathered once, here (one 8-byte-per-rank host all-reduce: every rank is in
  // this constructor), and a process that finds k > 1 ranks on its device sizes against 1/k of it from now on (set_gpu_share),
  // unless EESEN_GPU_SHARE says otherwise.  Nets created BEFORE the communicator keep the tiles they planned with.
  void share_colocated_device() {
    if (world <= 1 || world > kScratch) return;
    try {
      const double mine = device_word();
      double v[kScratch] = {0};
      v[rank] = mine;
      allreduce_host(v, world, 0);
      int here = 0;
      for (int r = 0; r < world; ++r) here += v[r] == mine;
      if (here > 1) {
        if (getenv("EESEN_GPU_SHARE") == nullptr) set_gpu_share(here);
        if (rank == 0 || here != world)
          fprintf(stderr, "LOG (eesen_hip) %d of the %d data-parallel ranks share this rank's device: persistent grids are sized against 1/%d of its CUs%s\n",
                  here, world, gpu_share_value(), getenv("EESEN_GPU_SHARE") ? " (EESEN_GPU_SHARE)" : "");
      }
    } catch (const Error& e) {   // never fatal: the explicit EESEN_GPU_SHARE remains
      fprintf(stderr, "WARNING (eesen_hip) could not find out whether data-parallel ranks share a device (%s)\n", e.what());
    }
  }
  // host hash (20 bits) | PCI domain (16) | bus (8) | device (5) | function (3) of this rank's GPU, + 1: exact in a double, never 0
  double device_word() const {
    char bus[64] = "";
    if (hipDeviceGetPCIBusId(bus, sizeof(bus), device) != hipSuccess) { (void)hipGetLastError(); bus[0] = 0; }
    unsigned dom = 0, b = 0, d = 0, f = 0;
    (void)sscanf(bus, "%x:%x:%x.%x", &dom, &b, &d, &f);
    char host[256] = "";
    (void)gethostname(host, sizeof(host) - 1);
    unsigned hh = 2166136261u;   // FNV-1a of the host name: ranks of different hosts never look like one device
    for (const char* c = host; *c; ++c) hh = (hh ^ (unsigned char)*c) * 16777619u;
    return (double)(((unsigned long long)(hh & 0xfffffu) << 32) | ((unsigned long long)(dom & 0xffffu) << 16) | ((b & 0xffu) << 8) | ((d & 0x1fu) << 3) | (f & 7u)) + 1.0;
  }
  ~Comm() {
    {
      std::lock_guard<std::mutex> g(mu);
      stop = true;
    }
    cv.notify_all();
    if (wd.joinable()) wd.join();
    (void)hipSetDevice(device);
    if (st) (void)hipStreamSynchronize(st);
    if (comm) (void)rccl().CommDestroy(comm);
    for (auto& p : pending) (void)hipEventDestroy(p.ev);
    for (auto e : free_ev) (void)hipEventDestroy(e);
    if (scratch_d) (void)hipFree(scratch_d);
    if (scratch_h) (void)hipHostFree(scratch_h);
    if (st) (void)hipStreamDestroy(st);
  }
  void check_alive() const {
    if (dead.load(std::memory_order_acquire)) throw Error(EESEN_ERR_COMM, dead_msg);
  }
  // an event behind the collective just enqueued on `on`: the watchdog sees when it completes
  void track(hipStream_t on) {
    hipEvent_t ev = nullptr;
    {
      std::lock_guard<std::mutex> g(mu);
      if (!free_ev.empty()) { ev = free_ev.back(); free_ev.pop_back(); }
    }
    if (!ev) EESEN_HIP_CHECK(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
    EESEN_HIP_CHECK(hipEventRecord(ev, on));
    {
      std::lock_guard<std::mutex> g(mu);
      pending.push_back({ev, now_s()});
    }
  }
  void watch() {
    (void)hipSetDevice(device);
    std::unique_lock<std::mutex> lk(mu);
    while (!stop) {
      cv.wait_for(lk, std::chrono::milliseconds(timeout_s < 5 ? 10 : 100));
      while (!pending.empty()) {
        const Pending p = pending.front();
        lk.unlock();
        const hipError_t e = hipEventQuery(p.ev);
        lk.lock();
        if (e == hipSuccess) {
          pending.pop_front();
          free_ev.push_back(p.ev);
          continue;
        }
        if (now_s() - p.t0 > timeout_s && !dead.load()) {
          dead_msg = "data-parallel exchange: a collective issued " + std::to_string((int)(now_s() - p.t0)) + " s ago has not completed on rank " +
                     std::to_string(rank) + " of " + std::to_string(world) + " (a peer died or stalled; EESEN_COMM_TIMEOUT_S = " +
                     std::to_string((int)timeout_s) + "): communicator aborted";
          fprintf(stderr, "ERROR (eesen_hip) %s\n", dead_msg.c_str());
          if (!rccl().CommAbort) {  // nothing can release the spinning kernels: better a dead process than a hung job
            fprintf(stderr, "ERROR (eesen_hip) this RCCL has no ncclCommAbort: exiting\n");
            std::_Exit(70);
          }
          dead.store(true, std::memory_order_release);
          ncclComm_t c = comm;
          comm = nullptr;  // ncclCommAbort frees it
          lk.unlock();
          (void)rccl().CommAbort(c);
          lk.lock();
        }
        break;
      }
    }
  }
  // in place, sum, fp32, enqueued on `on`
  void allreduce_f32(float* buf, size_t n, hipStream_t on) {
    if (n == 0) return;
    check_alive();
    EESEN_NCCL_CHECK(rccl().AllReduce(buf, buf, n, ncclFloat32, ncclSum, comm, on));
    track(on);
  }
  // host scalars: sum (op 0) or max (op 1) over the ranks; blocks until done
  void allreduce_host(double* v, int n, int op) {
    EESEN_REQUIRE(n >= 0 && n <= kScratch, EESEN_ERR_INVALID, "at most 64 scalars per call");
    if (n == 0) return;
    check_alive();
    EESEN_HIP_CHECK(hipSetDevice(device));
    std::memcpy(scratch_h, v, n * sizeof(double));
    EESEN_HIP_CHECK(hipMemcpyAsync(scratch_d, scratch_h, n * sizeof(double), hipMemcpyHostToDevice, st));
    EESEN_NCCL_CHECK(rccl().AllReduce(scratch_d, scratch_d, (size_t)n, ncclFloat64, op == 1 ? ncclMax : ncclSum, comm, st));
    track(st);
    EESEN_HIP_CHECK(hipMemcpyAsync(scratch_h, scratch_d, n * sizeof(double), hipMemcpyDeviceToHost, st));
    EESEN_HIP_CHECK(hipStreamSynchronize(st));  // returns when the collective has run -- or when the watchdog has aborted it
    check_alive();
    std::memcpy(v, scratch_h, n * sizeof(double));
  }
  // What this communicator really is, as JSON -- COLLECTIVE (every rank calls it; it gathers one word per rank): the library the
  // dynamic linker resolved and its version, whether it is the tests' stand-in, the ranks the library itself counts, and the
  // PCI bus id of every rank's device, so that a reader of a bench line can tell "N ranks on N distinct GPUs through RCCL" from
  // "N ranks sharing one GPU through a stand-in" without trusting the launcher.
  std::string describe() {
    check_alive();
    EESEN_HIP_CHECK(hipSetDevice(device));
    int version = 0, seen = -1, nccl_dev = -1, nccl_rank = -1;
    if (rccl().GetVersion) (void)rccl().GetVersion(&version);
    if (rccl().CommCount) (void)rccl().CommCount(comm, &seen);
    if (rccl().CommCuDevice) (void)rccl().CommCuDevice(comm, &nccl_dev);
    if (rccl().CommUserRank) (void)rccl().CommUserRank(comm, &nccl_rank);
    const double mine = device_word();
    std::vector<double> ids;
    for (int base = 0; base < world; base += kScratch) {   // gather = sum of vectors that are zero except at the own rank
      const int n = std::min(kScratch, world - base);
      double v[kScratch] = {0};
      if (rank >= base && rank < base + n) v[rank - base] = mine;
      allreduce_host(v, n, 0);
      ids.insert(ids.end(), v, v + n);
    }
    std::vector<double> uniq(ids);
    std::sort(uniq.begin(), uniq.end());
    const int distinct = (int)(std::unique(uniq.begin(), uniq.end()) - uniq.begin());
    std::string o = "{\"library\": \"" + rccl().path + "\", \"rccl_version\": " + std::to_string(version) + ", \"stand_in\": " + (rccl().stand_in ? "true" : "false") +
                    ", \"rank\": " + std::to_string(rank) + ", \"world\": " + std::to_string(world) + ", \"world_seen\": " + std::to_string(seen) +
                    ", \"rank_seen\": " + std::to_string(nccl_rank) + ", \"device\": " + std::to_string(device) + ", \"device_seen\": " + std::to_string(nccl_dev) + ", \"devices\": [";
    for (int r = 0; r < world; ++r) {
      const unsigned long long u = (unsigned long long)(ids[r] - 1.0);
      char t[64];
      snprintf(t, sizeof(t), "%s\"%05x/%04x:%02x:%02x.%x\"", r ? ", " : "", (unsigned)((u >> 32) & 0xfffffu), (unsigned)((u >> 16) & 0xffffu), (unsigned)((u >> 8) & 0xffu),
               (unsigned)((u >> 3) & 0x1fu), (unsigned)(u & 7u));
      o += t;
    }
    o += "], \"distinct_devices\": " + std::to_string(distinct) + ", \"ranks_share_devices\": " + (distinct < world ? "true" : "false") + "}";
    return o;
  }
  // a rank that cannot keep the collective sequence (Net::fail_step_buckets could not issue what the peers will wait for)
  void abort_now(const std::string& why) {
    std::unique_lock<std::mutex> lk(mu);
    if (dead.load()) return;
    dead_msg = "data-parallel exchange aborted on rank " + std::to_string(rank) + ": " + why;
    fprintf(stderr, "ERROR (eesen_hip) %s\n", dead_msg.c_str());
    dead.store(true, std::memory_order_release);
    ncclComm_t c = comm;
    comm = nullptr;
    lk.unlock();
    if (rccl().CommAbort && c) (void)rccl().CommAbort(c);
  }
};

void comm_check_alive(const Comm* c) {
  if (c) c->check_alive();
}

// ---- Net side: per-layer buckets -------------------------------------------------------------------------------------
void Net::set_comm(Comm* c) {
  EESEN_REQUIRE(finalized, EESEN_ERR_STATE, "net not finalized");
  EESEN_REQUIRE(!c || c->device == device, EESEN_ERR_INVALID, "communicator and net live on different devices");
  wait_buckets_host();
  comm = c;
  // RCCL's kernels share the CUs with the cooperative recurrence kernels: a workgroup of the latter may have to wait for an
  // all-reduce block to retire before it becomes resident.  That always ends (the collectives never depend on later compute),
  // so give the bounded spins ten times the room instead of treating it as a lost peer.
  if (c && !tn.spin_limit_set) spin_limit = std::max(spin_limit, 4000000);
  if (c && ev_ready.size() < layers.size()) {
    EESEN_HIP_CHECK(hipSetDevice(device));
    const size_t old = ev_ready.size();
    ev_ready.resize(layers.size(), nullptr);
    ev_bucket.resize(layers.size(), nullptr);
    for (size_t i = old; i < layers.size(); ++i) {
      EESEN_HIP_CHECK(hipEventCreateWithFlags(&ev_ready[i], hipEventDisableTiming));
      EESEN_HIP_CHECK(hipEventCreateWithFlags(&ev_bucket[i], hipEventDisableTiming));
    }
  }
  bucket_pending.assign(layers.size(), 0);
}

// The liveness word sits right behind the LAST trainable layer's block of the gradient buffer (which is the first bucket a
// backward pass issues), so it travels with that bucket at no cost.
int Net::top_trainable() const {
  for (int li = (int)layers.size() - 1; li >= 0; --li)
    if (layers[li].p_n) return li;
  return -1;
}

// layer li's fresh gradients are complete once everything enqueued on `producer` so far has run: sum them over the ranks
// on the communicator's stream; update() makes the compute stream wait for exactly this bucket
void Net::bucket_allreduce(int li, hipStream_t producer) {
  if (!comm || !layers[li].p_n) return;
  bucket_log.push_back(li);
  EESEN_HIP_CHECK(hipEventRecord(ev_ready[li], producer));
  if (exchange_deferred) { deferred_buckets.push_back(li); return; }   // issued by flush_deferred_buckets, in this order
  issue_bucket(li);
}

// Deferred mode (EESEN_COMM_DEFER=1): every bucket of this backward pass, in the order the layers completed, once the main stream's
// last recurrence and input-gradient GEMM have run -- the collectives then meet only the side stream's finite GEMMs on the chip, never
// a persistent grid that needs every CU.  Same buckets, same order, same sums as the overlapped mode; update() waits bucket by bucket.
void Net::flush_deferred_buckets() {
  if (!comm || deferred_buckets.empty()) return;
  if (!ev_bwd_done) EESEN_HIP_CHECK(hipEventCreateWithFlags(&ev_bwd_done, hipEventDisableTiming));
  EESEN_HIP_CHECK(hipEventRecord(ev_bwd_done, st));
  EESEN_HIP_CHECK(hipStreamWaitEvent(comm->st, ev_bwd_done, 0));
  for (int li : deferred_buckets) issue_bucket(li);
  deferred_buckets.clear();
}

void Net::issue_bucket(int li) {
  EESEN_HIP_CHECK(hipStreamWaitEvent(comm->st, ev_ready[li], 0));
  const bool top = li == top_trainable();  // its block ends at P: the liveness word (4 floats of padding) rides along
  // phase 6 of the profiling spans (eesen_net_get_phase_spans): this bucket's collective on the communicator's stream, from the
  // moment the bucket is ready AND the stream is free to the end of the all-reduce
  // A rank whose recurrence kernels raised the error word in this step (a bounded spin gave up, or the early-GEMM waiter did) holds
  // garbage gradients.  It must neither put them into the sum nor skip the update the other ranks make (they would diverge): it
  // contributes ZERO -- decided on the device, where the word is -- and then applies the same summed gradient as everybody else.
  // Its liveness word goes with the top bucket, so a step in which EVERY rank failed is a no-op on every rank alike.
  if (ctl.p) zero_if_set(comm->st, fresh.p + layers[li].p_off, (long)(layers[li].p_n + (top ? kLiveWords : 0)), ctl.p + kCtlWords - 1);
  grads_sanitized = true;
  const int ti_ = timer.begin(comm->st, 6);
  comm->allreduce_f32(fresh.p + layers[li].p_off, layers[li].p_n + (top ? kLiveWords : 0), comm->st);
  timer.end(comm->st, ti_);
  EESEN_HIP_CHECK(hipEventRecord(ev_bucket[li], comm->st));
  bucket_pending[li] = 1;
}

// Backpropagate threw half-way (a refused option, a HIP error in a lower layer) on a rank whose peers are in the same step: they
// will issue EVERY bucket of it, in the order of the layers.  The overlapped schedule has issued the upper layers' buckets by
// then, the deferred one none: in both cases the peers would spin until the watchdog's EESEN_COMM_TIMEOUT_S.  So the
// failing rank completes the sequence before the exception leaves the library: buckets whose gradients are complete go as they
// are, the others as ZEROS (liveness 0 with the top bucket) -- the ranks' models stay identical, the step loses this rank's
// share -- and if even that cannot be enqueued the communicator is aborted so that this process, at least, stops at once.
void Net::fail_step_buckets() noexcept {
  if (!comm) return;
  try {
    EESEN_HIP_CHECK(hipSetDevice(device));
    std::vector<char> logged(layers.size(), 0);
    for (int li : bucket_log) logged[li] = 1;
    flush_deferred_buckets();                    // complete gradients, recorded but not yet issued
    if (!ev_bwd_done) EESEN_HIP_CHECK(hipEventCreateWithFlags(&ev_bwd_done, hipEventDisableTiming));
    for (hipStream_t s : {st, st2}) {            // whatever was enqueued before the failure may still write the gradient buffer
      if (!s) continue;
      EESEN_HIP_CHECK(hipEventRecord(ev_bwd_done, s));
      EESEN_HIP_CHECK(hipStreamWaitEvent(comm->st, ev_bwd_done, 0));
    }
    const int top = top_trainable();
    for (int li = (int)layers.size() - 1; li >= 0; --li) {
      if (!layers[li].p_n || logged[li]) continue;
      const size_t n = layers[li].p_n + (li == top ? kLiveWords : 0);
      EESEN_HIP_CHECK(hipMemsetAsync(fresh.p + layers[li].p_off, 0, n * sizeof(float), comm->st));
      bucket_log.push_back(li);
      EESEN_HIP_CHECK(hipEventRecord(ev_ready[li], comm->st));
      issue_bucket(li);
    }
    fprintf(stderr, "WARNING (eesen_hip) Backpropagate failed on this rank in a data-parallel step: its unfinished gradient buckets were "
                    "all-reduced as zeros so that the other ranks are not left waiting\n");
  } catch (...) {
    try { comm->abort_now("Backpropagate failed and the step's remaining gradient buckets could not be issued"); } catch (...) {}
  }
}

void Net::wait_buckets_host() {
  if (!comm) return;
  bool any = false;
  for (char p : bucket_pending) any |= p != 0;
  if (any) EESEN_HIP_CHECK(hipStreamSynchronize(comm->st));
  comm->check_alive();
}

// a rank that has no minibatch this step: zero gradient, liveness 0, same collectives in the same (top-down) order
void Net::backpropagate_zero() {
  EESEN_REQUIRE(finalized, EESEN_ERR_STATE, "net not finalized");
  comm_check_alive(comm);
  EESEN_HIP_CHECK(hipSetDevice(device));
  if (P) EESEN_HIP_CHECK(hipMemsetAsync(fresh.p, 0, (P + kLiveWords) * sizeof(float), st));
  bucket_log.clear();
  deferred_buckets.clear();
  exchange_deferred = false;   // no recurrence runs in this step: nothing to defer behind
  live_valid = comm != nullptr;
  for (int li = (int)layers.size() - 1; li >= 0; --li) bucket_allreduce(li, st);
  flush_deferred_buckets();
}

// How many ranks had a minibatch in the step whose top bucket was issued last (the sum of the liveness words).  For a
// rank in the zero-gradient protocol: 0 = every rank is out of data, stop.  Blocks until that bucket has arrived.
int Net::live_ranks() {
  EESEN_REQUIRE(finalized && comm, EESEN_ERR_STATE, "live_ranks needs an attached communicator");
  comm->check_alive();
  EESEN_HIP_CHECK(hipSetDevice(device));
  const int top = top_trainable();
  EESEN_REQUIRE(top >= 0, EESEN_ERR_STATE, "the net has no trainable layer");
  if (bucket_pending[top]) EESEN_HIP_CHECK(hipStreamWaitEvent(st, ev_bucket[top], 0));  // update() has not waited for it yet
  if (!live_pin) EESEN_HIP_CHECK(hipHostMalloc(reinterpret_cast<void**>(&live_pin), sizeof(float), hipHostMallocDefault));
  EESEN_HIP_CHECK(hipMemcpyAsync(live_pin, fresh.p + P, sizeof(float), hipMemcpyDeviceToHost, st));
  EESEN_HIP_CHECK(hipStreamSynchronize(st));
  comm->check_alive();
  return (int)(*live_pin + 0.5f);
}

void Net::allreduce_grads(Comm* c) {
  EESEN_REQUIRE(finalized, EESEN_ERR_STATE, "net not finalized");
  EESEN_REQUIRE(c && c->device == device, EESEN_ERR_INVALID, "communicator missing or on another device");
  EESEN_HIP_CHECK(hipSetDevice(device));
  if (ctl.p) zero_if_set(st, fresh.p, (long)P, ctl.p + kCtlWords - 1);   // a failed step contributes zero (see bucket_allreduce)
  grads_sanitized = true;
  c->allreduce_f32(fresh.p, P, st);  // ordered on the compute stream: after Backpropagate's kernels, before Update's
}

}  // namespace eesen

using namespace eesen;
struct eesen_comm : public Comm { using Comm::Comm; };

extern "C" {

int eesen_comm_get_unique_id(char* id128) {
  return guard([&] {
    REQ_PTR(id128);
    ncclUniqueId id;
    EESEN_NCCL_CHECK(rccl().GetUniqueId(&id));
    std::memcpy(id128, id.internal, NCCL_UNIQUE_ID_BYTES);
  });
}
int eesen_comm_exchange(const char* addr, int port, int rank, int world, char* buf, int nbytes, int timeout_s) {
  return guard([&] { REQ_PTR(buf); comm_exchange(addr, port, rank, world, buf, nbytes, timeout_s); });
}
int eesen_comm_create(int device, const char* id128, int rank, int world, eesen_comm_t** out) {
  return guard([&] { REQ_PTR(id128); REQ_PTR(out); *out = new eesen_comm(device, id128, rank, world); });
}
int eesen_comm_create_tcp(int device, const char* addr, int port, int rank, int world, int timeout_s, eesen_comm_t** out) {
  return guard([&] {
    REQ_PTR(out);
    char id[NCCL_UNIQUE_ID_BYTES] = {0};
    if (rank == 0) {
      ncclUniqueId u;
      EESEN_NCCL_CHECK(rccl().GetUniqueId(&u));
      std::memcpy(id, u.internal, NCCL_UNIQUE_ID_BYTES);
    }
    comm_exchange(addr, port, rank, world, id, NCCL_UNIQUE_ID_BYTES, timeout_s);
    *out = new eesen_comm(device, id, rank, world);
  });
}
int eesen_comm_destroy(eesen_comm_t* comm) {
  return guard([&] { delete comm; });
}
int eesen_comm_info(eesen_comm_t* comm, int* rank, int* world) {
  return guard([&] { REQ_PTR(comm); if (rank) *rank = comm->rank; if (world) *world = comm->world; });
}
int eesen_comm_describe(eesen_comm_t* comm, char* json, int cap) {
  return guard([&] {
    REQ_PTR(comm); REQ_PTR(json);
    const std::string s = comm->describe();
    EESEN_REQUIRE(cap > (int)s.size(), EESEN_ERR_INVALID, "eesen_comm_describe: buffer too small (" + std::to_string(s.size() + 1) + " bytes needed)");
    std::memcpy(json, s.c_str(), s.size() + 1);
  });
}
int eesen_comm_allreduce_host(eesen_comm_t* comm, double* values, int n, int op) {
  return guard([&] { REQ_PTR(comm); REQ_PTR(values); comm->allreduce_host(values, n, op); });
}
int eesen_net_set_comm(eesen_net_t* net, eesen_comm_t* comm) {
  return guard([&] { REQ_PTR(net); net->set_comm(comm); });
}
int eesen_net_allreduce_grads(eesen_net_t* net, eesen_comm_t* comm) {
  return guard([&] { REQ_PTR(net); net->allreduce_grads(comm); });
}
int eesen_net_backpropagate_zero(eesen_net_t* net) {
  return guard([&] { REQ_PTR(net); net->backpropagate_zero(); });
}
int eesen_net_live_ranks(eesen_net_t* net, int* live) {
  return guard([&] { REQ_PTR(net); REQ_PTR(live); *live = net->live_ranks(); });
}
int eesen_net_bucket_order(eesen_net_t* net, int* layers_out, int cap, int* n) {
  return guard([&] {
    REQ_PTR(net); REQ_PTR(n);
    *n = (int)net->bucket_log.size();
    for (int i = 0; i < *n && i < cap && layers_out; ++i) layers_out[i] = net->bucket_log[i];
  });
}

}  // extern "C"
