// fake_rccl.hip — TEST INFRASTRUCTURE, never shipped, never loaded by the product on its own: a stand-in for librccl.so that
// lets TWO (or more) processes sharing ONE GPU run the collective protocol of csrc/cpmppi_comm.hip against a real peer.
// (RCCL itself refuses two ranks on one device; the pool hands out one GPU per box: without this the side-stream protocol of
// cpmppi_step_gather had only ever met a slow collective, never another rank - VERDICT r5, missing #1.)
//
// It exports exactly the eight symbols csrc/cpmppi_comm.hip binds with dlsym (load_rccl) and is handed to the library through the
// EXISTING `rccl_path` argument of cpmppi_comm_unique_id / cpmppi_comm_init - the product code path is the production one, only
// the collective library behind it differs.
//
// What it is: an intra-node all-gather the way RCCL does it over P2P mappings, reduced to its bones -
//   * ncclGetUniqueId      128 random bytes (a magic prefix + /dev/urandom)
//   * ncclCommInitRank     rendezvous through a directory named after the id (FAKE_RCCL_DIR, default /tmp): every rank allocates a
//                          MAILBOX (two payload slots + sequence words), publishes how to reach it, and maps every peer's:
//                            transport "ipc": device memory, hipIpcGetMemHandle / hipIpcOpenMemHandle (fine-grained);
//                            transport "shm": a POSIX shared-memory file mapped by both processes and hipHostRegister'ed
//                          (FAKE_RCCL_TRANSPORT; the same kernel runs over either)
//   * ncclAllGather        ONE kernel on the caller's stream: copy `send` into my mailbox slot, release "ready = n" (system scope),
//                          then for every peer: spin until its ready >= n, copy its slot into recv[peer], release "consumed = n" into
//                          ITS mailbox (flow control: a slot is rewritten two gathers later, after every peer has consumed it).
//                          Stream-ordered, no host involvement, device-side flags only - what a collective kernel does.
//   Every device-side spin is bounded (FAKE_RCCL_TIMEOUT_S, default 20 s): a peer that never arrives raises a sticky error word the
//   next call returns as ncclInternalError; the kernel never hangs the GPU.
#include <errno.h>
#include <fcntl.h>
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <time.h>
#include <unistd.h>

#include <string>

#include <rccl/rccl.h>      // the prototypes these definitions must match, and the types

namespace {

constexpr int MAX_RANKS = 8;
constexpr size_t SLOT_FLOATS = 1u << 20;                  // 4 MB per slot: far beyond [E_local, H] of any configuration here
constexpr uint32_t MAGIC = 0x46414b45u;                   // "FAKE"

struct Mailbox {
  uint32_t ready;                   // gathers whose payload is complete in slot[(n - 1) & 1]
  uint32_t pad0[15];
  uint32_t consumed[MAX_RANKS];     // consumed[p]: peer p has copied gather n out of my slot (written by p)
  uint32_t pad1[8];
  float slot[2][SLOT_FLOATS];
};

struct Peers {
  Mailbox* box[MAX_RANKS];          // device-visible address of every rank's mailbox (mine included)
};

struct RankFile {                   // what a rank publishes in <dir>/rank<r>
  uint32_t magic, rank, transport;  // 0 = ipc, 1 = shm
  uint32_t pid;
  hipIpcMemHandle_t handle;
  char shm_name[64];
};

struct FakeComm {
  int world = 0, rank = 0, device = 0, transport = 0;
  Mailbox* mine = nullptr;          // device address of my mailbox
  void* mine_host = nullptr;        // shm: host mapping
  size_t bytes = 0;
  Peers peers{};
  void* peer_host[MAX_RANKS] = {};  // shm: host mappings of the peers' files
  uint32_t* err = nullptr;          // pinned host word: a device-side spin gave up
  uint32_t seq = 0;                 // gathers enqueued
  unsigned long long timeout_ticks = 2000000000ull;
  std::string dir, shm_name;
};

std::string hex_of(const char* p, int n) {
  static const char* d = "0123456789abcdef";
  std::string s;
  for (int i = 0; i < n; ++i) { s += d[(p[i] >> 4) & 15]; s += d[p[i] & 15]; }
  return s;
}

double now_s() {
  timespec t;
  clock_gettime(CLOCK_MONOTONIC, &t);
  return (double)t.tv_sec + 1e-9 * (double)t.tv_nsec;
}

bool write_file_atomic(const std::string& path, const void* data, size_t n) {
  const std::string tmp = path + ".tmp";
  FILE* f = fopen(tmp.c_str(), "wb");
  if (!f) return false;
  const bool ok = fwrite(data, 1, n, f) == n;
  fclose(f);
  return ok && rename(tmp.c_str(), path.c_str()) == 0;
}

bool read_file(const std::string& path, void* data, size_t n) {
  FILE* f = fopen(path.c_str(), "rb");
  if (!f) return false;
  const bool ok = fread(data, 1, n, f) == n;
  fclose(f);
  return ok;
}

__device__ __forceinline__ bool spin_ge(const uint32_t* word, uint32_t need, unsigned long long timeout_ticks, uint32_t* err) {
  const uint64_t t0 = __builtin_amdgcn_s_memrealtime();
  while ((int32_t)(__hip_atomic_load(word, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) - need) < 0) {
    __builtin_amdgcn_s_sleep(32);
    if (__builtin_amdgcn_s_memrealtime() - t0 > timeout_ticks) {
      __hip_atomic_store(err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      return false;
    }
  }
  return true;
}

// peer payloads are read with system-scope loads: the slot was written by ANOTHER process' kernel (another VMID, possibly another
// XCD's L2); the acquire on `ready` orders, the scope keeps this CU's caches out of it
__device__ __forceinline__ float ld_sys(const float* p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
__device__ __forceinline__ void st_sys(float* p, float v) {
  __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

__global__ __launch_bounds__(256) void allgather_kernel(Peers peers, int world, int rank, uint32_t n, const float* send, float* recv,
                                                        size_t count, unsigned long long timeout_ticks, uint32_t* err,
                                                        unsigned long long delay_ticks) {
  __shared__ uint32_t ok;
  const uint32_t tid = threadIdx.x;
  Mailbox* mine = peers.box[rank];
  float* slot = mine->slot[(n - 1u) & 1u];
  if (tid == 0) {
    if (delay_ticks) {                                 // tests: this rank joins the collective late
      const uint64_t t0 = __builtin_amdgcn_s_memrealtime();
      while (__builtin_amdgcn_s_memrealtime() - t0 < delay_ticks) __builtin_amdgcn_s_sleep(64);
    }
    uint32_t good = 1u;
    if (n > 2u)                                        // the slot written now was read by every peer two gathers ago
      for (int p = 0; p < world && good; ++p)
        if (p != rank) good = spin_ge(&mine->consumed[p], n - 2u, timeout_ticks, err) ? 1u : 0u;
    ok = good;
  }
  __syncthreads();
  if (!ok) return;
  for (size_t i = tid; i < count; i += 256) {
    const float v = send[i];
    st_sys(slot + i, v);
    recv[(size_t)rank * count + i] = v;
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");
  __syncthreads();
  if (tid == 0) __hip_atomic_store(&mine->ready, n, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
  for (int p = 0; p < world; ++p) {
    if (p == rank) continue;
    Mailbox* theirs = peers.box[p];
    __syncthreads();
    if (tid == 0) ok = spin_ge(&theirs->ready, n, timeout_ticks, err) ? 1u : 0u;
    __syncthreads();
    if (!ok) return;
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");
    const float* src = theirs->slot[(n - 1u) & 1u];
    for (size_t i = tid; i < count; i += 256) recv[(size_t)p * count + i] = ld_sys(src + i);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");
    __syncthreads();
    if (tid == 0) __hip_atomic_store(&theirs->consumed[rank], n, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
  }
}

thread_local std::string g_err;
ncclResult_t failed(ncclResult_t r, const std::string& m) {
  g_err = m;
  fprintf(stderr, "[fake_rccl] %s\n", m.c_str());
  return r;
}

void release(FakeComm* c) {
  if (!c) return;
  for (int p = 0; p < c->world; ++p) {
    if (p == c->rank) continue;
    if (c->transport == 0 && c->peers.box[p]) (void)hipIpcCloseMemHandle(c->peers.box[p]);
    if (c->transport == 1 && c->peer_host[p]) { (void)hipHostUnregister(c->peer_host[p]); munmap(c->peer_host[p], c->bytes); }
  }
  if (c->transport == 0 && c->mine) (void)hipFree(c->mine);
  if (c->transport == 1 && c->mine_host) {
    (void)hipHostUnregister(c->mine_host);
    munmap(c->mine_host, c->bytes);
    shm_unlink(c->shm_name.c_str());
  }
  if (c->err) (void)hipHostFree(c->err);
  if (!c->dir.empty()) {
    unlink((c->dir + "/rank" + std::to_string(c->rank)).c_str());
    unlink((c->dir + "/mapped" + std::to_string(c->rank)).c_str());
    rmdir(c->dir.c_str());                              // (succeeds for the last rank out)
  }
  delete c;
}

}  // namespace

extern "C" {

ncclResult_t ncclGetVersion(int* version) {
  if (!version) return ncclInvalidArgument;
  *version = 99;                                        // (nobody mistakes this for an RCCL release)
  return ncclSuccess;
}

const char* ncclGetErrorString(ncclResult_t r) {
  if (r == ncclSuccess) return "no error";
  return g_err.empty() ? "fake_rccl error" : g_err.c_str();
}

ncclResult_t ncclGetUniqueId(ncclUniqueId* id) {
  if (!id) return ncclInvalidArgument;
  memset(id->internal, 0, sizeof(id->internal));
  memcpy(id->internal, &MAGIC, 4);
  FILE* f = fopen("/dev/urandom", "rb");
  if (!f || fread(id->internal + 4, 1, 16, f) != 16) {
    if (f) fclose(f);
    return failed(ncclSystemError, "ncclGetUniqueId: /dev/urandom");
  }
  fclose(f);
  return ncclSuccess;
}

ncclResult_t ncclCommInitRank(ncclComm_t* comm, int nranks, ncclUniqueId id, int rank) {
  if (!comm || nranks < 1 || nranks > MAX_RANKS || rank < 0 || rank >= nranks) return failed(ncclInvalidArgument, "ncclCommInitRank: bad argument");
  uint32_t magic;
  memcpy(&magic, id.internal, 4);
  if (magic != MAGIC) return failed(ncclInvalidArgument, "ncclCommInitRank: this id was not drawn by fake_rccl");
  FakeComm* c = new FakeComm();
  c->world = nranks; c->rank = rank;
  c->bytes = sizeof(Mailbox);
  if (hipGetDevice(&c->device) != hipSuccess) { delete c; return failed(ncclUnhandledCudaError, "hipGetDevice"); }
  const char* tr = getenv("FAKE_RCCL_TRANSPORT");
  c->transport = (tr && strcmp(tr, "shm") == 0) ? 1 : 0;
  if (const char* t = getenv("FAKE_RCCL_TIMEOUT_S")) c->timeout_ticks = (unsigned long long)(atof(t) * 1.0e8);
  const char* base = getenv("FAKE_RCCL_DIR");
  c->dir = std::string(base && base[0] ? base : "/tmp") + "/fake_rccl_" + hex_of(id.internal + 4, 16);
  if (mkdir(c->dir.c_str(), 0700) != 0 && errno != EEXIST) { const std::string m = "mkdir " + c->dir + ": " + strerror(errno); c->dir.clear(); release(c); return failed(ncclSystemError, m); }
  hipError_t e = hipHostMalloc((void**)&c->err, 64, hipHostMallocMapped | hipHostMallocCoherent);
  if (e != hipSuccess) { release(c); return failed(ncclUnhandledCudaError, std::string("hipHostMalloc: ") + hipGetErrorString(e)); }
  memset(c->err, 0, 64);
  RankFile me{};
  me.magic = MAGIC; me.rank = (uint32_t)rank; me.transport = (uint32_t)c->transport; me.pid = (uint32_t)getpid();
  if (c->transport == 0) {
    // fine-grained device memory: coherent across the XCDs' L2s for system-scope accesses - what RCCL allocates for its own flags
    e = hipExtMallocWithFlags((void**)&c->mine, c->bytes, hipDeviceMallocFinegrained);
    if (e == hipSuccess) e = hipMemset(c->mine, 0, offsetof(Mailbox, slot));
    if (e == hipSuccess) e = hipDeviceSynchronize();
    if (e == hipSuccess) e = hipIpcGetMemHandle(&me.handle, c->mine);
    if (e != hipSuccess) { const std::string m = std::string("mailbox (ipc): ") + hipGetErrorString(e); release(c); return failed(ncclUnhandledCudaError, m); }
  } else {
    c->shm_name = "/fake_rccl_" + hex_of(id.internal + 4, 8) + "_" + std::to_string(rank);
    snprintf(me.shm_name, sizeof(me.shm_name), "%s", c->shm_name.c_str());
    const int fd = shm_open(c->shm_name.c_str(), O_CREAT | O_RDWR | O_TRUNC, 0600);
    if (fd < 0 || ftruncate(fd, (off_t)c->bytes) != 0) { const std::string m = std::string("shm_open: ") + strerror(errno); if (fd >= 0) close(fd); release(c); return failed(ncclSystemError, m); }
    c->mine_host = mmap(nullptr, c->bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
    close(fd);
    if (c->mine_host == MAP_FAILED) { c->mine_host = nullptr; release(c); return failed(ncclSystemError, "mmap"); }
    memset(c->mine_host, 0, offsetof(Mailbox, slot));
    e = hipHostRegister(c->mine_host, c->bytes, hipHostRegisterMapped | hipHostRegisterPortable);
    void* dp = nullptr;
    if (e == hipSuccess) e = hipHostGetDevicePointer(&dp, c->mine_host, 0);
    if (e != hipSuccess) { const std::string m = std::string("mailbox (shm): ") + hipGetErrorString(e); release(c); return failed(ncclUnhandledCudaError, m); }
    c->mine = (Mailbox*)dp;
  }
  c->peers.box[rank] = c->mine;
  if (!write_file_atomic(c->dir + "/rank" + std::to_string(rank), &me, sizeof(me))) { release(c); return failed(ncclSystemError, "rendezvous: cannot publish this rank"); }
  const double t0 = now_s(), limit = 120.0;
  for (int p = 0; p < nranks; ++p) {
    if (p == rank) continue;
    RankFile them{};
    while (!read_file(c->dir + "/rank" + std::to_string(p), &them, sizeof(them))) {
      if (now_s() - t0 > limit) { release(c); return failed(ncclSystemError, "rendezvous: rank " + std::to_string(p) + " never arrived"); }
      usleep(2000);
    }
    if (them.magic != MAGIC || them.rank != (uint32_t)p || them.transport != (uint32_t)c->transport) { release(c); return failed(ncclInternalError, "rendezvous: bad rank file"); }
    if (c->transport == 0) {
      void* dp = nullptr;
      e = hipIpcOpenMemHandle(&dp, them.handle, hipIpcMemLazyEnablePeerAccess);
      if (e != hipSuccess) { const std::string m = std::string("hipIpcOpenMemHandle: ") + hipGetErrorString(e); release(c); return failed(ncclUnhandledCudaError, m); }
      c->peers.box[p] = (Mailbox*)dp;
    } else {
      const int fd = shm_open(them.shm_name, O_RDWR, 0600);
      void* hp = fd >= 0 ? mmap(nullptr, c->bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0) : MAP_FAILED;
      if (fd >= 0) close(fd);
      if (hp == MAP_FAILED) { release(c); return failed(ncclSystemError, std::string("peer shm: ") + strerror(errno)); }
      c->peer_host[p] = hp;
      void* dp = nullptr;
      e = hipHostRegister(hp, c->bytes, hipHostRegisterMapped | hipHostRegisterPortable);
      if (e == hipSuccess) e = hipHostGetDevicePointer(&dp, hp, 0);
      if (e != hipSuccess) { const std::string m = std::string("peer mailbox (shm): ") + hipGetErrorString(e); release(c); return failed(ncclUnhandledCudaError, m); }
      c->peers.box[p] = (Mailbox*)dp;
    }
  }
  // second phase: nobody leaves (and starts tearing down files) before everybody has mapped everybody
  const char one = 1;
  write_file_atomic(c->dir + "/mapped" + std::to_string(rank), &one, 1);
  for (int p = 0; p < nranks; ++p) {
    char b;
    while (!read_file(c->dir + "/mapped" + std::to_string(p), &b, 1)) {
      if (now_s() - t0 > limit) { release(c); return failed(ncclSystemError, "rendezvous: rank " + std::to_string(p) + " never finished mapping"); }
      usleep(2000);
    }
  }
  *comm = reinterpret_cast<ncclComm_t>(c);
  return ncclSuccess;
}

ncclResult_t ncclCommDestroy(ncclComm_t comm) {
  FakeComm* c = reinterpret_cast<FakeComm*>(comm);
  if (!c) return ncclInvalidArgument;
  (void)hipDeviceSynchronize();
  // leave the files until every rank's kernels are done with this rank's mailbox: the caller's own protocol (cpmppi_comm_sync on
  // every rank + a barrier of its own) orders that; here a short grace so that a peer's last all-gather can still read the slot
  usleep(50000);
  release(c);
  return ncclSuccess;
}

ncclResult_t ncclCommCount(const ncclComm_t comm, int* count) {
  if (!comm || !count) return ncclInvalidArgument;
  *count = reinterpret_cast<const FakeComm*>(comm)->world;
  return ncclSuccess;
}

ncclResult_t ncclCommUserRank(const ncclComm_t comm, int* rank) {
  if (!comm || !rank) return ncclInvalidArgument;
  *rank = reinterpret_cast<const FakeComm*>(comm)->rank;
  return ncclSuccess;
}

ncclResult_t ncclAllGather(const void* send, void* recv, size_t count, ncclDataType_t type, ncclComm_t comm, hipStream_t stream) {
  FakeComm* c = reinterpret_cast<FakeComm*>(comm);
  if (!c || !send || !recv) return failed(ncclInvalidArgument, "ncclAllGather: null argument");
  if (type != ncclFloat) return failed(ncclInvalidArgument, "ncclAllGather: fake_rccl gathers ncclFloat only");
  if (count > SLOT_FLOATS) return failed(ncclInvalidArgument, "ncclAllGather: more than a mailbox slot holds");
  if (__atomic_load_n(c->err, __ATOMIC_ACQUIRE) != 0u) return failed(ncclInternalError, "ncclAllGather: an earlier all-gather gave up waiting for a peer");
  const uint32_t n = ++c->seq;
  unsigned long long delay = 0;
  if (const char* d = getenv("FAKE_RCCL_DELAY_US")) {            // "<gather number>:<microseconds>": that gather joins late on this rank
    unsigned g = 0, us = 0, permille = 0, seed = 0;
    if (sscanf(d, "rand:%u:%u:%u", &permille, &us, &seed) == 3) {
      // "rand:<per mille>:<max microseconds>:<seed>": a random subset of the gathers joins late by a random time (stress runs)
      uint32_t x = (seed * 2654435761u) ^ (n * 40503u) ^ ((uint32_t)c->rank * 0x9E3779B9u);
      x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
      if (x % 1000u < permille) delay = (unsigned long long)((x >> 10) % (us + 1u)) * 100ull;
    } else if (sscanf(d, "%u:%u", &g, &us) == 2 && g == n) {
      delay = (unsigned long long)us * 100ull;
    }
  }
  hipLaunchKernelGGL(allgather_kernel, dim3(1), dim3(256), 0, stream, c->peers, c->world, c->rank, n, (const float*)send, (float*)recv,
                     count, c->timeout_ticks, c->err, delay);
  const hipError_t e = hipGetLastError();
  if (e != hipSuccess) return failed(ncclUnhandledCudaError, std::string("allgather_kernel: ") + hipGetErrorString(e));
  return ncclSuccess;
}

}  // extern "C"
