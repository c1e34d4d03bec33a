// TEST DOUBLE of the seven RCCL entry points mfgpu_dist.hip calls, for tests/test_gpu_dist_rccl_path.py: lets TWO
// PROCESSES ON ONE GPU run the library's real RCCL transport path (ncclCommInitRank, grouped ncclSend / ncclRecv on the
// side stream, the events around them) -- RCCL itself refuses two ranks of a communicator on one device.  Loaded with
// LD_PRELOAD in front of librccl.so; never part of the product.  Data moves through a POSIX shared-memory segment:
// a send = stream sync + device-to-host copy into the pair's slot + sequence flag; a recv = wait for the flag +
// host-to-device copy.  ncclGroupEnd runs the group's sends, then its receives (no deadlock between two ranks that
// both send first), and synchronises the stream: coarser than RCCL in time, identical in what lands where.
#include <dlfcn.h>
#include <fcntl.h>
#include <sys/mman.h>
#include <unistd.h>

#include <atomic>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

extern "C" {
typedef enum { ncclSuccess = 0, ncclUnhandledCudaError = 1, ncclSystemError = 2, ncclInternalError = 3, ncclInvalidArgument = 4 } ncclResult_t;
typedef enum { ncclInt8 = 0, ncclUint8 = 1, ncclInt32 = 2, ncclUint32 = 3, ncclInt64 = 4, ncclUint64 = 5, ncclFloat16 = 6,
               ncclFloat32 = 7, ncclFloat64 = 8 } ncclDataType_t;
typedef struct { char internal[128]; } ncclUniqueId;
struct fake_comm;
typedef struct fake_comm *ncclComm_t;
}

// The HIP runtime is NOT linked: the three calls are looked up at run time in whatever HIP runtime the process has
// loaded (PyTorch bundles its own libamdhip64; a second runtime in the process would not know libmfgpu's kernels).
typedef void *hipStream_t;
namespace {
typedef int (*memcpy_fn)(void *, const void *, size_t, int);
typedef int (*memcpy_async_fn)(void *, const void *, size_t, int, hipStream_t);
typedef int (*sync_fn)(hipStream_t);
template <typename F>
F hip_sym(const char *name) {
  void *p = dlsym(RTLD_DEFAULT, name);
  if (!p) {  // loaded RTLD_LOCAL (a dependency of a ctypes-loaded library): ask the already-loaded runtime itself
    for (const char *lib : {"libamdhip64.so.7", "libamdhip64.so.6", "libamdhip64.so"}) {
      void *h = dlopen(lib, RTLD_NOLOAD | RTLD_NOW);
      if (h && (p = dlsym(h, name))) break;
    }
  }
  if (!p) {
    std::fprintf(stderr, "fake_rccl: %s not found (no HIP runtime loaded yet?)\n", name);
    std::abort();
  }
  return (F)p;
}
int hipMemcpy_(void *d, const void *s, size_t n, int kind) { static memcpy_fn f = hip_sym<memcpy_fn>("hipMemcpy"); return f(d, s, n, kind); }
int hipMemcpyAsync_(void *d, const void *s, size_t n, int kind, hipStream_t st) {
  static memcpy_async_fn f = hip_sym<memcpy_async_fn>("hipMemcpyAsync");
  return f(d, s, n, kind, st);
}
int hipStreamSynchronize_(hipStream_t st) { static sync_fn f = hip_sym<sync_fn>("hipStreamSynchronize"); return f(st); }
constexpr int kH2D = 1, kD2H = 2;  // hipMemcpyHostToDevice, hipMemcpyDeviceToHost
}  // namespace

namespace {
constexpr size_t kSlotBytes = 8u << 20;  // per ordered pair (src, dst)
struct Slot {
  std::atomic<uint64_t> seq_written, seq_read;
  uint64_t bytes;
  unsigned char data[kSlotBytes];
};
struct Header {
  std::atomic<int> arrived;
};
struct Op {
  bool send;
  void *buf;
  size_t bytes;
  int peer;
  hipStream_t stream;
};
thread_local std::vector<Op> g_ops;
thread_local int g_depth = 0;
thread_local struct fake_comm *g_last = nullptr;  // the communicator of the open group (mfgpu_dist uses one)
size_t dsize(ncclDataType_t t) { return t == ncclFloat64 || t == ncclInt64 || t == ncclUint64 ? 8 : t == ncclFloat16 ? 2 : t <= ncclUint8 ? 1 : 4; }
}  // namespace

struct fake_comm {
  int rank, world;
  Header *hdr;
  Slot *slots;  // [world * world], slot(src, dst) = src * world + dst
  size_t map_bytes;
  std::vector<uint64_t> sent, received;  // per peer
};

static ncclResult_t run_group(ncclComm_t c);

extern "C" {

const char *ncclGetErrorString(ncclResult_t r) { return r == ncclSuccess ? "no error" : "fake RCCL error"; }

ncclResult_t ncclGetUniqueId(ncclUniqueId *id) {
  std::memset(id, 0, sizeof(*id));
  std::snprintf(id->internal, sizeof(id->internal), "/mfgpu_fake_rccl_%d_%ld", (int)getpid(), (long)random());
  return ncclSuccess;
}

ncclResult_t ncclCommInitRank(ncclComm_t *comm, int world, ncclUniqueId id, int rank) {
  const size_t bytes = sizeof(Header) + (size_t)world * world * sizeof(Slot);
  int fd = shm_open(id.internal, O_CREAT | O_RDWR, 0600);
  if (fd < 0 || ftruncate(fd, (off_t)bytes) != 0) return ncclSystemError;
  void *p = mmap(nullptr, bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
  close(fd);
  if (p == MAP_FAILED) return ncclSystemError;
  fake_comm *c = new fake_comm();
  c->rank = rank;
  c->world = world;
  c->hdr = (Header *)p;
  c->slots = (Slot *)((char *)p + sizeof(Header));
  c->map_bytes = bytes;
  c->sent.assign(world, 0);
  c->received.assign(world, 0);
  c->hdr->arrived.fetch_add(1);
  while (c->hdr->arrived.load() < world) usleep(100);  // every rank has mapped the (zero-filled) segment
  if (rank == 0) shm_unlink(id.internal);
  *comm = c;
  return ncclSuccess;
}

ncclResult_t ncclCommDestroy(ncclComm_t c) {
  if (c) {
    munmap((void *)c->hdr, c->map_bytes);
    delete c;
  }
  return ncclSuccess;
}

ncclResult_t ncclGroupStart() {
  ++g_depth;
  return ncclSuccess;
}

// fault injection for the callers' fallback paths: FAKE_RCCL_FAULT=error (every send fails) | corrupt (one entry in
// the middle of every message is changed on the way)
static int fault() {
  static const int f = [] {
    const char *e = getenv("FAKE_RCCL_FAULT");
    return !e ? 0 : !strcmp(e, "error") ? 1 : !strcmp(e, "corrupt") ? 2 : 0;
  }();
  return f;
}

ncclResult_t ncclSend(const void *buf, size_t count, ncclDataType_t dt, int peer, ncclComm_t c, hipStream_t st) {
  if (!c || peer < 0 || peer >= c->world || count * dsize(dt) > kSlotBytes) return ncclInvalidArgument;
  if (fault() == 1) return ncclInternalError;
  g_ops.push_back({true, const_cast<void *>(buf), count * dsize(dt), peer, st});
  g_last = c;
  return g_depth ? ncclSuccess : run_group(c);
}

ncclResult_t ncclRecv(void *buf, size_t count, ncclDataType_t dt, int peer, ncclComm_t c, hipStream_t st) {
  if (!c || peer < 0 || peer >= c->world || count * dsize(dt) > kSlotBytes) return ncclInvalidArgument;
  g_ops.push_back({false, buf, count * dsize(dt), peer, st});
  g_last = c;
  return g_depth ? ncclSuccess : run_group(c);
}

ncclResult_t ncclGroupEnd() {
  if (--g_depth > 0) return ncclSuccess;
  return g_last ? run_group(g_last) : ncclSuccess;
}

}  // extern "C"

static ncclResult_t run_group(ncclComm_t c) {
  std::vector<Op> ops;
  ops.swap(g_ops);
  for (const Op &o : ops)
    if (o.send) {
      Slot &s = c->slots[(size_t)c->rank * c->world + o.peer];
      const uint64_t n = ++c->sent[o.peer];
      while (s.seq_read.load() < n - 1) usleep(50);  // the previous message of this pair has been taken
      if (hipStreamSynchronize_(o.stream) != 0) return ncclUnhandledCudaError;  // everything queued before the send
      if (hipMemcpy_(s.data, o.buf, o.bytes, kD2H) != 0) return ncclUnhandledCudaError;
      // (the exponent of a double in the middle of the plane: its first entries are Dirichlet rows, which are not summed)
      if (fault() == 2 && o.bytes > 15) ((unsigned char *)s.data)[(o.bytes / 16) * 8 + 7] ^= 0x40;
      s.bytes = o.bytes;
      s.seq_written.store(n);
    }
  for (const Op &o : ops)
    if (!o.send) {
      Slot &s = c->slots[(size_t)o.peer * c->world + c->rank];
      const uint64_t n = ++c->received[o.peer];
      while (s.seq_written.load() < n) usleep(50);
      if (s.bytes != o.bytes) return ncclInvalidArgument;  // the two sides disagree on the plane's size
      if (hipMemcpyAsync_(o.buf, s.data, o.bytes, kH2D, o.stream) != 0) return ncclUnhandledCudaError;
      if (hipStreamSynchronize_(o.stream) != 0) return ncclUnhandledCudaError;
      s.seq_read.store(n);
    }
  return ncclSuccess;
}

