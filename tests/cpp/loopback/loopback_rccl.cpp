// loopback_rccl.cpp — TEST INFRASTRUCTURE, never shipped and never measured.
//
// A stand-in for the eleven RCCL entry points libqsx.so binds at run time (quickstep_amd/csrc/comm.hip, rccl()): with
// QSX_RCCL_LIBRARY=<this library> the multi-GPU entry points of the C ABI (qsx_exchange_counts, qsx_alltoallv,
// qsx_allgather, qsx_bitmap_allreduce_or, qsx_agg_reduce_scatter, qsx_agg_allgather_merge) run at world sizes 2 and 3 on
// a box with ONE GPU — RCCL itself refuses two ranks on one device.  The pattern is the reference's distributed test
// runner, which runs Shiftboss / Foreman "nodes" as threads of one process over an in-process message bus
// (query_optimizer/tests/DistributedExecutionGeneratorTestRunner.cpp:72-150).
//
// Ranks are threads or processes that share a device; data moves rank -> host file under /dev/shm -> rank.  Every
// communication call is a COLLECTIVE ROUND of all ranks of the communicator (what the callers in libqsx.so do): stream
// synchronise, publish what this rank sends, barrier, fetch what the peers published for this rank, barrier.  That is a
// legal (if slow and fully synchronous) implementation of RCCL's stream-ordered semantics for programs in which all ranks
// issue the same sequence of calls.  Not supported, and refused with ncclInvalidUsage: groups that only some ranks
// join, collectives other than send / recv inside a group, reductions other than sum / min / max on 64-bit words, 32-bit
// words, bytes and doubles.
#include <fcntl.h>
#include <sched.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <time.h>
#include <unistd.h>

#include <atomic>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include <hip/hip_runtime_api.h>
#include <rccl/rccl.h>

// LOOPBACK_HOST_MEMORY: "device" buffers are host memory (the protocol's own self-test, tests/cpp/loopback/selftest.cpp,
// which runs where there is no GPU)
#ifdef LOOPBACK_HOST_MEMORY
#define hipMemcpy(dst, src, bytes, kind) (std::memcpy((dst), (src), (bytes)), hipSuccess)
#define hipStreamSynchronize(stream) ((void)(stream), hipSuccess)
#endif

namespace {

constexpr uint32_t kMagic = 0x51534c42;   // "QSLB"
constexpr int kMaxWorld = 16;
// LOOPBACK_RCCL_TIMEOUT_S shortens the wait for a peer that never arrives (tests of the failure paths)
const double kTimeoutSeconds = std::getenv("LOOPBACK_RCCL_TIMEOUT_S") != nullptr ? std::atof(std::getenv("LOOPBACK_RCCL_TIMEOUT_S")) : 120.0;

struct Control {
  std::atomic<uint32_t> magic;
  std::atomic<int> joined;
  std::atomic<int> left;
  std::atomic<int> arrived;
  std::atomic<int> generation;
  std::atomic<int> failed;
};

struct SendRecord {
  uint64_t dst, bytes, offset;
};

struct Op {
  bool is_send;
  const void *send;
  void *recv;
  size_t bytes;
  int peer;
  hipStream_t stream;
};

struct Comm {
  Control *control = nullptr;
  std::string name;
  int world = 0, rank = 0;
  int files[kMaxWorld];
};

thread_local int group_depth = 0;
thread_local std::vector<Op> group_ops;
thread_local Comm *group_comm = nullptr;
// A group this rank put nothing into is still a round of its communicator: the one this thread used last, else — rank
// processes issue their collectives from whichever Worker thread runs the work order — the process's communicator.
thread_local Comm *last_comm = nullptr;
std::atomic<Comm *> process_comm{nullptr};

double now_seconds() {
  timespec ts;
  clock_gettime(CLOCK_MONOTONIC, &ts);
  return ts.tv_sec + ts.tv_nsec * 1e-9;
}

std::atomic<unsigned> id_counter{0};

size_t type_bytes(ncclDataType_t t) {
  switch (t) {
    case ncclInt8: case ncclUint8: return 1;
    case ncclInt32: case ncclUint32: case ncclFloat32: return 4;
    case ncclInt64: case ncclUint64: case ncclFloat64: return 8;
    default: return 0;
  }
}

bool barrier(Comm *c) {
  Control *k = c->control;
  const int gen = k->generation.load(std::memory_order_acquire);
  if (k->arrived.fetch_add(1, std::memory_order_acq_rel) + 1 == c->world) {
    k->arrived.store(0, std::memory_order_relaxed);
    k->generation.store(gen + 1, std::memory_order_release);
    return true;
  }
  const double start = now_seconds();
  int spins = 0;
  while (k->generation.load(std::memory_order_acquire) == gen) {
    if (k->failed.load(std::memory_order_relaxed) != 0) return false;
    if (++spins > 200) {
      sched_yield();
      if ((spins & 1023) == 0 && now_seconds() - start > kTimeoutSeconds) {
        k->failed.store(1, std::memory_order_relaxed);
        std::fprintf(stderr, "loopback_rccl: rank %d waited %.0f s for its peers — they are not issuing the same calls\n", c->rank, kTimeoutSeconds);
        return false;
      }
    }
  }
  return true;
}

bool write_all(int fd, const void *data, size_t bytes, off_t at) {
  const char *p = static_cast<const char *>(data);
  while (bytes > 0) {
    const ssize_t w = pwrite(fd, p, bytes, at);
    if (w <= 0) return false;
    p += w;
    at += w;
    bytes -= static_cast<size_t>(w);
  }
  return true;
}

bool read_all(int fd, void *data, size_t bytes, off_t at) {
  char *p = static_cast<char *>(data);
  while (bytes > 0) {
    const ssize_t r = pread(fd, p, bytes, at);
    if (r <= 0) return false;
    p += r;
    at += r;
    bytes -= static_cast<size_t>(r);
  }
  return true;
}

// device -> this rank's file at `at`
bool publish(Comm *c, const void *dev, size_t bytes, off_t at, std::vector<char> *staging) {
  if (bytes == 0) return true;
  staging->resize(bytes);
  if (hipMemcpy(staging->data(), dev, bytes, hipMemcpyDeviceToHost) != hipSuccess) return false;
  return write_all(c->files[c->rank], staging->data(), bytes, at);
}

template <typename T>
void reduce_into(T *acc, const T *in, size_t n, ncclRedOp_t op) {
  for (size_t i = 0; i < n; ++i) {
    if (op == ncclSum) acc[i] = static_cast<T>(acc[i] + in[i]);
    else if (op == ncclMin) acc[i] = in[i] < acc[i] ? in[i] : acc[i];
    else acc[i] = in[i] > acc[i] ? in[i] : acc[i];
  }
}

bool reduce_typed(void *acc, const void *in, size_t n, ncclDataType_t t, ncclRedOp_t op) {
  switch (t) {
    case ncclInt64: reduce_into(static_cast<int64_t *>(acc), static_cast<const int64_t *>(in), n, op); return true;
    case ncclUint64: reduce_into(static_cast<uint64_t *>(acc), static_cast<const uint64_t *>(in), n, op); return true;
    case ncclFloat64: reduce_into(static_cast<double *>(acc), static_cast<const double *>(in), n, op); return true;
    case ncclInt32: reduce_into(static_cast<int32_t *>(acc), static_cast<const int32_t *>(in), n, op); return true;
    case ncclUint32: reduce_into(static_cast<uint32_t *>(acc), static_cast<const uint32_t *>(in), n, op); return true;
    case ncclFloat32: reduce_into(static_cast<float *>(acc), static_cast<const float *>(in), n, op); return true;
    case ncclUint8: reduce_into(static_cast<uint8_t *>(acc), static_cast<const uint8_t *>(in), n, op); return true;
    default: return false;
  }
}

// One round of grouped sends / receives.  This rank's file: [u64 count][SendRecord x count][payloads].
ncclResult_t run_group(Comm *c, std::vector<Op> &ops) {
  // a rank that gives up tells its peers (they leave their barrier with an error instead of waiting for the timeout)
  auto fail = [&](ncclResult_t why) {
    c->control->failed.store(1, std::memory_order_relaxed);
    return why;
  };
  for (const Op &op : ops) {
    if (hipStreamSynchronize(op.stream) != hipSuccess) return fail(ncclUnhandledCudaError);
  }
  std::vector<SendRecord> records;
  uint64_t payload_at = 0;
  for (const Op &op : ops) {
    if (op.is_send) records.push_back({static_cast<uint64_t>(op.peer), op.bytes, 0});
  }
  const uint64_t header = 8 + sizeof(SendRecord) * records.size();
  payload_at = header;
  std::vector<char> staging;
  size_t r = 0;
  for (const Op &op : ops) {
    if (!op.is_send) continue;
    records[r].offset = payload_at;
    if (!publish(c, op.send, op.bytes, static_cast<off_t>(payload_at), &staging)) return fail(ncclSystemError);
    payload_at += op.bytes;
    ++r;
  }
  const uint64_t count = records.size();
  if (!write_all(c->files[c->rank], &count, 8, 0)) return fail(ncclSystemError);
  if (count > 0 && !write_all(c->files[c->rank], records.data(), sizeof(SendRecord) * count, 8)) return fail(ncclSystemError);
  if (!barrier(c)) return ncclSystemError;
  // the k-th receive from peer p takes the k-th record of p's file addressed to this rank
  std::vector<uint64_t> taken(static_cast<size_t>(c->world), 0);
  ncclResult_t result = ncclSuccess;
  for (const Op &op : ops) {
    if (op.is_send || result != ncclSuccess) continue;
    uint64_t peer_count = 0;
    if (!read_all(c->files[op.peer], &peer_count, 8, 0)) { result = ncclSystemError; break; }
    std::vector<SendRecord> theirs(peer_count);
    if (peer_count > 0 && !read_all(c->files[op.peer], theirs.data(), sizeof(SendRecord) * peer_count, 8)) { result = ncclSystemError; break; }
    uint64_t seen = 0;
    const SendRecord *match = nullptr;
    for (const SendRecord &rec : theirs) {
      if (rec.dst != static_cast<uint64_t>(c->rank)) continue;
      if (seen++ == taken[static_cast<size_t>(op.peer)]) { match = &rec; break; }
    }
    if (match == nullptr || match->bytes != op.bytes) {
      std::fprintf(stderr, "loopback_rccl: rank %d expects %zu bytes from rank %d, which sends %s\n", c->rank, op.bytes, op.peer,
                   match == nullptr ? "nothing" : "another size");
      result = ncclInvalidUsage;
      break;
    }
    ++taken[static_cast<size_t>(op.peer)];
    if (op.bytes == 0) continue;
    staging.resize(op.bytes);
    if (!read_all(c->files[op.peer], staging.data(), op.bytes, static_cast<off_t>(match->offset))) { result = ncclSystemError; break; }
    if (hipMemcpy(op.recv, staging.data(), op.bytes, hipMemcpyHostToDevice) != hipSuccess) { result = ncclUnhandledCudaError; break; }
  }
  if (result != ncclSuccess) c->control->failed.store(1, std::memory_order_relaxed);
  if (!barrier(c)) return result != ncclSuccess ? result : ncclSystemError;
  return result;
}

// every rank publishes `bytes` at offset 0 of its file; fetch(r, host) reads rank r's contribution
template <typename Consume>
ncclResult_t run_collective(Comm *c, const void *send, size_t bytes, hipStream_t stream, Consume consume) {
  if (group_depth > 0) return ncclInvalidUsage;
  std::vector<char> staging;
  if (hipStreamSynchronize(stream) != hipSuccess || !publish(c, send, bytes, 0, &staging)) {
    c->control->failed.store(1, std::memory_order_relaxed);
    return ncclSystemError;
  }
  if (!barrier(c)) return ncclSystemError;
  ncclResult_t result = ncclSuccess;
  std::vector<char> theirs(bytes);
  for (int r = 0; r < c->world && result == ncclSuccess; ++r) {
    if (bytes > 0 && !read_all(c->files[r], theirs.data(), bytes, 0)) { result = ncclSystemError; break; }
    result = consume(r, theirs.data());
  }
  if (result != ncclSuccess) c->control->failed.store(1, std::memory_order_relaxed);
  if (!barrier(c)) return result != ncclSuccess ? result : ncclSystemError;
  return result;
}

}  // namespace

extern "C" {

ncclResult_t ncclGetUniqueId(ncclUniqueId *id) {
  if (id == nullptr) return ncclInvalidArgument;
  std::memset(id, 0, sizeof(*id));
  std::snprintf(id->internal, sizeof(id->internal), "/qsx_loopback_%d_%u_%lld", static_cast<int>(getpid()), id_counter.fetch_add(1),
                static_cast<long long>(now_seconds() * 1e6));
  const int fd = shm_open(id->internal, O_CREAT | O_EXCL | O_RDWR, 0600);
  if (fd < 0 || ftruncate(fd, sizeof(Control)) != 0) return ncclSystemError;
  void *p = mmap(nullptr, sizeof(Control), PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
  close(fd);
  if (p == MAP_FAILED) return ncclSystemError;
  Control *k = new (p) Control;
  k->joined.store(0);
  k->left.store(0);
  k->arrived.store(0);
  k->generation.store(0);
  k->failed.store(0);
  k->magic.store(kMagic, std::memory_order_release);
  munmap(p, sizeof(Control));
  return ncclSuccess;
}

ncclResult_t ncclCommInitRank(ncclComm_t *out, int world, ncclUniqueId id, int rank) {
  if (out == nullptr || world < 1 || world > kMaxWorld || rank < 0 || rank >= world) return ncclInvalidArgument;
  id.internal[sizeof(id.internal) - 1] = 0;
  const int fd = shm_open(id.internal, O_RDWR, 0600);
  if (fd < 0) return ncclInvalidArgument;
  void *p = mmap(nullptr, sizeof(Control), PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
  close(fd);
  if (p == MAP_FAILED) return ncclSystemError;
  Comm *c = new Comm;
  c->control = static_cast<Control *>(p);
  if (c->control->magic.load(std::memory_order_acquire) != kMagic) {
    munmap(p, sizeof(Control));
    delete c;
    return ncclInvalidArgument;
  }
  c->name = id.internal;
  c->world = world;
  c->rank = rank;
  for (int r = 0; r < kMaxWorld; ++r) c->files[r] = -1;
  const std::string mine = c->name + "." + std::to_string(rank);
  c->files[rank] = shm_open(mine.c_str(), O_CREAT | O_RDWR, 0600);
  if (c->files[rank] < 0) return ncclSystemError;
  c->control->joined.fetch_add(1);
  if (!barrier(c)) return ncclSystemError;             // every rank's file exists
  for (int r = 0; r < world; ++r) {
    if (r == rank) continue;
    c->files[r] = shm_open((c->name + "." + std::to_string(r)).c_str(), O_RDONLY, 0600);
    if (c->files[r] < 0) return ncclSystemError;
  }
  if (!barrier(c)) return ncclSystemError;
  *out = reinterpret_cast<ncclComm_t>(c);
  last_comm = c;
  process_comm.store(c);
  return ncclSuccess;
}

ncclResult_t ncclCommDestroy(ncclComm_t comm) {
  Comm *c = reinterpret_cast<Comm *>(comm);
  if (c == nullptr) return ncclSuccess;
  if (last_comm == c) last_comm = nullptr;
  Comm *expected = c;
  process_comm.compare_exchange_strong(expected, nullptr);
  for (int r = 0; r < c->world; ++r) {
    if (c->files[r] >= 0) close(c->files[r]);
  }
  shm_unlink((c->name + "." + std::to_string(c->rank)).c_str());
  if (c->control->left.fetch_add(1) + 1 == c->world) shm_unlink(c->name.c_str());
  munmap(c->control, sizeof(Control));
  delete c;
  return ncclSuccess;
}

// A rank that gives a step up: its peers leave their barriers with an error (the control block's `failed` word) instead of
// waiting for it; the communicator's files are released like ncclCommDestroy does.
ncclResult_t ncclCommAbort(ncclComm_t comm) {
  Comm *c = reinterpret_cast<Comm *>(comm);
  if (c == nullptr) return ncclSuccess;
  c->control->failed.store(1, std::memory_order_relaxed);
  return ncclCommDestroy(comm);
}

ncclResult_t ncclGroupStart() {
  ++group_depth;
  return ncclSuccess;
}

ncclResult_t ncclGroupEnd() {
  if (group_depth <= 0) return ncclInvalidUsage;
  if (--group_depth > 0) return ncclSuccess;
  ncclResult_t result = ncclSuccess;
  if (group_comm == nullptr) group_comm = last_comm != nullptr ? last_comm : process_comm.load();
  if (group_comm != nullptr) result = run_group(group_comm, group_ops);
  group_ops.clear();
  group_comm = nullptr;
  return result;
}

static ncclResult_t queue_op(Op op, ncclComm_t comm) {
  Comm *c = reinterpret_cast<Comm *>(comm);
  if (c == nullptr || op.peer < 0 || op.peer >= c->world) return ncclInvalidArgument;
  if (group_comm != nullptr && group_comm != c) return ncclInvalidUsage;
  last_comm = c;
  if (group_depth == 0) {          // a lone send / recv is a group of one
    std::vector<Op> one{op};
    return run_group(c, one);
  }
  group_comm = c;
  group_ops.push_back(op);
  return ncclSuccess;
}

ncclResult_t ncclSend(const void *sendbuff, size_t count, ncclDataType_t datatype, int peer, ncclComm_t comm, hipStream_t stream) {
  const size_t width = type_bytes(datatype);
  if (width == 0) return ncclInvalidArgument;
  return queue_op(Op{true, sendbuff, nullptr, count * width, peer, stream}, comm);
}

ncclResult_t ncclRecv(void *recvbuff, size_t count, ncclDataType_t datatype, int peer, ncclComm_t comm, hipStream_t stream) {
  const size_t width = type_bytes(datatype);
  if (width == 0) return ncclInvalidArgument;
  return queue_op(Op{false, nullptr, recvbuff, count * width, peer, stream}, comm);
}

ncclResult_t ncclAllGather(const void *sendbuff, void *recvbuff, size_t sendcount, ncclDataType_t datatype, ncclComm_t comm, hipStream_t stream) {
  Comm *c = reinterpret_cast<Comm *>(comm);
  const size_t bytes = sendcount * type_bytes(datatype);
  if (c == nullptr || type_bytes(datatype) == 0) return ncclInvalidArgument;
  return run_collective(c, sendbuff, bytes, stream, [&](int r, const char *theirs) {
    if (bytes == 0) return ncclSuccess;
    return hipMemcpy(static_cast<char *>(recvbuff) + static_cast<size_t>(r) * bytes, theirs, bytes, hipMemcpyHostToDevice) == hipSuccess
               ? ncclSuccess
               : ncclUnhandledCudaError;
  });
}

ncclResult_t ncclReduceScatter(const void *sendbuff, void *recvbuff, size_t recvcount, ncclDataType_t datatype, ncclRedOp_t op, ncclComm_t comm,
                               hipStream_t stream) {
  Comm *c = reinterpret_cast<Comm *>(comm);
  const size_t width = type_bytes(datatype);
  if (c == nullptr || width == 0 || (op != ncclSum && op != ncclMin && op != ncclMax)) return ncclInvalidArgument;
  const size_t mine = recvcount * width, all = mine * static_cast<size_t>(c->world);
  std::vector<char> acc(mine);
  const ncclResult_t result = run_collective(c, sendbuff, all, stream, [&](int r, const char *theirs) {
    const char *part = theirs + static_cast<size_t>(c->rank) * mine;
    if (r == 0) {
      std::memcpy(acc.data(), part, mine);     // ranks are combined in rank order: deterministic for doubles
      return ncclSuccess;
    }
    return reduce_typed(acc.data(), part, recvcount, datatype, op) ? ncclSuccess : ncclInvalidArgument;
  });
  if (result != ncclSuccess || mine == 0) return result;
  return hipMemcpy(recvbuff, acc.data(), mine, hipMemcpyHostToDevice) == hipSuccess ? ncclSuccess : ncclUnhandledCudaError;
}

ncclResult_t ncclAllReduce(const void *sendbuff, void *recvbuff, size_t count, ncclDataType_t datatype, ncclRedOp_t op, ncclComm_t comm,
                           hipStream_t stream) {
  Comm *c = reinterpret_cast<Comm *>(comm);
  const size_t width = type_bytes(datatype);
  if (c == nullptr || width == 0 || (op != ncclSum && op != ncclMin && op != ncclMax)) return ncclInvalidArgument;
  const size_t bytes = count * width;
  std::vector<char> acc(bytes);
  const ncclResult_t result = run_collective(c, sendbuff, bytes, stream, [&](int r, const char *theirs) {
    if (r == 0) {
      std::memcpy(acc.data(), theirs, bytes);
      return ncclSuccess;
    }
    return reduce_typed(acc.data(), theirs, count, datatype, op) ? ncclSuccess : ncclInvalidArgument;
  });
  if (result != ncclSuccess || bytes == 0) return result;
  return hipMemcpy(recvbuff, acc.data(), bytes, hipMemcpyHostToDevice) == hipSuccess ? ncclSuccess : ncclUnhandledCudaError;
}

const char *ncclGetErrorString(ncclResult_t result) {
  switch (result) {
    case ncclSuccess: return "no error";
    case ncclUnhandledCudaError: return "loopback: a HIP call failed";
    case ncclSystemError: return "loopback: system error (shared memory, or a peer never arrived)";
    case ncclInvalidArgument: return "loopback: invalid argument";
    case ncclInvalidUsage: return "loopback: invalid usage (unsupported call pattern, or ranks disagree on sizes)";
    default: return "loopback: error";
  }
}

}  // extern "C"
