// comm.hip — the exchange steps of the multi-GPU path behind the C ABI (include/qsx.h "multi-GPU"): one process per GPU,
// RCCL over xGMI.  The reference has no data-plane collective — its partitions share one address space
// (storage/InsertDestination.hpp:490-660 routes tuples, BuildHashOperator.cpp:82-91 / HashJoinOperator.cpp:220-231 make
// per-partition work orders) — so these entry points are this repo's design: what a C++ host needs to turn "partition p" into
// "GPU p" (quickstep_amd/distributed.py does the same through torch.distributed for the Python callers).
#include "comm.hpp"

#include <dlfcn.h>

#include <chrono>
#include <cstdlib>
#include <mutex>
#include <thread>
#include <string>
#include <vector>

namespace qsx {

void set_last_error(const char *what, hipError_t err);

const RcclApi *rccl() {
  static RcclApi api;
  static bool ok = false;
  static std::once_flag once;
  std::call_once(once, []() {
    // QSX_RCCL_LIBRARY names the library to bind instead — a TEST HOOK (the tests' loopback transport,
    // tests/cpp/loopback/loopback_rccl.cpp, which lets several ranks share one GPU).  An environment variable must not be
    // able to put another library under a production process's collectives by itself: it is honoured only together with
    // QSX_ALLOW_TEST_TRANSPORT=1, and without that switch a process that carries it gets NO transport at all (every
    // multi-GPU entry point fails with QSX_ERR_COMM) rather than a silent fall-back to the real RCCL.
    void *lib = nullptr;
    const char *named = std::getenv("QSX_RCCL_LIBRARY");
    if (named != nullptr && named[0] != 0) {
      const char *allow = std::getenv("QSX_ALLOW_TEST_TRANSPORT");
      if (allow == nullptr || std::string(allow) != "1") {
        set_last_error_text("QSX_RCCL_LIBRARY is set but QSX_ALLOW_TEST_TRANSPORT=1 is not: refusing to bind a substitute for RCCL");
        return;
      }
      lib = dlopen(named, RTLD_NOW | RTLD_LOCAL);
      if (lib == nullptr) {
        set_last_error_text((std::string("QSX_RCCL_LIBRARY: ") + dlerror()).c_str());
        return;
      }
    } else {
      lib = dlopen("librccl.so.1", RTLD_NOW | RTLD_NOLOAD | RTLD_GLOBAL);   // the copy the process already holds, if any
      if (lib == nullptr) lib = dlopen("librccl.so.1", RTLD_NOW | RTLD_LOCAL);
      if (lib == nullptr) lib = dlopen("/opt/rocm/lib/librccl.so.1", RTLD_NOW | RTLD_LOCAL);
    }
    if (lib == nullptr) {
      set_last_error_text("librccl.so.1 not found (multi-GPU entry points need RCCL)");
      return;
    }
    bool all = true;
    auto bind = [&](auto &slot, const char *name) {
      slot = reinterpret_cast<std::remove_reference_t<decltype(slot)>>(dlsym(lib, name));
      all = all && slot != nullptr;
    };
    bind(api.GetUniqueId, "ncclGetUniqueId");
    bind(api.CommInitRank, "ncclCommInitRank");
    bind(api.CommDestroy, "ncclCommDestroy");
    bind(api.GroupStart, "ncclGroupStart");
    bind(api.GroupEnd, "ncclGroupEnd");
    bind(api.Send, "ncclSend");
    bind(api.Recv, "ncclRecv");
    bind(api.AllGather, "ncclAllGather");
    bind(api.ReduceScatter, "ncclReduceScatter");
    bind(api.AllReduce, "ncclAllReduce");
    bind(api.GetErrorString, "ncclGetErrorString");
    const bool required = all;
    bind(api.CommAbort, "ncclCommAbort");          // optional
    all = required;
    if (!all) set_last_error_text("librccl.so.1 lacks a symbol the multi-GPU entry points need");
    ok = all;
  });
  return ok ? &api : nullptr;
}

int rccl_status(ncclResult_t r, const char *what) {
  if (r == ncclSuccess) return QSX_OK;
  const RcclApi *api = rccl();
  std::string text = std::string(what) + ": " + (api != nullptr ? api->GetErrorString(r) : "RCCL error");
  set_last_error_text(text.c_str());
  return QSX_ERR_COMM;
}

// hipStreamSynchronize with a deadline: a collective whose peers never arrive must not hold the caller for ever.
int comm_wait(::qsx_comm *c, hipStream_t s) {
  if (c->aborted.load(std::memory_order_acquire)) {
    set_last_error_text("the communicator was aborted");
    return QSX_ERR_COMM;
  }
  if (c->world == 1 || c->timeout_ms <= 0) {
    QSX_HIP_TRY(hipStreamSynchronize(s));
    return QSX_OK;
  }
  const auto deadline = std::chrono::steady_clock::now() + std::chrono::milliseconds(c->timeout_ms);
  int spins = 0;
  for (;;) {
    const hipError_t q = hipStreamQuery(s);
    if (q == hipSuccess) return QSX_OK;
    if (q != hipErrorNotReady) {
      set_last_error("hipStreamQuery", q);
      return QSX_ERR_HIP;
    }
    if (std::chrono::steady_clock::now() > deadline) {
      set_last_error_text("a collective did not finish before QSX_COMM_TIMEOUT_MS: a peer rank is not issuing the same calls; communicator aborted");
      (void)qsx_comm_abort(c);
      return QSX_ERR_COMM;
    }
    if (++spins < 2000) std::this_thread::yield(); else std::this_thread::sleep_for(std::chrono::microseconds(200));
  }
}

int comm_agree(::qsx_comm *c, int local_status, hipStream_t s) {
  if (c->aborted.load(std::memory_order_acquire)) {
    set_last_error_text("the communicator was aborted");
    return local_status != QSX_OK ? local_status : QSX_ERR_COMM;
  }
  if (c->world == 1) return local_status;
  const RcclApi *api = rccl();
  if (api == nullptr) return QSX_ERR_COMM;
  const int world = c->world;
  c->status_host[world] = local_status;
  int rc = QSX_OK;
  if (hipMemcpyAsync(c->status_dev + world, c->status_host + world, 8, hipMemcpyHostToDevice, s) != hipSuccess) rc = QSX_ERR_HIP;
  // (a rank that cannot even stage its word still enters the collective: its peers are waiting there)
  const int gathered = rccl_status(api->AllGather(c->status_dev + world, c->status_dev, 1, ncclInt64, c->comm, s), "ncclAllGather(status)");
  if (gathered != QSX_OK) return local_status != QSX_OK ? local_status : gathered;
  if (hipMemcpyAsync(c->status_host, c->status_dev, 8 * static_cast<size_t>(world), hipMemcpyDeviceToHost, s) != hipSuccess) rc = QSX_ERR_HIP;
  const int waited = comm_wait(c, s);
  if (waited != QSX_OK) return local_status != QSX_OK ? local_status : waited;
  if (local_status != QSX_OK) return local_status;
  if (rc != QSX_OK) return rc;
  for (int r = 0; r < world; ++r) {
    if (c->status_host[r] != QSX_OK) {
      const std::string text = "rank " + std::to_string(r) + " failed before a collective step (its status " + std::to_string(c->status_host[r]) +
                               "): every rank gives the step up";
      set_last_error_text(text.c_str());
      return QSX_ERR_COMM;
    }
  }
  return QSX_OK;
}

}  // namespace qsx

using namespace qsx;

extern "C" {

int qsx_comm_unique_id(void *out_id) {
  QSX_REQUIRE_DEVICE();
  if (out_id == nullptr) return QSX_ERR_INVALID_ARGUMENT;
  const RcclApi *api = rccl();
  if (api == nullptr) return QSX_ERR_COMM;
  static_assert(sizeof(ncclUniqueId) == QSX_COMM_ID_BYTES, "QSX_COMM_ID_BYTES mirrors NCCL_UNIQUE_ID_BYTES");
  ncclUniqueId id;
  QSX_RCCL_TRY(api->GetUniqueId(&id), "ncclGetUniqueId");
  std::memcpy(out_id, &id, sizeof(id));
  return QSX_OK;
}

int qsx_comm_create(int world, int rank, const void *id_bytes, qsx_comm_t **out) {
  QSX_REQUIRE_DEVICE();
  if (out == nullptr || id_bytes == nullptr || world < 1 || rank < 0 || rank >= world) return QSX_ERR_INVALID_ARGUMENT;
  const RcclApi *api = rccl();
  if (api == nullptr) return QSX_ERR_COMM;
  ncclUniqueId id;
  std::memcpy(&id, id_bytes, sizeof(id));
  qsx_comm *c = new qsx_comm;
  c->world = world;
  c->rank = rank;
  if (const char *t = std::getenv("QSX_COMM_TIMEOUT_MS")) c->timeout_ms = std::atoll(t);
  // The watchdog's promise — no wait on this communicator outlasts the deadline — needs ncclCommAbort to take the stalled
  // collective's kernels off the stream: the waits behind comm_wait (the scratch's and the pooled buffers' stream
  // synchronisations) would block on them otherwise.  A transport without it gets NO watchdog (plain waits, as with
  // QSX_COMM_TIMEOUT_MS=0) instead of a deadline that only moves the stall into the next wait.
  if (api->CommAbort == nullptr) c->timeout_ms = 0;
  if (hipMalloc(reinterpret_cast<void **>(&c->status_dev), 8 * static_cast<size_t>(world + 1)) != hipSuccess ||
      hipHostMalloc(reinterpret_cast<void **>(&c->status_host), 8 * static_cast<size_t>(world + 1), hipHostMallocDefault) != hipSuccess) {
    if (c->status_dev != nullptr) (void)hipFree(c->status_dev);
    delete c;
    return QSX_ERR_OUT_OF_MEMORY;
  }
  const int rc = rccl_status(api->CommInitRank(&c->comm, world, id, rank), "ncclCommInitRank");
  if (rc != QSX_OK) {
    (void)hipFree(c->status_dev);
    (void)hipHostFree(c->status_host);
    delete c;
    return rc;
  }
  *out = c;
  return QSX_OK;
}

int qsx_comm_destroy(qsx_comm_t *c) {
  if (c == nullptr) return QSX_OK;
  const RcclApi *api = rccl();
  // (an aborted communicator was already given back to RCCL by ncclCommAbort)
  if (api != nullptr && c->comm != nullptr && !(c->aborted.load() && api->CommAbort != nullptr)) (void)api->CommDestroy(c->comm);
  if (c->status_dev != nullptr) (void)hipFree(c->status_dev);
  if (c->status_host != nullptr) (void)hipHostFree(c->status_host);
  delete c;
  return QSX_OK;
}

int qsx_comm_abort(qsx_comm_t *c) {
  if (c == nullptr) return QSX_ERR_INVALID_ARGUMENT;
  if (c->aborted.exchange(true)) return QSX_OK;
  const RcclApi *api = rccl();
  if (api != nullptr && api->CommAbort != nullptr && c->comm != nullptr) (void)api->CommAbort(c->comm);
  return QSX_OK;
}

int qsx_comm_agree(qsx_comm_t *c, int local_status, qsx_stream_t stream) {
  QSX_REQUIRE_DEVICE();
  if (c == nullptr) return QSX_ERR_INVALID_ARGUMENT;
  return comm_agree(c, local_status, as_stream(stream));
}

int qsx_comm_synchronize(qsx_comm_t *c, qsx_stream_t stream) {
  QSX_REQUIRE_DEVICE();
  if (c == nullptr) return QSX_ERR_INVALID_ARGUMENT;
  return comm_wait(c, as_stream(stream));
}

int qsx_comm_rank(const qsx_comm_t *c, int *out_world, int *out_rank) {
  if (c == nullptr) return QSX_ERR_INVALID_ARGUMENT;
  if (out_world != nullptr) *out_world = c->world;
  if (out_rank != nullptr) *out_rank = c->rank;
  return QSX_OK;
}

int qsx_exchange_counts(qsx_comm_t *c, const int64_t *send_counts_dev, int64_t *recv_counts_dev, qsx_stream_t stream) {
  QSX_REQUIRE_DEVICE();
  if (c == nullptr || send_counts_dev == nullptr || recv_counts_dev == nullptr) return QSX_ERR_INVALID_ARGUMENT;
  const RcclApi *api = rccl();
  if (api == nullptr) return QSX_ERR_COMM;
  hipStream_t s = as_stream(stream);
  // a rank's word for itself never meets the transport (a send to oneself is a copy kernel of RCCL's behind its launch
  // machinery: 0.26-0.34 ms per call at world 1 under the tracer); the stream orders the copy like it would the collective
  QSX_HIP_TRY(hipMemcpyAsync(recv_counts_dev + c->rank, send_counts_dev + c->rank, sizeof(int64_t), hipMemcpyDeviceToDevice, s));
  if (c->world == 1) return QSX_OK;
  RcclGroup group(api);
  for (int p = 0; p < c->world && group.ok(); ++p) {
    if (p == c->rank) continue;
    group.add(api->Send(send_counts_dev + p, 1, ncclInt64, p, c->comm, s), "ncclSend");
    group.add(api->Recv(recv_counts_dev + p, 1, ncclInt64, p, c->comm, s), "ncclRecv");
  }
  return group.end();
}

int qsx_alltoallv(qsx_comm_t *c, int width, const void *send_dev, const int64_t *send_rows, void *recv_dev, const int64_t *recv_rows,
                  qsx_stream_t stream) {
  QSX_REQUIRE_DEVICE();
  if (c == nullptr || width < 1 || send_rows == nullptr || recv_rows == nullptr) return QSX_ERR_INVALID_ARGUMENT;
  const RcclApi *api = rccl();
  if (api == nullptr) return QSX_ERR_COMM;
  hipStream_t s = as_stream(stream);
  const char *send = static_cast<const char *>(send_dev);
  char *recv = static_cast<char *>(recv_dev);
  int64_t send_at = 0, recv_at = 0;
  for (int p = 0; p < c->world; ++p) {
    if (send_rows[p] < 0 || recv_rows[p] < 0) return QSX_ERR_INVALID_ARGUMENT;
    if ((send_rows[p] > 0 && send_dev == nullptr) || (recv_rows[p] > 0 && recv_dev == nullptr)) return QSX_ERR_INVALID_ARGUMENT;
  }
  if (send_rows[c->rank] != recv_rows[c->rank]) return QSX_ERR_INVALID_ARGUMENT;   // (what a rank sends itself is what it receives from itself)
  RcclGroup group(c->world > 1 ? api : nullptr);   // (every rank of a group of several opens and ends the batch, also an empty one)
  for (int p = 0; p < c->world && group.ok(); ++p) {
    if (p == c->rank) {   // the own piece: a copy on the stream, not a trip through the transport
      if (send_rows[p] > 0) {
        QSX_HIP_TRY(hipMemcpyAsync(recv + recv_at * width, send + send_at * width, static_cast<size_t>(send_rows[p]) * width, hipMemcpyDeviceToDevice, s));
      }
    } else {
      if (send_rows[p] > 0) group.add(api->Send(send + send_at * width, static_cast<size_t>(send_rows[p]) * width, ncclUint8, p, c->comm, s), "ncclSend");
      if (recv_rows[p] > 0) group.add(api->Recv(recv + recv_at * width, static_cast<size_t>(recv_rows[p]) * width, ncclUint8, p, c->comm, s), "ncclRecv");
    }
    send_at += send_rows[p];
    recv_at += recv_rows[p];
  }
  return group.end();
}

int qsx_allgather(qsx_comm_t *c, const void *send_dev, size_t bytes, void *recv_dev, qsx_stream_t stream) {
  QSX_REQUIRE_DEVICE();
  if (c == nullptr || (bytes > 0 && (send_dev == nullptr || recv_dev == nullptr))) return QSX_ERR_INVALID_ARGUMENT;
  if (bytes == 0) return QSX_OK;
  const RcclApi *api = rccl();
  if (api == nullptr) return QSX_ERR_COMM;
  if (c->world == 1) {   // (one rank: its part is the whole)
    if (recv_dev != send_dev) QSX_HIP_TRY(hipMemcpyAsync(recv_dev, send_dev, bytes, hipMemcpyDeviceToDevice, as_stream(stream)));
    return QSX_OK;
  }
  QSX_RCCL_TRY(api->AllGather(send_dev, recv_dev, bytes, ncclUint8, c->comm, as_stream(stream)), "ncclAllGather");
  return QSX_OK;
}

int qsx_bitmap_allreduce_or(qsx_comm_t *c, uint64_t *words_dev, int64_t num_words, qsx_stream_t stream) {
  QSX_REQUIRE_DEVICE();
  if (c == nullptr || num_words < 0 || (num_words > 0 && words_dev == nullptr)) return QSX_ERR_INVALID_ARGUMENT;
  if (num_words == 0 || c->world == 1) return QSX_OK;
  const RcclApi *api = rccl();
  if (api == nullptr) return QSX_ERR_COMM;
  // RCCL reduces with SUM / PROD / MIN / MAX / AVG only: the words are gathered and OR-ed here
  hipStream_t s = as_stream(stream);
  CallScratch scratch(s);
  const size_t bytes = static_cast<size_t>(num_words) * 8 * c->world;
  // (the scratch is this rank's own business: agree on it before the all-gather, or a rank without memory leaves the
  // others waiting in it)
  const int rc = comm_agree(c, scratch.reserve(CallScratch::padded(bytes)), s);
  if (rc != QSX_OK) return rc;
  unsigned long long *all = static_cast<unsigned long long *>(scratch.take(bytes));
  QSX_RCCL_TRY(api->AllGather(words_dev, all, static_cast<size_t>(num_words), ncclUint64, c->comm, s), "ncclAllGather");
  hipLaunchKernelGGL(or_words_kernel, dim3(grid_for(num_words, 256)), dim3(256), 0, s, all, c->world, static_cast<long long>(num_words), ~0ull, ~0ull,
                     reinterpret_cast<unsigned long long *>(words_dev));
  QSX_CHECK_LAUNCH();
  return QSX_OK;
}

}  // extern "C"
