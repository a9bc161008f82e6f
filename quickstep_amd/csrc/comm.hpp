// comm.hpp — RCCL, bound at run time (dlopen of librccl.so.1): libqsx.so carries no link-time dependency on it, a process
// that never shards never loads it, and inside a process that already holds a copy (PyTorch ships one under the same
// SONAME) the loader hands back that copy instead of a second one.
#ifndef QSX_CSRC_COMM_HPP_
#define QSX_CSRC_COMM_HPP_

#include <rccl/rccl.h>

#include <atomic>

#include "common.hpp"

struct qsx_comm;

namespace qsx {

struct RcclApi {
  decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
  decltype(&ncclCommInitRank) CommInitRank = nullptr;
  decltype(&ncclCommDestroy) CommDestroy = nullptr;
  decltype(&ncclGroupStart) GroupStart = nullptr;
  decltype(&ncclGroupEnd) GroupEnd = nullptr;
  decltype(&ncclSend) Send = nullptr;
  decltype(&ncclRecv) Recv = nullptr;
  decltype(&ncclAllGather) AllGather = nullptr;
  decltype(&ncclReduceScatter) ReduceScatter = nullptr;
  decltype(&ncclAllReduce) AllReduce = nullptr;
  decltype(&ncclGetErrorString) GetErrorString = nullptr;
  decltype(&ncclCommAbort) CommAbort = nullptr;   // optional: absent from a transport that cannot abort
};
// nullptr (and qsx_last_error set) when the library or one of the symbols is missing.
const RcclApi *rccl();
int rccl_status(ncclResult_t r, const char *what);   // QSX_OK or QSX_ERR_COMM with qsx_last_error set
void set_last_error_text(const char *text);
// The failure agreement of qsx_comm_agree for callers inside the library (csrc/comm.hip).
int comm_agree(::qsx_comm *c, int local_status, hipStream_t s);
// hipStreamSynchronize under the communicator's watchdog (qsx_comm_synchronize).
int comm_wait(::qsx_comm *c, hipStream_t s);

// out[w] = OR over r of parts[r * words + w], AND mask of the first / last word (bits of the neighbouring key ranges that
// share a boundary word are dropped)
static __global__ __launch_bounds__(256) void or_words_kernel(const unsigned long long *__restrict__ parts, int num_parts, long long words,
                                                       unsigned long long first_mask, unsigned long long last_mask,
                                                       unsigned long long *__restrict__ out) {
  for (long long w = static_cast<long long>(blockIdx.x) * 256 + threadIdx.x; w < words; w += static_cast<long long>(gridDim.x) * 256) {
    unsigned long long v = 0;
    for (int r = 0; r < num_parts; ++r) v |= parts[static_cast<long long>(r) * words + w];
    if (w == 0) v &= first_mask;
    if (w == words - 1) v &= last_mask;
    out[w] = v;
  }
}

}  // namespace qsx

struct qsx_comm {
  ncclComm_t comm = nullptr;
  int world = 1;
  int rank = 0;
  // the agreement's buffers, allocated once with the communicator: [world] gathered words + 1 contributed word on the
  // device, the same in pinned host memory — a status exchange never allocates
  long long *status_dev = nullptr;
  long long *status_host = nullptr;
  std::atomic<bool> aborted{false};
  long long timeout_ms = 600000;
};

namespace qsx {
// ncclGroupStart ... ncclGroupEnd around a batch of sends / receives.  A call that fails inside the batch must not leave
// the thread's group open (every later collective would queue into a group that never ends): the first error is kept,
// nothing more is issued, and ncclGroupEnd always runs — from end() or, on an early return, from the destructor.
class RcclGroup {
 public:
  // api == nullptr: a batch that turned out to hold nothing for the transport (every piece was the rank's own): no group
  explicit RcclGroup(const RcclApi *api) : api_(api) {
    if (api_ == nullptr) return;
    status_ = rccl_status(api_->GroupStart(), "ncclGroupStart");
    open_ = status_ == QSX_OK;
  }
  ~RcclGroup() {
    if (open_) (void)api_->GroupEnd();
  }
  RcclGroup(const RcclGroup &) = delete;
  RcclGroup &operator=(const RcclGroup &) = delete;
  bool ok() const { return status_ == QSX_OK; }
  void add(ncclResult_t r, const char *what) {
    if (status_ == QSX_OK) status_ = rccl_status(r, what);
  }
  int end() {
    if (open_) {
      open_ = false;
      const int rc = rccl_status(api_->GroupEnd(), "ncclGroupEnd");
      if (status_ == QSX_OK) status_ = rc;
    }
    return status_;
  }

 private:
  const RcclApi *api_;
  int status_ = QSX_OK;
  bool open_ = false;
};
}  // namespace qsx

#define QSX_RCCL_TRY(call, what)                              \
  do {                                                         \
    const int rc_rccl__ = qsx::rccl_status((call), what);      \
    if (rc_rccl__ != QSX_OK) return rc_rccl__;                 \
  } while (0)

#endif  // QSX_CSRC_COMM_HPP_
