// selftest.cpp — the loopback transport's protocol (rendezvous, grouped send / recv matching, collectives, reductions,
// empty groups) with host memory standing in for device memory: three rank threads, no GPU.  TEST INFRASTRUCTURE.
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <thread>
#include <vector>

#include <hip/hip_runtime_api.h>
#include <rccl/rccl.h>

#define CHECK(cond)                                                                  \
  do {                                                                               \
    if (!(cond)) {                                                                   \
      std::fprintf(stderr, "selftest: %s failed at line %d (rank %d)\n", #cond, __LINE__, rank); \
      failures.fetch_add(1);                                                         \
      return;                                                                        \
    }                                                                                \
  } while (0)

#include <atomic>
static std::atomic<int> failures{0};

static void rank_main(int world, int rank, ncclUniqueId id) {
  ncclComm_t comm;
  CHECK(ncclCommInitRank(&comm, world, id, rank) == ncclSuccess);
  // all-to-all(v): rank r sends (r + 1) * (p + 1) words of value 100 r + p to peer p
  std::vector<std::vector<int64_t>> send(world), recv(world);
  CHECK(ncclGroupStart() == ncclSuccess);
  for (int p = 0; p < world; ++p) {
    send[p].assign(static_cast<size_t>((rank + 1) * (p + 1)), 100 * rank + p);
    recv[p].assign(static_cast<size_t>((p + 1) * (rank + 1)), -1);
    CHECK(ncclSend(send[p].data(), send[p].size(), ncclInt64, p, comm, nullptr) == ncclSuccess);
    CHECK(ncclRecv(recv[p].data(), recv[p].size(), ncclInt64, p, comm, nullptr) == ncclSuccess);
  }
  CHECK(ncclGroupEnd() == ncclSuccess);
  for (int p = 0; p < world; ++p) {
    for (int64_t v : recv[p]) CHECK(v == 100 * p + rank);
  }
  // a round in which only rank 0 -> rank 1 moves anything; rank 2's group is empty
  int64_t word = rank == 0 ? 4242 : 0;
  CHECK(ncclGroupStart() == ncclSuccess);
  if (rank == 0 && world > 1) CHECK(ncclSend(&word, 1, ncclInt64, 1, comm, nullptr) == ncclSuccess);
  if (rank == 1) CHECK(ncclRecv(&word, 1, ncclInt64, 0, comm, nullptr) == ncclSuccess);
  CHECK(ncclGroupEnd() == ncclSuccess);
  if (rank == 1) CHECK(word == 4242);
  // all-gather
  std::vector<uint8_t> mine(5, static_cast<uint8_t>(rank + 1)), all(5 * world, 0);
  CHECK(ncclAllGather(mine.data(), all.data(), 5, ncclUint8, comm, nullptr) == ncclSuccess);
  for (int r = 0; r < world; ++r) {
    for (int i = 0; i < 5; ++i) CHECK(all[static_cast<size_t>(r) * 5 + i] == r + 1);
  }
  // reduce-scatter: sum of doubles, min / max of int64
  const size_t len = 7;
  std::vector<double> fs(len * world), fr(len);
  std::vector<int64_t> is(len * world), ir(len);
  for (size_t i = 0; i < fs.size(); ++i) {
    fs[i] = 0.5 * static_cast<double>(i) + rank;
    is[i] = static_cast<int64_t>(i) * (rank % 2 == 0 ? 1 : -1) + rank;
  }
  CHECK(ncclReduceScatter(fs.data(), fr.data(), len, ncclFloat64, ncclSum, comm, nullptr) == ncclSuccess);
  for (size_t i = 0; i < len; ++i) {
    double want = 0;
    for (int r = 0; r < world; ++r) want += 0.5 * static_cast<double>(rank * len + i) + r;
    CHECK(fr[i] == want);
  }
  for (ncclRedOp_t op : {ncclMin, ncclMax}) {
    CHECK(ncclReduceScatter(is.data(), ir.data(), len, ncclInt64, op, comm, nullptr) == ncclSuccess);
    for (size_t i = 0; i < len; ++i) {
      int64_t want = 0;
      for (int r = 0; r < world; ++r) {
        const int64_t v = static_cast<int64_t>(rank * len + i) * (r % 2 == 0 ? 1 : -1) + r;
        want = r == 0 ? v : (op == ncclMin ? (v < want ? v : want) : (v > want ? v : want));
      }
      CHECK(ir[i] == want);
    }
  }
  // all-reduce
  int64_t one = rank + 1, total = 0;
  CHECK(ncclAllReduce(&one, &total, 1, ncclInt64, ncclSum, comm, nullptr) == ncclSuccess);
  CHECK(total == world * (world + 1) / 2);
  // ranks that disagree on a size are told so instead of reading garbage
  CHECK(ncclGroupStart() == ncclSuccess);
  int64_t two[2] = {1, 2};
  if (rank == 0 && world > 1) CHECK(ncclSend(two, 2, ncclInt64, 1, comm, nullptr) == ncclSuccess);
  if (rank == 1) CHECK(ncclRecv(two, 1, ncclInt64, 0, comm, nullptr) == ncclSuccess);
  const ncclResult_t end = ncclGroupEnd();
  if (rank == 1) CHECK(end == ncclInvalidUsage);
  CHECK(ncclCommDestroy(comm) == ncclSuccess);
}

int main() {
  for (int world : {1, 2, 3}) {
    ncclUniqueId id;
    if (ncclGetUniqueId(&id) != ncclSuccess) return 2;
    std::vector<std::thread> ranks;
    for (int r = 0; r < world; ++r) ranks.emplace_back(rank_main, world, r, id);
    for (auto &t : ranks) t.join();
  }
  if (failures.load() != 0) return 1;
  std::printf("loopback selftest ok\n");
  return 0;
}
