// The pattern of a host-layer work order on pooled scratch: T threads, each with its own NON-BLOCKING stream, loop
// { hipMallocAsync(bytes); hipMemsetAsync 0; kernel adds a per-iteration tag; hipMemcpyAsync to pageable host memory;
//   hipStreamSynchronize; check; hipFreeAsync } while other threads hipMalloc / hipFree plain buffers.
// usage: pool_readback [threads] [iters] [bytes] [pooled = 1 | 0: hipMalloc / hipFree] [plain allocations next to it = 1 | 0] [what: 0 memset+atomic, 1 plain store kernel] [keep = 1: release threshold of the pool raised to everything]
#include <hip/hip_runtime.h>

#include <atomic>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <thread>
#include <vector>

__global__ void add_tag(unsigned long long *p, unsigned long long tag) {
  if (threadIdx.x == 0 && blockIdx.x == 0) atomicAdd(p, tag);
}

__global__ void store_tag(unsigned long long *p, unsigned long long tag) {
  if (threadIdx.x == 0) *p = tag;
}

int main(int argc, char **argv) {
  const int threads = argc > 1 ? atoi(argv[1]) : 4, iters = argc > 2 ? atoi(argv[2]) : 5000;
  const size_t bytes = argc > 3 ? static_cast<size_t>(atoll(argv[3])) : 8;
  const bool pooled = argc > 4 ? atoi(argv[4]) != 0 : true;
  const bool neighbours = argc > 5 ? atoi(argv[5]) != 0 : true;
  const int what = argc > 6 ? atoi(argv[6]) : 0;
  if (argc > 7 && atoi(argv[7]) != 0) {
    hipMemPool_t mp;
    hipDeviceGetDefaultMemPool(&mp, 0);
    uint64_t keep = UINT64_MAX;
    hipMemPoolSetAttribute(mp, hipMemPoolAttrReleaseThreshold, &keep);
  }
  std::atomic<int> mismatches{0}, api_failures{0};
  std::vector<std::thread> pool;
  for (int t = 0; t < threads; ++t) {
    pool.emplace_back([&, t]() {
      hipStream_t s;
      hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
      for (int it = 0; it < iters; ++it) {
        unsigned long long *p = nullptr;
        if ((pooled ? hipMallocAsync(reinterpret_cast<void **>(&p), bytes, s) : hipMalloc(reinterpret_cast<void **>(&p), bytes)) != hipSuccess) { ++api_failures; continue; }
        const unsigned long long tag = static_cast<unsigned long long>(t) * 1000003ull + it + 1;
        if (what == 0) {
          hipMemsetAsync(p, 0, 8, s);
          hipLaunchKernelGGL(add_tag, dim3(4), dim3(64), 0, s, p, tag);
        } else {
          hipLaunchKernelGGL(store_tag, dim3(1), dim3(64), 0, s, p, tag);
        }
        unsigned long long v = ~0ull;
        hipMemcpyAsync(&v, p, 8, hipMemcpyDeviceToHost, s);
        hipStreamSynchronize(s);
        if (v != tag) {
          ++mismatches;
          unsigned long long again = 0;
          hipMemcpy(&again, p, 8, hipMemcpyDeviceToHost);
          std::fprintf(stderr, "thread %d iteration %d: read %llx, expected %llx, second read %llx, p = %p\n", t, it, v, tag, again, static_cast<void *>(p));
        }
        if ((pooled ? hipFreeAsync(p, s) : hipFree(p)) != hipSuccess) ++api_failures;
        if (neighbours && t == 0 && it % 16 == 0) {          // plain allocations next to it, as storage blocks are
          void *q = nullptr;
          hipMalloc(&q, 300000);
          hipFree(q);
        }
      }
      hipStreamSynchronize(s);
      hipStreamDestroy(s);
    });
  }
  for (auto &th : pool) th.join();
  std::printf("{\"threads\": %d, \"iters\": %d, \"bytes\": %zu, \"pooled\": %d, \"api_failures\": %d, \"mismatches\": %d}\n", threads, iters, bytes, pooled ? 1 : 0,
              api_failures.load(), mismatches.load());
  return mismatches.load() != 0;
}
