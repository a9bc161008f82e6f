// What a scratch allocation costs a work order: hipMalloc / hipFree against the stream-ordered pool (hipMallocAsync /
// hipFreeAsync with the release threshold raised so freed memory stays in the pool).
// build: hipcc --offload-arch=gfx950 -O2 -o alloc_cost alloc_cost.hip
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdint>
#include <cstdio>

static double now_us() {
  return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

int main() {
  hipStream_t s;
  hipStreamCreate(&s);
  hipMemPool_t pool;
  hipDeviceGetDefaultMemPool(&pool, 0);
  uint64_t keep = UINT64_MAX;
  hipMemPoolSetAttribute(pool, hipMemPoolAttrReleaseThreshold, &keep);
  for (size_t bytes : {size_t(8), size_t(64) << 10, size_t(4) << 20, size_t(256) << 20}) {
    const int reps = 200;
    void *p = nullptr;
    hipMalloc(&p, bytes); hipFree(p);
    double t0 = now_us();
    for (int i = 0; i < reps; ++i) { hipMalloc(&p, bytes); hipFree(p); }
    const double sync_us = (now_us() - t0) / reps;
    hipMallocAsync(&p, bytes, s); hipFreeAsync(p, s); hipStreamSynchronize(s);
    t0 = now_us();
    for (int i = 0; i < reps; ++i) { hipMallocAsync(&p, bytes, s); hipFreeAsync(p, s); }
    hipStreamSynchronize(s);
    const double async_us = (now_us() - t0) / reps;
    std::printf("{\"bytes\": %zu, \"hipMalloc+hipFree_us\": %.1f, \"hipMallocAsync+hipFreeAsync_us\": %.1f}\n", bytes, sync_us, async_us);
  }
  return 0;
}
