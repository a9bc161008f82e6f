// Stress of the stream-ordered allocator the way the library uses it: T host threads, each with its own stream, loop
// { hipMallocAsync; kernel writes a per-iteration pattern; kernel verifies it; hipFreeAsync }.  Any cross-thread reuse of a
// block before its last reader finished shows up as a mismatch.  build + run: hipcc -O2 --offload-arch=gfx950 -o
// /tmp/alloc_stress tools/ubench/alloc_stress.hip -lpthread && /tmp/alloc_stress
#include <hip/hip_runtime.h>

#include <atomic>
#include <cstdio>
#include <thread>
#include <vector>

__global__ void fill(unsigned *p, size_t n, unsigned tag) {
  for (size_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += static_cast<size_t>(gridDim.x) * blockDim.x) p[i] = tag ^ static_cast<unsigned>(i);
}
__global__ void spin_then_check(const unsigned *p, size_t n, unsigned tag, unsigned *errors, int spin) {
  for (int s = 0; s < spin; ++s) __builtin_amdgcn_s_sleep(64);
  for (size_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += static_cast<size_t>(gridDim.x) * blockDim.x) {
    if (p[i] != (tag ^ static_cast<unsigned>(i))) atomicAdd(errors, 1u);
  }
}

int main(int argc, char **argv) {
  const int threads = argc > 1 ? atoi(argv[1]) : 4, iters = argc > 2 ? atoi(argv[2]) : 2000;
  unsigned *errors = nullptr;
  hipMalloc(&errors, 4);
  hipMemset(errors, 0, 4);
  std::atomic<int> api_failures{0};
  std::vector<std::thread> pool;
  for (int t = 0; t < threads; ++t) {
    pool.emplace_back([&, t]() {
      hipStream_t s;
      hipStreamCreate(&s);
      for (int it = 0; it < iters; ++it) {
        const size_t n = 1024 + ((it * 7919 + t * 104729) % 65536);
        unsigned *p = nullptr;
        if (hipMallocAsync(reinterpret_cast<void **>(&p), n * 4, s) != hipSuccess) { ++api_failures; continue; }
        const unsigned tag = static_cast<unsigned>(t * 1000003 + it);
        hipLaunchKernelGGL(fill, dim3(8), dim3(256), 0, s, p, n, tag);
        hipLaunchKernelGGL(spin_then_check, dim3(8), dim3(256), 0, s, p, n, tag, errors, 50);
        if (hipFreeAsync(p, s) != hipSuccess) ++api_failures;
        if (it % 64 == 63) hipStreamSynchronize(s);
      }
      hipStreamSynchronize(s);
      hipStreamDestroy(s);
    });
  }
  for (auto &th : pool) th.join();
  unsigned host_errors = 0;
  hipMemcpy(&host_errors, errors, 4, hipMemcpyDeviceToHost);
  std::printf("threads=%d iters=%d api_failures=%d mismatches=%u last_error=%s\n", threads, iters, api_failures.load(), host_errors,
              hipGetErrorString(hipGetLastError()));
  return host_errors != 0 || api_failures.load() != 0;
}
