// Microbenchmark for "Q1 over code stripes: code histograms instead of f64 atomics" (VERDICT r04 item 2): what does a row cost
// in LDS atomics under each layout?  Rows are synthetic (group by Q1's probabilities, uniform quantity / discount / tax codes).
//   V0  today's layout: 5 SUM f64 + 1 COUNT u64 atomics per row into lane-private bank columns (16 slots x 64 lanes)
//   V1  3 atomics per row: SUM(price) f64 and COUNT u32 into the cell (group, disc, tax) [99 cells per group], quantity histogram
//       u32 [group][50]; C copies of every plane, copy = lane % C (C = 1, 2, 4, 8)
//   V2  4 atomics: 3 f64 sums into lane-private columns + ONE u32 into the joint histogram (group, qty, disc) [550 bins per group]
// build: hipcc --offload-arch=gfx950 -O3 -munsafe-fp-atomics tools/ubench/lds_hist.hip -o /tmp/lds_hist
#include <hip/hip_runtime.h>
#include <cstdio>

__device__ __forceinline__ unsigned lcg(unsigned &s) { s = s * 1664525u + 1013904223u; return s >> 8; }
__device__ __forceinline__ int q1_group(unsigned r) {   // (A,F) .2466, (N,F) .0065, (N,O) .5005, (R,F) .2464 of 2^24
  return r < 4137000u ? 0 : (r < 4246000u ? 1 : (r < 12643000u ? 2 : 3));
}

template <int V, int C>
__global__ __launch_bounds__(256) void k(int iters, double *out) {
  extern __shared__ unsigned long long lds[];
  const int lane = threadIdx.x & 63;
  const int words = V == 0 ? 6 * 16 * 64 : (V == 1 ? C * (16 * 99 + (16 * 99 + 1) / 2 + (16 * 50 + 1) / 2) : 3 * 16 * 64 + C * (16 * 550 + 1) / 2);
  for (int i = threadIdx.x; i < words; i += 256) lds[i] = 0;
  __syncthreads();
  unsigned s = (threadIdx.x + blockIdx.x * 256) * 2654435761u;
  double *f = reinterpret_cast<double *>(lds);
  for (int i = 0; i < iters; ++i) {
    const int g = q1_group(lcg(s) & 0xFFFFFFu);
    const unsigned r = lcg(s);
    const int q = r % 50, d = (r >> 8) % 11, t = (r >> 16) % 9;
    const double price = 900.0 + (r & 0xFFFF);
    if (V == 0) {
      unsafeAtomicAdd(&f[(0 * 16 + g) * 64 + lane], double(q + 1));
      unsafeAtomicAdd(&f[(1 * 16 + g) * 64 + lane], price);
      unsafeAtomicAdd(&f[(2 * 16 + g) * 64 + lane], price * (1.0 - d * 0.01));
      unsafeAtomicAdd(&f[(3 * 16 + g) * 64 + lane], price * (1.0 - d * 0.01) * (1.0 + t * 0.01));
      unsafeAtomicAdd(&f[(4 * 16 + g) * 64 + lane], d * 0.01);
      atomicAdd(&lds[(5 * 16 + g) * 64 + lane], 1ull);
    } else if (V == 1) {
      const int copy = lane & (C - 1);
      const int cell = (g * 11 + d) * 9 + t;
      double *sum = f + copy * (16 * 99);
      unsigned *cnt = reinterpret_cast<unsigned *>(f + C * (16 * 99)) + copy * (16 * 99);
      unsigned *hq = reinterpret_cast<unsigned *>(f + C * (16 * 99) + C * ((16 * 99 + 1) / 2)) + copy * (16 * 50);
      unsafeAtomicAdd(&sum[cell], price);
      atomicAdd(&cnt[cell], 1u);
      atomicAdd(&hq[g * 50 + q], 1u);
    } else {
      const int copy = lane & (C - 1);
      unsafeAtomicAdd(&f[(0 * 16 + g) * 64 + lane], price);
      unsafeAtomicAdd(&f[(1 * 16 + g) * 64 + lane], price * (1.0 - d * 0.01));
      unsafeAtomicAdd(&f[(2 * 16 + g) * 64 + lane], price * (1.0 - d * 0.01) * (1.0 + t * 0.01));
      unsigned *h = reinterpret_cast<unsigned *>(f + 3 * 16 * 64) + copy * (16 * 550);
      atomicAdd(&h[(g * 50 + q) * 11 + d], 1u);
    }
  }
  __syncthreads();
  if (threadIdx.x < 64) out[blockIdx.x * 64 + threadIdx.x] = f[threadIdx.x] + f[words - 1 - threadIdx.x];
}

template <int V, int C>
void run(const char *name, int blocks_per_cu) {
  double *out;
  hipMalloc(&out, 256 * 8 * 64 * 8);
  const int iters = 20000;
  const size_t words = V == 0 ? 6 * 16 * 64 : (V == 1 ? C * (16 * 99 + (16 * 99 + 1) / 2 + (16 * 50 + 1) / 2) : 3 * 16 * 64 + C * (16 * 550 + 1) / 2);
  const size_t bytes = words * 8;
  if (bytes * blocks_per_cu > 150 * 1024) { printf("%-44s blocks/CU=%d  does not fit\n", name, blocks_per_cu); return; }
  hipFuncSetAttribute(reinterpret_cast<const void *>(k<V, C>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  hipEvent_t a, b;
  hipEventCreate(&a);
  hipEventCreate(&b);
  const int grid = 256 * blocks_per_cu;
  hipLaunchKernelGGL((k<V, C>), dim3(grid), dim3(256), bytes, 0, 100, out);
  hipEventRecord(a);
  hipLaunchKernelGGL((k<V, C>), dim3(grid), dim3(256), bytes, 0, iters, out);
  hipEventRecord(b);
  hipEventSynchronize(b);
  float ms;
  hipEventElapsedTime(&ms, a, b);
  const double rows_per_cu = double(iters) * 256 * blocks_per_cu;
  // 600 M rows over 256 CUs = 2.34 M rows per CU
  printf("%-44s blocks/CU=%d  LDS %5.1f KiB/block  %.3f ms  ->  %.3f ms per 600 M rows (compute side only)\n", name, blocks_per_cu, bytes / 1024.0, ms,
         ms * (600e6 / 256) / rows_per_cu);
  hipFree(out);
}

int main() {
  for (int b : {2, 4}) {
    run<0, 1>("V0 6 atomics, lane columns", b);
    run<1, 1>("V1 cells (g,d,t) + qty hist, 1 copy", b);
    run<1, 2>("V1 cells (g,d,t) + qty hist, 2 copies", b);
    run<1, 4>("V1 cells (g,d,t) + qty hist, 4 copies", b);
    run<1, 8>("V1 cells (g,d,t) + qty hist, 8 copies", b);
    run<2, 1>("V2 3 f64 lane columns + joint hist, 1 copy", b);
    run<2, 2>("V2 3 f64 lane columns + joint hist, 2 copies", b);
  }
  return 0;
}
