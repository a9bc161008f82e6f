// Microbenchmark: does an LDS read of a wave wait for that wave's outstanding LDS-DMA (global_load_lds)?
// A wave issues one 1 KiB DMA (asm statement, uncounted by the compiler) from a cold global address into LDS buffer B, then
// reads LDS buffer A and waits lgkmcnt(0) (t1), then waits vmcnt(0) (t2).  t1 << t2: LDS traffic runs under the DMA.
// t1 ~ t2: the LGKM wait (or the LDS pipeline) holds the wave until the DMA has landed — prefetching a tile from the
// wave that computes cannot overlap then.  Also timed: the same with a ds_add_f64 instead of the read.
// build: hipcc --offload-arch=gfx950 -O3 -munsafe-fp-atomics tools/ubench/lds_dma_wait.hip -o tools/ubench/lds_dma_wait
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

__device__ __forceinline__ void dma16(const char *g, char *lds) {
  const unsigned at = __builtin_amdgcn_readfirstlane(static_cast<unsigned>(reinterpret_cast<uintptr_t>((__attribute__((address_space(3))) char *)lds)));
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off nt\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(g), "s"(at) : "memory");
}

template <int MODE>   // 0: ds_read after the DMA, 1: ds_add_f64 after the DMA, 2: no DMA (reference for the LDS op alone)
__global__ __launch_bounds__(64) void k(const char *src, size_t stride, int iters, unsigned long long *out) {
  __shared__ __attribute__((aligned(16))) char lds[8192];
  const int lane = threadIdx.x;
  double *a = reinterpret_cast<double *>(lds);
  a[lane] = 1.0;
  __syncthreads();
  unsigned long long s1 = 0, s2 = 0;
  double sink = 0;
  for (int i = 0; i < iters; ++i) {
    const char *g = src + (static_cast<size_t>(blockIdx.x) * iters + i) * stride + lane * 16;
    const unsigned long long t0 = __builtin_readcyclecounter();
    if (MODE != 2) dma16(g, lds + 4096);
    if (MODE == 1) {
      unsafeAtomicAdd(&a[lane], 1.5);
    } else {
      double x;
      asm volatile("ds_read_b64 %0, %1" : "=v"(x) : "v"(static_cast<unsigned>(lane * 8)) : "memory");
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      sink += x;
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    const unsigned long long t1 = __builtin_readcyclecounter();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned long long t2 = __builtin_readcyclecounter();
    s1 += t1 - t0;
    s2 += t2 - t0;
  }
  if (lane == 0) {
    out[blockIdx.x * 2] = s1;
    out[blockIdx.x * 2 + 1] = s2;
  }
  if (sink == 12345.678) out[0] = 0;
}

template <int MODE>
void run(const char *name, const char *src, size_t stride, int grid, int iters, unsigned long long *out) {
  hipLaunchKernelGGL(k<MODE>, dim3(grid), dim3(64), 0, 0, src, stride, iters, out);
  hipDeviceSynchronize();
  std::vector<unsigned long long> h(grid * 2);
  hipMemcpy(h.data(), out, grid * 16, hipMemcpyDeviceToHost);
  double s1 = 0, s2 = 0;
  for (int b = 0; b < grid; ++b) { s1 += h[b * 2]; s2 += h[b * 2 + 1]; }
  std::printf("{\"case\": \"%s\", \"grid\": %d, \"cycles_until_lgkmcnt0\": %.0f, \"cycles_until_vmcnt0\": %.0f}\n", name, grid,
              s1 / grid / iters, s2 / grid / iters);
}

int main() {
  const int iters = 200;
  const size_t stride = 1 << 20;   // a cold line per iteration
  char *src;
  unsigned long long *out;
  const int max_grid = 1024;
  hipMalloc(&src, static_cast<size_t>(max_grid) * iters * stride / 64 + (1 << 20));   // (stride below is divided for the large grid)
  hipMalloc(&out, max_grid * 16);
  for (int grid : {1, 256, 1024}) {
    const size_t st = grid == 1 ? stride : stride / 64;
    run<0>("dma_then_ds_read", src, st, grid, iters, out);
    run<1>("dma_then_ds_add_f64", src, st, grid, iters, out);
    run<2>("ds_read_alone", src, st, grid, iters, out);
  }
  return 0;
}
