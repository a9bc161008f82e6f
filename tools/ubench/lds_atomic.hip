// Microbenchmark: throughput of LDS atomics on lane-private bank columns
// (ds_add_f64 / ds_add_u64 / ds_add_u32), one 256-thread workgroup per CU slot.
// build: hipcc --offload-arch=gfx950 -O3 -munsafe-fp-atomics tools/ubench/lds_atomic.hip -o /tmp/lds_atomic
#include <hip/hip_runtime.h>
#include <cstdio>

template <int MODE>
__global__ __launch_bounds__(256) void k(int iters, int slots, double *out) {
  extern __shared__ double acc[];  // [slots][64]
  const int lane = threadIdx.x & 63;
  for (int i = threadIdx.x; i < slots * 64; i += 256) acc[i] = 0;
  __syncthreads();
  unsigned s = (threadIdx.x * 2654435761u) >> 8;
  for (int i = 0; i < iters; ++i) {
    s = s * 1664525u + 1013904223u;
    const int slot = (s >> 16) % slots;   // random group per lane, like real data
    if (MODE == 0) unsafeAtomicAdd(&acc[slot * 64 + lane], 1.5);
    if (MODE == 1) atomicAdd(reinterpret_cast<unsigned long long *>(&acc[slot * 64 + lane]), 3ull);
    if (MODE == 2) atomicAdd(reinterpret_cast<unsigned *>(&acc[slot * 64 + lane]), 3u);
    if (MODE == 3) acc[slot * 64 + lane] += 1.5;  // non-atomic read-modify-write (wrong across waves; rate reference)
    // how the cost of a wave's atomic scales with its active lanes (an exec-masked add: rows folded away, filtered rows)
    if (MODE == 4 && lane < 32) unsafeAtomicAdd(&acc[slot * 64 + lane], 1.5);          // lower half of the wave
    if (MODE == 5 && (s & 0x300) == 0) unsafeAtomicAdd(&acc[slot * 64 + lane], 1.5);    // a random quarter of the lanes
    if (MODE == 6 && (lane & 15) == 0) unsafeAtomicAdd(&acc[slot * 64 + lane], 1.5);    // 4 lanes
    if (MODE == 7) unsafeAtomicAdd(&acc[slot * 64 + (lane & 15)], 1.5);                 // 16 copies: 4 lanes per column (rep_shift 4)
  }
  __syncthreads();
  if (threadIdx.x < 64) out[blockIdx.x * 64 + threadIdx.x] = acc[threadIdx.x];
}

template <int MODE>
void run(const char *name, int blocks_per_cu) {
  double *out;
  hipMalloc(&out, 256 * 8 * 64 * 8);
  const int iters = 20000, slots = 24;
  hipEvent_t a, b;
  hipEventCreate(&a);
  hipEventCreate(&b);
  const int grid = 256 * blocks_per_cu;
  hipLaunchKernelGGL(k<MODE>, dim3(grid), dim3(256), slots * 64 * 8, 0, 100, slots, out);
  hipEventRecord(a);
  hipLaunchKernelGGL(k<MODE>, dim3(grid), dim3(256), slots * 64 * 8, 0, iters, slots, out);
  hipEventRecord(b);
  hipEventSynchronize(b);
  float ms;
  hipEventElapsedTime(&ms, a, b);
  const double wave_instrs_per_cu = double(iters) * 4 * blocks_per_cu;
  printf("%-28s blocks/CU=%d  %.3f ms  %.1f ns per wave-instr per CU  (%.2f G lane-ops/s chip)\n", name, blocks_per_cu, ms,
         ms * 1e6 / wave_instrs_per_cu, double(iters) * 256 * grid / ms / 1e6);
  hipFree(out);
}

int main() {
  for (int b : {1, 2, 4}) {
    run<0>("ds_add_f64 (lane columns)", b);
    run<1>("ds_add_u64 (lane columns)", b);
    run<2>("ds_add_u32 (lane columns)", b);
    run<3>("ds rmw f64 non-atomic", b);
    run<4>("ds_add_f64 32 of 64 lanes", b);
    run<5>("ds_add_f64 random quarter", b);
    run<6>("ds_add_f64 4 of 64 lanes", b);
    run<7>("ds_add_f64 16 copies (4 lanes/col)", b);
  }
  return 0;
}
