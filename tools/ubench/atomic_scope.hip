// What a random atomic add into a dense state array costs by SCOPE: 100 M random 8-byte atomic adds (u64 and f64) into a table
// of N entries,
//   agent      device-scope atomics, every workgroup anywhere in the table (what the dense aggregation does)
//   xcd_local  workgroup-scope atomics (executed in the XCD's own L2), every workgroup inside the eighth of the table that
//              belongs to the XCD it runs on (HW_REG_XCC_ID) — the access pattern of a state partitioned by XCD
//   xcd_agent  the same slicing with device-scope atomics (separates "slice fits L2" from "scope")
// and whether the xcd_local sums are right (they must be: one XCD's L2 is coherent for its own CUs, and the kernel's end writes
// it back).  usage: atomic_scope [rows = 100000000]
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)

__device__ __forceinline__ uint32_t rnd(uint64_t i, uint32_t range) {
  uint64_t x = i * 0x9E3779B97F4A7C15ull + 12345;
  x ^= x >> 29; x *= 0xBF58476D1CE4E5B9ull; x ^= x >> 32; x *= 0x94D049BB133111EBull; x ^= x >> 29;
  return (uint32_t)(((x >> 32) * (uint64_t)range) >> 32);
}
__device__ __forceinline__ int xcc_id() { return __builtin_amdgcn_s_getreg((3 << 11) | 20) & 7; }   // HW_REG_XCC_ID, 4 bits

// MODE 0 agent / whole table, 1 workgroup scope / own slice, 2 agent scope / own slice.  F64: atomic add of doubles.
template <int MODE, bool F64>
__global__ __launch_bounds__(256) void add_kernel(unsigned long long *__restrict__ table, uint32_t entries, int64_t n, unsigned int *xcd_rows) {
  const int x = xcc_id();
  const uint32_t slice = entries / 8;
  unsigned long long mine = 0;
  for (int64_t i = blockIdx.x * 256ll + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    uint32_t k = MODE == 0 ? rnd(i, entries) : x * slice + rnd(i, slice);
    if (F64) {
      double *p = reinterpret_cast<double *>(table) + k;
      if (MODE == 1) __hip_atomic_fetch_add(p, 1.0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      else __hip_atomic_fetch_add(p, 1.0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    } else {
      if (MODE == 1) __hip_atomic_fetch_add(&table[k], 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      else __hip_atomic_fetch_add(&table[k], 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    ++mine;
  }
  if (threadIdx.x == 0) atomicAdd(&xcd_rows[x], 1u);
  (void)mine;
}

template <int MODE, bool F64>
static void run(const char *name, unsigned long long *table, uint32_t entries, int64_t n, unsigned int *xcd_rows) {
  const int grid = 256 * 8;
  CHECK(hipMemset(table, 0, (size_t)entries * 8));
  hipLaunchKernelGGL((add_kernel<MODE, F64>), dim3(grid), dim3(256), 0, 0, table, entries, n, xcd_rows);
  CHECK(hipDeviceSynchronize());
  hipEvent_t a, b;
  CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
  CHECK(hipMemset(table, 0, (size_t)entries * 8));
  CHECK(hipEventRecord(a));
  const int reps = 3;
  for (int r = 0; r < reps; ++r) hipLaunchKernelGGL((add_kernel<MODE, F64>), dim3(grid), dim3(256), 0, 0, table, entries, n, xcd_rows);
  CHECK(hipEventRecord(b));
  CHECK(hipEventSynchronize(b));
  float ms = 0;
  CHECK(hipEventElapsedTime(&ms, a, b));
  // the sum over the table must be reps * n
  std::vector<unsigned long long> host(entries);
  CHECK(hipMemcpy(host.data(), table, (size_t)entries * 8, hipMemcpyDeviceToHost));
  double total = 0;
  for (uint32_t i = 0; i < entries; ++i) total += F64 ? *reinterpret_cast<double *>(&host[i]) : (double)host[i];
  printf("{\"mode\": \"%s\", \"type\": \"%s\", \"entries\": %u, \"table_MiB\": %.2f, \"rows\": %lld, \"ms\": %.3f, \"G_atomics_per_s\": %.1f, \"sum_ok\": %s}\n",
         name, F64 ? "f64" : "u64", entries, entries * 8.0 / (1 << 20), (long long)n, ms / reps, n / (ms / reps) / 1e6,
         total == (double)reps * (double)n ? "true" : "false");
  fflush(stdout);
}

int main(int argc, char **argv) {
  const int64_t n = argc > 1 ? atoll(argv[1]) : 100000000;
  unsigned int *xcd_rows;
  CHECK(hipMalloc(&xcd_rows, 64));
  CHECK(hipMemset(xcd_rows, 0, 64));
  for (uint32_t entries : {100000u * 8 / 8, 1000000u, 10000000u, 80000000u}) {
    entries = entries / 8 * 8;
    unsigned long long *table;
    CHECK(hipMalloc(&table, (size_t)entries * 8));
    run<0, false>("agent", table, entries, n, xcd_rows);
    run<1, false>("xcd_local", table, entries, n, xcd_rows);
    run<2, false>("xcd_agent", table, entries, n, xcd_rows);
    run<0, true>("agent", table, entries, n, xcd_rows);
    run<1, true>("xcd_local", table, entries, n, xcd_rows);
    CHECK(hipFree(table));
  }
  unsigned int h[8];
  CHECK(hipMemcpy(h, xcd_rows, 32, hipMemcpyDeviceToHost));
  printf("{\"workgroups_per_xcd\": [%u, %u, %u, %u, %u, %u, %u, %u]}\n", h[0], h[1], h[2], h[3], h[4], h[5], h[6], h[7]);
  return 0;
}
