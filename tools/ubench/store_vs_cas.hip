// What claiming a word of a directly addressed table costs: N random 4-byte words of a table of S MiB
//   cas     atomicCAS(&head[k], 0, tid + 1)   — what dense_build_kernel does per build row (the global-atomic ceiling)
//   store   head[k] = tid + 1                 — a divergent plain store, nothing comes back
//   verify  sum += head[k] == tid + 1         — the read-back an optimistic build would need (a gather)
// for tables inside one XCD's L2 (4 MiB), inside the MALL (75 MiB) and beyond (512 MiB).  DESIGN §4 open items.
// usage: store_vs_cas [rows = 20000000]
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)

// a permutation-like spread: distinct rows mostly hit distinct words (a primary key)
__device__ __forceinline__ uint32_t slot_of(uint64_t i, uint32_t words) {
  uint64_t x = i * 0x9E3779B97F4A7C15ull + 12345;
  x ^= x >> 29; x *= 0xBF58476D1CE4E5B9ull; x ^= x >> 32; x *= 0x94D049BB133111EBull; x ^= x >> 29;
  return (uint32_t)(((x >> 32) * (uint64_t)words) >> 32);
}

template <int MODE>
__global__ __launch_bounds__(256) void claim_kernel(uint32_t *__restrict__ head, uint32_t words, int64_t n, unsigned long long *__restrict__ out) {
  constexpr int R = 8;
  unsigned long long hits = 0;
  const int64_t stride = (int64_t)gridDim.x * 256;
  for (int64_t i0 = blockIdx.x * 256ll + threadIdx.x; i0 < n; i0 += stride * R) {
    uint32_t k[R];
#pragma unroll
    for (int r = 0; r < R; ++r) k[r] = slot_of(i0 + r * stride, words);
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const int64_t i = i0 + r * stride;
      if (i >= n) continue;
      const uint32_t mine = (uint32_t)i + 1u;
      if (MODE == 0) hits += atomicCAS(&head[k[r]], 0u, mine) == 0u;
      else if (MODE == 1) head[k[r]] = mine;
      else hits += head[k[r]] == mine;
    }
  }
  if (MODE != 1 && hits != 0) atomicAdd(out, hits);
}

template <int MODE>
static float timed(uint32_t *head, uint32_t words, int64_t n, unsigned long long *out, bool clear) {
  hipEvent_t a, b;
  CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
  float best = 1e30f;
  for (int rep = 0; rep < 3; ++rep) {
    if (clear) CHECK(hipMemset(head, 0, (size_t)words * 4));
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(a));
    hipLaunchKernelGGL((claim_kernel<MODE>), dim3(2048), dim3(256), 0, 0, head, words, n, out);
    CHECK(hipEventRecord(b));
    CHECK(hipEventSynchronize(b));
    float ms = 0;
    CHECK(hipEventElapsedTime(&ms, a, b));
    if (ms < best) best = ms;
  }
  return best;
}

int main(int argc, char **argv) {
  const int64_t n = argc > 1 ? atoll(argv[1]) : 20000000;
  unsigned long long *out;
  CHECK(hipMalloc(&out, 8));
  for (double mib : {4.0, 75.0, 512.0}) {
    const uint32_t words = (uint32_t)(mib * (1 << 20) / 4);
    uint32_t *head;
    CHECK(hipMalloc(&head, (size_t)words * 4));
    const float cas = timed<0>(head, words, n, out, true);
    const float store = timed<1>(head, words, n, out, true);
    CHECK(hipMemset(out, 0, 8));
    const float verify = timed<2>(head, words, n, out, false);   // the table as the stores left it
    unsigned long long owners = 0;
    CHECK(hipMemcpy(&owners, out, 8, hipMemcpyDeviceToHost));
    printf("{\"table_MiB\": %.0f, \"rows\": %lld, \"cas_ms\": %.3f, \"store_ms\": %.3f, \"verify_ms\": %.3f, \"G_cas_per_s\": %.1f, \"G_stores_per_s\": %.1f, "
           "\"rows_that_own_their_word\": %.4f}\n",
           mib, (long long)n, cas, store, verify, n / cas / 1e6, n / store / 1e6, owners / 3.0 / (double)n);
    fflush(stdout);
    CHECK(hipFree(head));
  }
  return 0;
}
