// Prices "fingerprints INSIDE the bucket's line" against "a fingerprint plane in front of the slots" for the hashed join table
// over sparse keys (DESIGN.md open item 10): 100 M probes of 1 M keys, every probe a hit.
//   plane:  16 bytes of a 1.25 MiB plane (L2-resident), then 8 bytes of a 10 MiB slot array (one random 128-byte line)
//   inline: 16 bytes at the end of the bucket's 128-byte line (10 MiB of lines), then 8 bytes of the SAME line
// Both: the second read depends on the first.  Pairs are written (8 bytes per row, non-temporal).
// build: hipcc --offload-arch=gfx950 -O3 tools/ubench/line_probe.hip -o /tmp/line_probe
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>

__global__ void fill_keys(uint32_t *keys, int64_t n, uint32_t range, uint64_t seed) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    uint64_t x = (uint64_t)i * 0x9E3779B97F4A7C15ull + seed;
    x ^= x >> 29; x *= 0xBF58476D1CE4E5B9ull; x ^= x >> 32; x *= 0x94D049BB133111EBull; x ^= x >> 29;
    keys[i] = (uint32_t)(((x >> 32) * (uint64_t)range) >> 32);
  }
}

// MODE 0: plane + slots, 1: in-line, 2: a single 8-byte read of the 10 MiB array (the floor of one random line),
// 3: plane + a 4-byte slot of an array HALF the size (64-byte buckets of 16 4-byte slots: the "compact slot" of open item 12)
template <int MODE, int R>
__global__ __launch_bounds__(256) void probe(const uint32_t *__restrict__ keys, int64_t n, const uint4 *__restrict__ plane,
                                             const unsigned long long *__restrict__ lines, uint32_t buckets, int32_t *__restrict__ out_a,
                                             int32_t *__restrict__ out_b, unsigned long long *__restrict__ sum) {
  constexpr int kTile = 256 * R;
  const int64_t tiles = n / kTile;
  unsigned long long acc = 0;
  for (int64_t tile = blockIdx.x; tile < tiles; tile += gridDim.x) {
    uint32_t b[R];
#pragma unroll
    for (int r = 0; r < R; ++r) b[r] = __builtin_nontemporal_load(&keys[tile * kTile + r * 256 + threadIdx.x]);
    uint32_t pick[R];
    if (MODE == 2) {
#pragma unroll
      for (int r = 0; r < R; ++r) pick[r] = b[r] & 7;
    } else {
      uint4 w[R];
#pragma unroll
      for (int r = 0; r < R; ++r) w[r] = (MODE == 0 || MODE == 3) ? plane[b[r]] : reinterpret_cast<const uint4 *>(lines + (uint64_t)b[r] * 16)[7];
#pragma unroll
      for (int r = 0; r < R; ++r) pick[r] = (w[r].x ^ w[r].y ^ w[r].z ^ w[r].w ^ b[r]) % 14;
    }
    unsigned long long e[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
      if (MODE == 3) e[r] = reinterpret_cast<const uint32_t *>(lines)[(uint64_t)b[r] * 16 + pick[r]];
      else e[r] = lines[(uint64_t)b[r] * 16 + pick[r]];
    }
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const int64_t row = tile * kTile + r * 256 + threadIdx.x;
      __builtin_nontemporal_store((int32_t)row, &out_a[row]);
      __builtin_nontemporal_store((int32_t)e[r], &out_b[row]);
      acc += e[r] >> 32;
    }
  }
  if (acc == 0x123456789ull) atomicAdd(sum, acc);
}

template <int MODE, int R>
void run(const char *name, const uint32_t *keys, int64_t n, const uint4 *plane, const unsigned long long *lines, uint32_t buckets, int32_t *oa, int32_t *ob,
         unsigned long long *sum, int per_cu) {
  hipEvent_t a, b;
  hipEventCreate(&a); hipEventCreate(&b);
  float best = 1e9f;
  for (int rep = 0; rep < 4; ++rep) {
    hipEventRecord(a);
    hipLaunchKernelGGL((probe<MODE, R>), dim3(256 * per_cu), dim3(256), 0, 0, keys, n, plane, lines, buckets, oa, ob, sum);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms;
    hipEventElapsedTime(&ms, a, b);
    if (rep > 0 && ms < best) best = ms;
  }
  printf("%-44s R=%2d workgroups/CU=%d  %.3f ms per %lld M probes\n", name, R, per_cu, best, (long long)(n / 1000000));
}

int main(int argc, char **argv) {
  const int64_t n = (argc > 1 ? atoll(argv[1]) : 100) * 1000000ll / 4096 * 4096;
  const uint32_t buckets = argc > 2 ? atoi(argv[2]) : 81920;       // x 128 bytes = 10 MiB
  uint32_t *keys; uint4 *plane; unsigned long long *lines, *sum; int32_t *oa, *ob;
  hipMalloc(&keys, n * 4); hipMalloc(&plane, (size_t)buckets * 16); hipMalloc(&lines, (size_t)buckets * 128); hipMalloc(&sum, 8);
  hipMalloc(&oa, n * 4); hipMalloc(&ob, n * 4);
  hipMemset(plane, 1, (size_t)buckets * 16); hipMemset(lines, 2, (size_t)buckets * 128); hipMemset(sum, 0, 8);
  hipLaunchKernelGGL(fill_keys, dim3(4096), dim3(256), 0, 0, keys, n, buckets, 7ull);
  hipDeviceSynchronize();
  for (int per_cu : {4, 8}) {
    run<2, 16>("one random line (8 bytes)", keys, n, plane, lines, buckets, oa, ob, sum, per_cu);
    run<0, 16>("plane (16 B, L2) then slot line (8 B)", keys, n, plane, lines, buckets, oa, ob, sum, per_cu);
    run<1, 16>("in-line: 16 B then 8 B of the same line", keys, n, plane, lines, buckets, oa, ob, sum, per_cu);
    run<1, 8>("in-line: 16 B then 8 B of the same line", keys, n, plane, lines, buckets, oa, ob, sum, per_cu);
    run<3, 16>("plane then a 4-byte slot of a 5 MiB array", keys, n, plane, lines, buckets, oa, ob, sum, per_cu);
  }
  return 0;
}
