// What bounds "stream the keys, read one table word per row": random reads of a table of S MiB by 100 M keys, as a function of
// the table size (L2-resident or not), the entry width (4-byte words, 3-byte packed entries read with an unaligned 4-byte load,
// 8-byte entries, 16-byte units) and the number of reads a lane keeps in flight.  Count-only (a sum) and pair-emitting
// (8 bytes written per row, non-temporal) forms.
// usage: gather_floor [rows = 100000000]
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)

__global__ void fill_keys(uint32_t *keys, int64_t n, uint32_t range, uint64_t seed) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    uint64_t x = (uint64_t)i * 0x9E3779B97F4A7C15ull + seed;
    x ^= x >> 29; x *= 0xBF58476D1CE4E5B9ull; x ^= x >> 32; x *= 0x94D049BB133111EBull; x ^= x >> 29;
    keys[i] = (uint32_t)(((x >> 32) * (uint64_t)range) >> 32);
  }
}

// WIDTH: 3 (packed 24-bit entries), 4, 8, 16 bytes per entry.  EMIT: write (row, word) pairs.
template <int WIDTH, int R, bool EMIT>
__global__ __launch_bounds__(256) void gather_kernel(const uint32_t *__restrict__ keys, int64_t n, const unsigned char *__restrict__ table,
                                                      unsigned long long *__restrict__ sum, int32_t *__restrict__ out_a, int32_t *__restrict__ out_b) {
  constexpr int kTile = 256 * R;
  const int64_t tiles = n / kTile;
  unsigned long long acc = 0;
  uint32_t key[R], next_key[R];
  int64_t tile = blockIdx.x;
  if (tile < tiles) {
#pragma unroll
    for (int r = 0; r < R; ++r) key[r] = __builtin_nontemporal_load(&keys[tile * kTile + r * 256 + threadIdx.x]);
  }
  for (; tile < tiles; tile += gridDim.x) {
    if (tile + gridDim.x < tiles) {
#pragma unroll
      for (int r = 0; r < R; ++r) next_key[r] = __builtin_nontemporal_load(&keys[(tile + gridDim.x) * kTile + r * 256 + threadIdx.x]);
    }
    uint32_t h[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
      if (WIDTH == 4) {
        h[r] = reinterpret_cast<const uint32_t *>(table)[key[r]];
      } else if (WIDTH == 3) {
        uint32_t w;
        __builtin_memcpy(&w, table + (uint64_t)key[r] * 3, 4);
        h[r] = w & 0xFFFFFFu;
      } else if (WIDTH == 8) {
        const uint2 w = reinterpret_cast<const uint2 *>(table)[key[r]];
        h[r] = w.x ^ w.y;
      } else {
        const uint4 w = reinterpret_cast<const uint4 *>(table)[key[r]];
        h[r] = w.x ^ w.y ^ w.z ^ w.w;
      }
    }
#pragma unroll
    for (int r = 0; r < R; ++r) {
      if (EMIT) {
        const int64_t row = tile * kTile + r * 256 + threadIdx.x;
        __builtin_nontemporal_store((int32_t)row, &out_a[row]);
        __builtin_nontemporal_store((int32_t)h[r], &out_b[row]);
      } else {
        acc += h[r];
      }
    }
#pragma unroll
    for (int r = 0; r < R; ++r) key[r] = next_key[r];
  }
  if (!EMIT) {
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);
    if ((threadIdx.x & 63) == 0) atomicAdd(sum, acc);
  }
}

template <int WIDTH, int R, bool EMIT>
static float run(const uint32_t *keys, int64_t n, const unsigned char *table, unsigned long long *sum, int32_t *a, int32_t *b, int blocks_per_cu) {
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0));
  CHECK(hipEventCreate(&e1));
  const int grid = 256 * blocks_per_cu;
  gather_kernel<WIDTH, R, EMIT><<<grid, 256>>>(keys, n, table, sum, a, b);
  CHECK(hipEventRecord(e0));
  const int reps = 5;
  for (int i = 0; i < reps; ++i) gather_kernel<WIDTH, R, EMIT><<<grid, 256>>>(keys, n, table, sum, a, b);
  CHECK(hipEventRecord(e1));
  CHECK(hipEventSynchronize(e1));
  float ms = 0;
  CHECK(hipEventElapsedTime(&ms, e0, e1));
  return ms / reps;
}

int main(int argc, char **argv) {
  const int64_t n = (argc > 1 ? atoll(argv[1]) : 100000000ll) / 4096 * 4096;
  uint32_t *keys;
  unsigned char *table;
  unsigned long long *sum;
  int32_t *a, *b;
  const size_t table_bytes = 64u << 20;
  CHECK(hipMalloc(&keys, n * 4));
  CHECK(hipMalloc(&table, table_bytes + 16));
  CHECK(hipMemset(table, 1, table_bytes + 16));
  CHECK(hipMalloc(&sum, 8));
  CHECK(hipMalloc(&a, n * 4));
  CHECK(hipMalloc(&b, n * 4));
  const double mibs[] = {1, 2, 2.5, 3, 3.5, 4, 6, 8, 16, 32};
  for (double mib : mibs) {
    const size_t bytes = (size_t)(mib * 1048576);
    for (int width : {4, 3, 8, 16}) {
      const uint32_t entries = (uint32_t)(bytes / width);
      fill_keys<<<1024, 256>>>(keys, n, entries, 12345);
      CHECK(hipDeviceSynchronize());
      float c8, c16, c32, e16, c16_8 = 0;
      if (width == 4) {
        c8 = run<4, 8, false>(keys, n, table, sum, a, b, 8); c16 = run<4, 16, false>(keys, n, table, sum, a, b, 8);
        c32 = run<4, 32, false>(keys, n, table, sum, a, b, 4); e16 = run<4, 16, true>(keys, n, table, sum, a, b, 8);
        c16_8 = run<4, 16, false>(keys, n, table, sum, a, b, 4);
      } else if (width == 3) {
        c8 = run<3, 8, false>(keys, n, table, sum, a, b, 8); c16 = run<3, 16, false>(keys, n, table, sum, a, b, 8);
        c32 = run<3, 32, false>(keys, n, table, sum, a, b, 4); e16 = run<3, 16, true>(keys, n, table, sum, a, b, 8);
      } else if (width == 8) {
        c8 = run<8, 8, false>(keys, n, table, sum, a, b, 8); c16 = run<8, 16, false>(keys, n, table, sum, a, b, 8);
        c32 = run<8, 32, false>(keys, n, table, sum, a, b, 4); e16 = run<8, 16, true>(keys, n, table, sum, a, b, 8);
      } else {
        c8 = run<16, 8, false>(keys, n, table, sum, a, b, 8); c16 = run<16, 16, false>(keys, n, table, sum, a, b, 8);
        c32 = run<16, 16, false>(keys, n, table, sum, a, b, 4); e16 = run<16, 16, true>(keys, n, table, sum, a, b, 8);
      }
      printf("{\"table_MiB\": %.1f, \"entry_bytes\": %d, \"rows\": %lld, \"count_ms_R8\": %.3f, \"count_ms_R16\": %.3f, \"count_ms_R32_or_4perCU\": %.3f, "
             "\"count_ms_R16_4perCU\": %.3f, \"pairs_ms_R16\": %.3f}\n", mib, width, (long long)n, c8, c16, c32, c16_8, e16);
      fflush(stdout);
    }
  }
  return 0;
}
