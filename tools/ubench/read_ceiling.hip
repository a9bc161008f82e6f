// What a kernel that only READS can get out of HBM on this box: the ceiling the aggregation's roofline fraction is to be read
// against (8 TB/s is the data-sheet number; torch.sum reaches 5.96 TB/s, the Q1 aggregation 6.2–6.3).  One buffer of `gib`
// GiB summed by persistent workgroups; varied: threads per workgroup, workgroups per CU, 16-byte loads in flight per lane,
// plain / non-temporal loads, and K streams read side by side (the aggregation reads six columns, not one array).
// usage: read_ceiling [gib = 16]
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)

typedef unsigned long long u64x2 __attribute__((ext_vector_type(2)));

// STREAMS equal parts of the buffer read in step (tile t of every part by the same workgroup), U loads of 16 bytes per lane
// and stream in flight.
template <int BLOCK, int U, int STREAMS, bool NT>
__global__ __launch_bounds__(BLOCK) void read_kernel(const u64x2 *__restrict__ data, int64_t vecs_per_stream, unsigned long long *__restrict__ out) {
  constexpr int kTile = BLOCK * U;
  const int64_t tiles = vecs_per_stream / kTile;
  unsigned long long acc = 0;
  for (int64_t tile = blockIdx.x; tile < tiles; tile += gridDim.x) {
    u64x2 v[STREAMS][U];
#pragma unroll
    for (int s = 0; s < STREAMS; ++s) {
      const u64x2 *p = data + s * vecs_per_stream + tile * kTile + threadIdx.x;
#pragma unroll
      for (int u = 0; u < U; ++u) v[s][u] = NT ? __builtin_nontemporal_load(p + u * BLOCK) : p[u * BLOCK];
    }
#pragma unroll
    for (int s = 0; s < STREAMS; ++s) {
#pragma unroll
      for (int u = 0; u < U; ++u) acc += v[s][u].x ^ v[s][u].y;
    }
  }
  if (acc == 0x1234567ull) atomicAdd(out, acc);   // (keeps the loads)
}

// The same bytes with every workgroup walking a contiguous chunk of its own (what a stable partition pass does: workgroup b
// owns rows [b * chunk, (b + 1) * chunk)) instead of the tiles being dealt round-robin: G streams far apart instead of one front.
template <int BLOCK, int U, bool NT>
__global__ __launch_bounds__(BLOCK) void read_chunked_kernel(const u64x2 *__restrict__ data, int64_t vecs, unsigned long long *__restrict__ out) {
  constexpr int kTile = BLOCK * U;
  const int64_t tiles = vecs / kTile, per_block = (tiles + gridDim.x - 1) / gridDim.x;
  const int64_t first = blockIdx.x * per_block, last = first + per_block < tiles ? first + per_block : tiles;
  unsigned long long acc = 0;
  for (int64_t tile = first; tile < last; ++tile) {
    u64x2 v[U];
    const u64x2 *p = data + tile * kTile + threadIdx.x;
#pragma unroll
    for (int u = 0; u < U; ++u) v[u] = NT ? __builtin_nontemporal_load(p + u * BLOCK) : p[u * BLOCK];
#pragma unroll
    for (int u = 0; u < U; ++u) acc += v[u].x ^ v[u].y;
  }
  if (acc == 0x1234567ull) atomicAdd(out, acc);
}

template <int BLOCK, int U, bool NT>
static void run_chunked(const u64x2 *data, int64_t vecs, unsigned long long *out, int grid, hipEvent_t a, hipEvent_t b) {
  float best = 1e30f;
  for (int rep = 0; rep < 4; ++rep) {
    CHECK(hipEventRecord(a));
    hipLaunchKernelGGL((read_chunked_kernel<BLOCK, U, NT>), dim3(grid), dim3(BLOCK), 0, 0, data, vecs, out);
    CHECK(hipEventRecord(b));
    CHECK(hipEventSynchronize(b));
    float ms = 0;
    CHECK(hipEventElapsedTime(&ms, a, b));
    if (rep > 0 && ms < best) best = ms;
  }
  const double bytes = static_cast<double>(vecs / (BLOCK * U)) * (BLOCK * U) * 16.0;
  printf("{\"chunk_per_workgroup\": true, \"block\": %d, \"loads_in_flight\": %d, \"nontemporal\": %s, \"workgroups\": %d, \"ms\": %.3f, \"GBps\": %.0f}\n",
         BLOCK, U, NT ? "true" : "false", grid, best, bytes / best / 1e6);
  fflush(stdout);
}

template <int BLOCK, int U, int STREAMS, bool NT>
static void run(const u64x2 *data, int64_t vecs, unsigned long long *out, int per_cu, hipEvent_t a, hipEvent_t b) {
  const int64_t per_stream = vecs / STREAMS;
  const int grid = 256 * per_cu;
  float best = 1e30f;
  for (int rep = 0; rep < 4; ++rep) {
    CHECK(hipEventRecord(a));
    hipLaunchKernelGGL((read_kernel<BLOCK, U, STREAMS, NT>), dim3(grid), dim3(BLOCK), 0, 0, data, per_stream, out);
    CHECK(hipEventRecord(b));
    CHECK(hipEventSynchronize(b));
    float ms = 0;
    CHECK(hipEventElapsedTime(&ms, a, b));
    if (rep > 0 && ms < best) best = ms;
  }
  const double bytes = static_cast<double>(per_stream / (BLOCK * U)) * (BLOCK * U) * STREAMS * 16.0;
  printf("{\"block\": %d, \"loads_in_flight\": %d, \"streams\": %d, \"nontemporal\": %s, \"workgroups_per_cu\": %d, \"ms\": %.3f, \"GBps\": %.0f}\n",
         BLOCK, U, STREAMS, NT ? "true" : "false", per_cu, best, bytes / best / 1e6);
  fflush(stdout);
}

int main(int argc, char **argv) {
  const int64_t gib = argc > 1 ? atoll(argv[1]) : 16;
  const int64_t vecs = gib * (1ll << 30) / 16;
  u64x2 *data = nullptr;
  unsigned long long *out = nullptr;
  CHECK(hipMalloc(&data, vecs * 16));
  CHECK(hipMalloc(&out, 8));
  CHECK(hipMemset(data, 1, vecs * 16));
  CHECK(hipMemset(out, 0, 8));
  hipEvent_t a, b;
  CHECK(hipEventCreate(&a));
  CHECK(hipEventCreate(&b));
  for (int grid : {256, 512, 1024, 2048, 8192}) {
    run_chunked<256, 4, false>(data, vecs, out, grid, a, b);
    run_chunked<256, 4, true>(data, vecs, out, grid, a, b);
  }
  for (int per_cu : {1, 2, 4, 8}) {
    run<256, 4, 1, false>(data, vecs, out, per_cu, a, b);
    run<256, 4, 1, true>(data, vecs, out, per_cu, a, b);
    run<256, 8, 1, true>(data, vecs, out, per_cu, a, b);
    run<512, 4, 1, true>(data, vecs, out, per_cu, a, b);
    run<1024, 2, 1, true>(data, vecs, out, per_cu, a, b);
    run<1024, 4, 1, true>(data, vecs, out, per_cu, a, b);
    run<256, 2, 6, true>(data, vecs, out, per_cu, a, b);
    run<256, 1, 6, true>(data, vecs, out, per_cu, a, b);
    run<512, 1, 6, true>(data, vecs, out, per_cu, a, b);
    run<256, 2, 6, false>(data, vecs, out, per_cu, a, b);
  }
  return 0;
}
