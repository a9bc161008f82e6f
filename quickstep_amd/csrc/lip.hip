// lip.hip — K12: LIP (lookahead information passing) filters.
//
// Reference (paths in the Quickstep tree):
//   utility/lip_filter/SingleIdentityHashFilter.hpp:97-107,156-169  bit = value % cardinality
//   utility/lip_filter/BitVectorExactFilter.hpp:140-176            bit = value - min
//   utility/lip_filter/LIPFilterAdaptiveProber.hpp:83-90,113-228   probe = AND over filters
//   relational_operators/BuildHashOperator.cpp:187-190             build inside BuildHashWorkOrder
// The adaptive re-ordering of filters only changes speed, never the result
// set; here each filter is one pass that ANDs into a bitmap.

#include "common.hpp"
#include "block_runs.hpp"
#include "lip_view.hpp"

#include <vector>

namespace qsx {

constexpr int kLBlock = 256;

// A wave owns groups of R x 64 rows; the R filter words of a group come with one load (lane r holds word r) and the
// keys of the next group are requested before the bits of the current one are set (as in lip_probe_kernel below).
// kRuns: the rows are a run of blocks (qsx_lip_build_blocks; block_runs.hpp), a group belongs to one block.
template <typename KeyT, int R, bool kRuns = false>
__global__ __launch_bounds__(kLBlock) void lip_build_kernel(LipView f, const KeyT *__restrict__ keys, int64_t n_arg,
                                                           const uint64_t *__restrict__ filter, const long long *__restrict__ runs = nullptr) {
  using Source = ProbeTileSource<KeyT>;
  const int lane = lane_id();
  const int64_t num_groups = kRuns ? runs[2] : (((n_arg + 63) >> 6) + R - 1) / R;
  const int64_t wave = __builtin_amdgcn_readfirstlane(static_cast<int>(blockIdx.x * (kLBlock / kWave) + (threadIdx.x >> 6)));
  const int64_t num_waves = static_cast<int64_t>(gridDim.x) * (kLBlock / kWave);
  KeyT key[R], next_key[R];
  uint64_t words = ~0ull, next_words = ~0ull;
  auto source_of = [&](int64_t group) { return probe_tile_source<KeyT, R * kWave, kRuns>(runs, group, keys, n_arg, 0, filter, nullptr); };
  auto request = [&](const Source &src, KeyT (&k)[R], uint64_t &fw) {
    const int64_t sw0 = src.base >> 6;
    if (kRuns && src.code_width != 0) {   // a compressed key stripe (block_runs.hpp): read as it lies
      coded_keys(src, k, [&](int r) {
        const int64_t row = ((sw0 + r) << 6) + lane;
        return row < src.n ? row : src.n - 1;
      });
    } else {
#pragma unroll
      for (int r = 0; r < R; ++r) {
        const int64_t row = ((sw0 + r) << 6) + lane;
        k[r] = src.keys[row < src.n ? row : src.n - 1];   // clamped, not guarded: no branch around the read
      }
    }
    fw = ~0ull;
    if (src.filter != nullptr && lane < R && sw0 + lane < ((src.n + 63) >> 6)) fw = src.filter[sw0 + lane];
  };
  Source cur = Source(), next = Source();
  int64_t group = wave;
  if (group < num_groups) {
    cur = source_of(group);
    request(cur, key, words);
  }
  for (; group < num_groups; group += num_waves) {
    if (group + num_waves < num_groups) {
      next = source_of(group + num_waves);
      request(next, next_key, next_words);
    }
    const int64_t w0 = cur.base >> 6;
    const int64_t n = cur.n;
    cur = next;
    // all R filter words are read first (unconditionally: dead lanes read word 0), then the missing bits are set with
    // fire-and-forget atomics: one round trip per group instead of one per row
    unsigned long long bit[R];
    bool set_it[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const int64_t row = ((w0 + r) << 6) + lane;
      const uint64_t fw = __shfl(words, r, kWave);   // before any branch: every lane takes part
      const long long v = static_cast<long long>(key[r]);
      bool ok = row < n && msb_bit(fw, lane);
      if (f.exact) {
        const long long off = v - f.min_value;
        ok = ok && off >= 0 && off < f.cardinality;      // outside the declared [min, max]: cannot be represented
        bit[r] = static_cast<unsigned long long>(off);
      } else {
        // value converted to size_t first: a negative key sign-extends (SingleIdentityHashFilter.hpp:156-169)
        bit[r] = static_cast<unsigned long long>(v) % static_cast<unsigned long long>(f.cardinality);
      }
      if (!ok) bit[r] = 0;
      set_it[r] = ok;
    }
    unsigned long long have[R];
#pragma unroll
    for (int r = 0; r < R; ++r) have[r] = __hip_atomic_load(&f.words[bit[r] >> 6], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    // Neighbouring lanes that set bits of the SAME word hand their bits up the wave and the last lane of the run sets them with
    // one atomic: a build side in key order (o_orderkey of the qualifying orders, c_custkey under a filter) puts 6 - 13 keys
    // into a 64-bit word, and the device retires 23.7 G atomics/s whatever they carry (5.5 M keys: 0.25 ms one by one).  Bits
    // handed to a lane whose word merely repeats further down the wave are set twice, which an OR does not mind.
    const bool merge = f.cardinality <= (1ll << 37);   // (word numbers are compared as 32-bit values)
#pragma unroll
    for (int r = 0; r < R; ++r) {
      unsigned long long mask = set_it[r] ? 1ull << (bit[r] & 63) : 0ull;
      if (merge) {
        const unsigned int word = set_it[r] ? static_cast<unsigned int>(bit[r] >> 6) : 0xFFFFFFFFu;
#pragma unroll
        for (int off = 1; off < kWave; off <<= 1) {
          const unsigned int word_below = __shfl_up(word, off, kWave);
          const unsigned long long mask_below = __shfl_up(mask, off, kWave);
          if (lane >= off && word_below == word) mask |= mask_below;
        }
        const unsigned int word_above = __shfl_down(word, 1, kWave);
        if (lane != kWave - 1 && word_above == word) mask = 0;   // the lane above carries these bits on
      }
      if (mask != 0 && (have[r] & mask) != mask) atomicOr(&f.words[bit[r] >> 6], mask);
    }
#pragma unroll
    for (int r = 0; r < R; ++r) key[r] = next_key[r];
    words = next_words;
  }
}

// R bitmap words (64-row groups) per wave iteration: R independent key loads, then R independent
// 4-byte reads of the filter in flight per lane (one dependent pair per iteration left the probe
// latency-bound: 0.52 ms / 100 M rows).  Filters of up to kLipLdsWords 32-bit words are copied into
// LDS first (a 1 M-key exact filter is 128 KiB): the random reads then never leave the CU.
constexpr int kLipLdsWords = 32 * 1024;  // 128 KiB
// kRuns: the rows are a run of blocks (qsx_lip_probe_blocks): a tile belongs to one block and takes its key stripe, row
// count, input and output bitmap from the run table (whose tiles are R x blockDim.x rows); the counter is the run's.
template <typename KeyT, int R, bool kInLds, bool kRuns = false>
__global__ __launch_bounds__(kInLds ? 1024 : kLBlock) void lip_probe_kernel(LipView f, const KeyT *__restrict__ keys, int64_t n_arg,
                                                           const uint64_t *__restrict__ in_bitmap_arg,
                                                           uint64_t *__restrict__ out_bitmap_arg,
                                                           unsigned long long *__restrict__ out_count,
                                                           const long long *__restrict__ runs = nullptr) {
  using Source = ProbeTileSource<KeyT>;
  extern __shared__ uint32_t s_filter[];
  const uint32_t *__restrict__ bits32 = reinterpret_cast<const uint32_t *>(f.words);
  if (kInLds) {
    const long long words32 = (f.cardinality + 31) >> 5;
    for (long long i = threadIdx.x; i < words32; i += blockDim.x) s_filter[i] = bits32[i];
    __syncthreads();
  }
  const int lane = lane_id();
  const int waves_per_block = static_cast<int>(blockDim.x) / kWave;   // LDS variant: 16 waves share one copy of the filter
  const int wave = threadIdx.x >> 6;
  unsigned long long count = 0;
  // A workgroup owns tiles of R x blockDim.x rows; row r of a thread is tile_base + r * blockDim.x + tid, so the
  // workgroup reads blockDim.x consecutive keys per step and a wave's ballot is one bitmap word (index
  // tile_word0 + r * waves_per_block + wave; lane r loads / stores word r: one load and one store per tile and wave).
  // The keys and input words of the NEXT tile are requested before the filter words of the current one are read.
  constexpr int kTileRows = R * (kInLds ? 1024 : kLBlock);   // = R * blockDim.x
  const int64_t num_tiles = kRuns ? runs[2] : (n_arg + kTileRows - 1) / kTileRows;
  KeyT key[R], next_key[R];
  uint64_t in_words = ~0ull, next_in_words = ~0ull;
  auto source_of = [&](int64_t tile) {
    return probe_tile_source<KeyT, kTileRows, kRuns>(runs, tile, keys, n_arg, 0, in_bitmap_arg, out_bitmap_arg);
  };
  auto request = [&](const Source &src, KeyT (&k)[R], uint64_t &words) {
    if (kRuns && src.code_width != 0) {   // a compressed key stripe (block_runs.hpp): read as it lies
      coded_keys(src, k, [&](int r) {
        const int64_t row = src.base + static_cast<int64_t>(r) * blockDim.x + threadIdx.x;
        return row < src.n ? row : src.n - 1;
      });
    } else {
#pragma unroll
      for (int r = 0; r < R; ++r) {
        const int64_t row = src.base + static_cast<int64_t>(r) * blockDim.x + threadIdx.x;
        k[r] = __builtin_nontemporal_load(&src.keys[row < src.n ? row : src.n - 1]);   // clamped, not guarded
      }
    }
    words = ~0ull;
    if (src.filter != nullptr && lane < R) {
      const int64_t w = (src.base >> 6) + lane * waves_per_block + wave;
      if (w < ((src.n + 63) >> 6)) words = src.filter[w];
    }
  };
  Source cur = Source(), next = Source();
  if (static_cast<int64_t>(blockIdx.x) < num_tiles) {
    cur = source_of(blockIdx.x);
    request(cur, key, in_words);
  }
  for (int64_t tile = blockIdx.x; tile < num_tiles; tile += gridDim.x) {
    if (tile + gridDim.x < num_tiles) {
      next = source_of(tile + gridDim.x);
      request(next, next_key, next_in_words);
    }
    const int64_t tile_base = cur.base;
    const int64_t n = cur.n;
    uint64_t *const out_bitmap = cur.out_bitmap;
    const int64_t num_words = (n + 63) >> 6;
    cur = next;
    // bit position in 32 bits (filters have fewer than 2^32 bits: checked at creation); kNoBit = not representable
    constexpr uint32_t kNoBit = 0xFFFFFFFFu;
    uint32_t pos[R];
    uint32_t live_mask = 0;
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const int64_t row = tile_base + static_cast<int64_t>(r) * blockDim.x + threadIdx.x;
      const uint64_t in_word = __shfl(in_words, r, kWave);   // before any branch: every lane takes part
      const bool live = row < n && msb_bit(in_word, lane);
      live_mask |= live ? (1u << r) : 0u;
      const long long bit = live ? lip_bit_index(f, static_cast<long long>(key[r])) : -1;
      pos[r] = bit >= 0 ? static_cast<uint32_t>(bit) : kNoBit;
    }
    uint32_t word[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const uint32_t at = pos[r] != kNoBit ? (pos[r] >> 5) : 0u;   // unconditional read: a guarded one serialises the R reads
      word[r] = kInLds ? s_filter[at] : bits32[at];
    }
    uint64_t mine = 0;
    const bool out_of_range_hit = f.exact && f.is_anti != 0;   // outside the exact range: BitVectorExactFilter.hpp:158-172
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const bool set = (word[r] >> (pos[r] & 31)) & 1u;
      const bool hit = pos[r] == kNoBit ? out_of_range_hit : (f.is_anti ? !set : set);
      const uint64_t out = msb_first(__ballot(((live_mask >> r) & 1u) && hit));
      count += __popcll(out);
      if (lane == r) mine = out;
    }
    const int64_t w = (tile_base >> 6) + lane * waves_per_block + wave;
    if (lane < R && w < num_words) out_bitmap[w] = mine;
#pragma unroll
    for (int r = 0; r < R; ++r) key[r] = next_key[r];
    in_words = next_in_words;
  }
  if (out_count != nullptr) {
    if (lane != 0) count = 0;   // every lane counted the same wave-uniform ballots
    count = wave_reduce_add(count);
    if (lane == 0 && count != 0) atomicAdd(out_count, count);
  }
}

}  // namespace qsx

using namespace qsx;

struct qsx_lip_filter {
  int kind;
  long long cardinality;
  long long min_value;
  int is_anti;
  unsigned long long *words;
  long long num_words;
  LipView view() const {
    LipView v;
    v.words = words;
    v.cardinality = cardinality;
    v.min_value = min_value;
    v.exact = kind == QSX_LIP_BITVECTOR_EXACT ? 1 : 0;
    v.is_anti = is_anti;
    return v;
  }
};

namespace qsx {
LipView lip_filter_view(const qsx_lip_filter *f) { return f->view(); }
}  // namespace qsx

extern "C" {

int qsx_lip_filter_create(int kind, int64_t cardinality, int64_t min_value, int is_anti, qsx_lip_filter_t **out) {
  QSX_REQUIRE_DEVICE();
  if (out == nullptr || cardinality < 1) return QSX_ERR_INVALID_ARGUMENT;
  if (cardinality >= (1ll << 32) - 1) return QSX_ERR_UNSUPPORTED;   // bit positions are 32-bit in the probe kernel (512 MiB of filter)
  if (kind != QSX_LIP_SINGLE_IDENTITY_HASH && kind != QSX_LIP_BITVECTOR_EXACT) return QSX_ERR_UNSUPPORTED;
  if (kind == QSX_LIP_SINGLE_IDENTITY_HASH && is_anti) return QSX_ERR_UNSUPPORTED;
  qsx_lip_filter *f = new qsx_lip_filter();
  f->kind = kind;
  f->cardinality = cardinality;
  f->min_value = min_value;
  f->is_anti = is_anti ? 1 : 0;
  f->num_words = (cardinality + 63) / 64;
  hipError_t err = device_malloc(reinterpret_cast<void **>(&f->words), sizeof(unsigned long long) * f->num_words);
  if (err == hipSuccess) err = hipMemset(f->words, 0, sizeof(unsigned long long) * f->num_words);
  if (err == hipSuccess) err = hipDeviceSynchronize();
  if (err != hipSuccess) {
    set_last_error("qsx_lip_filter_create", err);
    delete f;
    return err == hipErrorOutOfMemory ? QSX_ERR_OUT_OF_MEMORY : QSX_ERR_HIP;
  }
  *out = f;
  return QSX_OK;
}

int qsx_lip_filter_destroy(qsx_lip_filter_t *f) {
  if (f == nullptr) return QSX_OK;
  (void)synchronize_owner_device(f->words);
  (void)device_free_idle(f->words);
  delete f;
  return QSX_OK;
}

int qsx_lip_build(qsx_lip_filter_t *f, int key_type, const void *keys_dev, int64_t n, const uint64_t *filter_dev,
                  qsx_stream_t stream) {
  QSX_REQUIRE_DEVICE();
  if (f == nullptr || n < 0 || (n > 0 && keys_dev == nullptr)) return QSX_ERR_INVALID_ARGUMENT;
  if (n == 0) return QSX_OK;
  const int grid = grid_for((n + 63) >> 6, (kLBlock / kWave) * 8);
  if (key_type == QSX_INT) {
    hipLaunchKernelGGL((lip_build_kernel<int32_t, 8>), dim3(grid), dim3(kLBlock), 0, as_stream(stream), f->view(),
                       static_cast<const int32_t *>(keys_dev), n, filter_dev);
  } else if (key_type == QSX_LONG) {
    hipLaunchKernelGGL((lip_build_kernel<int64_t, 8>), dim3(grid), dim3(kLBlock), 0, as_stream(stream), f->view(),
                       static_cast<const int64_t *>(keys_dev), n, filter_dev);
  } else {
    return QSX_ERR_UNSUPPORTED;
  }
  QSX_CHECK_LAUNCH();
  return QSX_OK;
}

int qsx_lip_probe(const qsx_lip_filter_t *f, int key_type, const void *keys_dev, int64_t n,
                  const uint64_t *in_bitmap_dev, uint64_t *out_bitmap_dev, int64_t *out_count_dev,
                  qsx_stream_t stream) {
  QSX_REQUIRE_DEVICE();
  if (f == nullptr || n < 0 || (n > 0 && (keys_dev == nullptr || out_bitmap_dev == nullptr))) return QSX_ERR_INVALID_ARGUMENT;
  hipStream_t s = as_stream(stream);
  if (out_count_dev != nullptr) QSX_HIP_TRY(hipMemsetAsync(out_count_dev, 0, sizeof(int64_t), s));
  if (n == 0) return QSX_OK;
  unsigned long long *count = reinterpret_cast<unsigned long long *>(out_count_dev);
  if (key_type != QSX_INT && key_type != QSX_LONG) return QSX_ERR_UNSUPPORTED;
  constexpr int R = 8;
  const long long words32 = (f->cardinality + 31) >> 5;
  // LDS copy pays when every workgroup amortises it over enough rows: one workgroup per CU, >= 64 K rows each
  const bool in_lds = words32 <= kLipLdsWords && n >= static_cast<int64_t>(kCUs) * 65536;
  if (in_lds) {
    const size_t lds = static_cast<size_t>(words32) * 4;
    const int grid = kCUs;
#define QSX_LIP_LAUNCH_LDS(KeyT)                                                                                      \
    do {                                                                                                              \
      static bool attribute_set = false;                                                                              \
      if (!attribute_set) {                                                                                           \
        QSX_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(&lip_probe_kernel<KeyT, R, true>),             \
                                        hipFuncAttributeMaxDynamicSharedMemorySize, kLipLdsWords * 4));              \
        attribute_set = true;                                                                                         \
      }                                                                                                               \
      hipLaunchKernelGGL((lip_probe_kernel<KeyT, R, true>), dim3(grid), dim3(1024), lds, s, f->view(),                \
                         static_cast<const KeyT *>(keys_dev), n, in_bitmap_dev, out_bitmap_dev, count);               \
    } while (0)
    if (key_type == QSX_INT) QSX_LIP_LAUNCH_LDS(int32_t); else QSX_LIP_LAUNCH_LDS(int64_t);
#undef QSX_LIP_LAUNCH_LDS
  } else {
    constexpr int RG = 8;    // 2048-row tiles, 8 workgroups per CU (4 and 8 rows per thread measure alike, 16 is 10 % slower)
    const int64_t tiles = (n + RG * kLBlock - 1) / (RG * kLBlock);
    const int grid = static_cast<int>(tiles < 8 * kCUs ? tiles : 8 * kCUs);
    if (key_type == QSX_INT) {
      hipLaunchKernelGGL((lip_probe_kernel<int32_t, RG, false>), dim3(grid), dim3(kLBlock), 0, s, f->view(),
                         static_cast<const int32_t *>(keys_dev), n, in_bitmap_dev, out_bitmap_dev, count);
    } else {
      hipLaunchKernelGGL((lip_probe_kernel<int64_t, RG, false>), dim3(grid), dim3(kLBlock), 0, s, f->view(),
                         static_cast<const int64_t *>(keys_dev), n, in_bitmap_dev, out_bitmap_dev, count);
    }
  }
  QSX_CHECK_LAUNCH();
  return QSX_OK;
}

// The table of a run (block_runs.hpp) on the device; *tiles = its tile count (0: nothing to do).
static int upload_lip_run(long long tile_rows, int64_t num_blocks, const int64_t *block_rows, const void *const *block_keys,
                          const uint64_t *const *block_in, uint64_t *const *block_out, hipStream_t s, const long long **runs_dev,
                          long long *tiles, const qsx_key_coding_t *coding = nullptr) {
  const int32_t *code_widths = coding != nullptr ? coding->block_code_width : nullptr;
  bool any_coded = false;
  for (int64_t b = 0; b < num_blocks && code_widths != nullptr; ++b) {
    const int w = code_widths[b];
    if (w != 0 && w != 1 && w != 2 && w != 4) return QSX_ERR_INVALID_ARGUMENT;
    if (w == 0 && coding->block_dictionaries != nullptr && coding->block_dictionaries[b] != nullptr) return QSX_ERR_INVALID_ARGUMENT;
    any_coded = any_coded || w != 0;
  }
  if (!any_coded) code_widths = nullptr;   // every stripe holds values: the plain run
  for (int64_t b = 0; b < num_blocks; ++b) {
    if (block_rows[b] < 0 || (block_rows[b] > 0 && (block_keys[b] == nullptr || (block_out != nullptr && block_out[b] == nullptr)))) {
      return QSX_ERR_INVALID_ARGUMENT;
    }
  }
  std::vector<long long> table;
  *tiles = build_run_table(tile_rows, num_blocks, block_rows, block_keys, reinterpret_cast<const void *const *>(block_in),
                           reinterpret_cast<void *const *>(block_out), nullptr, &table, code_widths,
                           code_widths != nullptr ? coding->block_dictionaries : nullptr);
  if (*tiles < 0) return QSX_ERR_INVALID_ARGUMENT;
  if (*tiles == 0) return QSX_OK;
  const size_t bytes = table.size() * sizeof(long long);
  *runs_dev = static_cast<const long long *>(staged_device_buffer(s, bytes));
  if (*runs_dev == nullptr) return QSX_ERR_OUT_OF_MEMORY;
  return staged_upload(s, table.data(), bytes);
}

static int lip_build_blocks_impl(qsx_lip_filter_t *f, int key_type, int64_t num_blocks, const int64_t *block_rows,
                                 const void *const *block_keys, const uint64_t *const *block_filters, const qsx_key_coding_t *coding,
                                 qsx_stream_t stream) {
  QSX_REQUIRE_DEVICE();
  if (f == nullptr || num_blocks < 0 || (num_blocks > 0 && (block_rows == nullptr || block_keys == nullptr))) return QSX_ERR_INVALID_ARGUMENT;
  if (key_type != QSX_INT && key_type != QSX_LONG) return QSX_ERR_UNSUPPORTED;
  hipStream_t s = as_stream(stream);
  const long long *runs_dev = nullptr;
  long long groups = 0;
  const int rc = upload_lip_run(8 * kWave, num_blocks, block_rows, block_keys, block_filters, nullptr, s, &runs_dev, &groups, coding);
  if (rc != QSX_OK || groups == 0) return rc;
  const int grid = grid_for(groups, kLBlock / kWave);
  if (key_type == QSX_INT) {
    hipLaunchKernelGGL((lip_build_kernel<int32_t, 8, true>), dim3(grid), dim3(kLBlock), 0, s, f->view(), static_cast<const int32_t *>(nullptr),
                       int64_t{0}, static_cast<const uint64_t *>(nullptr), runs_dev);
  } else {
    hipLaunchKernelGGL((lip_build_kernel<int64_t, 8, true>), dim3(grid), dim3(kLBlock), 0, s, f->view(), static_cast<const int64_t *>(nullptr),
                       int64_t{0}, static_cast<const uint64_t *>(nullptr), runs_dev);
  }
  QSX_CHECK_LAUNCH();
  return QSX_OK;
}

int qsx_lip_build_blocks(qsx_lip_filter_t *f, int key_type, int64_t num_blocks, const int64_t *block_rows,
                         const void *const *block_keys, const uint64_t *const *block_filters, qsx_stream_t stream) {
  return lip_build_blocks_impl(f, key_type, num_blocks, block_rows, block_keys, block_filters, nullptr, stream);
}
int qsx_lip_build_blocks_coded(qsx_lip_filter_t *f, int key_type, int64_t num_blocks, const int64_t *block_rows,
                               const void *const *block_keys, const qsx_key_coding_t *coding, const uint64_t *const *block_filters,
                               qsx_stream_t stream) {
  return lip_build_blocks_impl(f, key_type, num_blocks, block_rows, block_keys, block_filters, coding, stream);
}

static int lip_probe_blocks_impl(const qsx_lip_filter_t *f, int key_type, int64_t num_blocks, const int64_t *block_rows,
                                 const void *const *block_keys, const uint64_t *const *block_in_bitmaps, uint64_t *const *block_out_bitmaps,
                                 int64_t *out_count_dev, const qsx_key_coding_t *coding, qsx_stream_t stream) {
  QSX_REQUIRE_DEVICE();
  if (f == nullptr || num_blocks < 0 || (num_blocks > 0 && (block_rows == nullptr || block_keys == nullptr || block_out_bitmaps == nullptr))) {
    return QSX_ERR_INVALID_ARGUMENT;
  }
  if (key_type != QSX_INT && key_type != QSX_LONG) return QSX_ERR_UNSUPPORTED;
  hipStream_t s = as_stream(stream);
  if (out_count_dev != nullptr) QSX_HIP_TRY(hipMemsetAsync(out_count_dev, 0, sizeof(int64_t), s));
  unsigned long long *count = reinterpret_cast<unsigned long long *>(out_count_dev);
  int64_t total = 0;
  for (int64_t b = 0; b < num_blocks; ++b) total += block_rows[b] > 0 ? block_rows[b] : 0;
  constexpr int R = 8;
  const long long words32 = (f->cardinality + 31) >> 5;
  const bool in_lds = words32 <= kLipLdsWords && total >= static_cast<int64_t>(kCUs) * 65536;
  const long long *runs_dev = nullptr;
  long long tiles = 0;
  const int rc = upload_lip_run(static_cast<long long>(R) * (in_lds ? 1024 : kLBlock), num_blocks, block_rows, block_keys, block_in_bitmaps,
                                block_out_bitmaps, s, &runs_dev, &tiles, coding);
  if (rc != QSX_OK || tiles == 0) return rc;
  if (in_lds) {
    const size_t lds = static_cast<size_t>(words32) * 4;
#define QSX_LIP_LAUNCH_LDS_RUNS(KeyT)                                                                                 \
    do {                                                                                                              \
      static bool attribute_set = false;                                                                              \
      if (!attribute_set) {                                                                                           \
        QSX_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(&lip_probe_kernel<KeyT, R, true, true>),       \
                                        hipFuncAttributeMaxDynamicSharedMemorySize, kLipLdsWords * 4));              \
        attribute_set = true;                                                                                         \
      }                                                                                                               \
      hipLaunchKernelGGL((lip_probe_kernel<KeyT, R, true, true>), dim3(kCUs), dim3(1024), lds, s, f->view(),         \
                         static_cast<const KeyT *>(nullptr), int64_t{0}, static_cast<const uint64_t *>(nullptr),     \
                         static_cast<uint64_t *>(nullptr), count, runs_dev);                                         \
    } while (0)
    if (key_type == QSX_INT) QSX_LIP_LAUNCH_LDS_RUNS(int32_t); else QSX_LIP_LAUNCH_LDS_RUNS(int64_t);
#undef QSX_LIP_LAUNCH_LDS_RUNS
  } else {
    const int grid = static_cast<int>(tiles < 8 * kCUs ? tiles : 8 * kCUs);
    if (key_type == QSX_INT) {
      hipLaunchKernelGGL((lip_probe_kernel<int32_t, R, false, true>), dim3(grid), dim3(kLBlock), 0, s, f->view(),
                         static_cast<const int32_t *>(nullptr), int64_t{0}, static_cast<const uint64_t *>(nullptr),
                         static_cast<uint64_t *>(nullptr), count, runs_dev);
    } else {
      hipLaunchKernelGGL((lip_probe_kernel<int64_t, R, false, true>), dim3(grid), dim3(kLBlock), 0, s, f->view(),
                         static_cast<const int64_t *>(nullptr), int64_t{0}, static_cast<const uint64_t *>(nullptr),
                         static_cast<uint64_t *>(nullptr), count, runs_dev);
    }
  }
  QSX_CHECK_LAUNCH();
  return QSX_OK;
}
int qsx_lip_probe_blocks(const qsx_lip_filter_t *f, int key_type, int64_t num_blocks, const int64_t *block_rows,
                         const void *const *block_keys, const uint64_t *const *block_in_bitmaps, uint64_t *const *block_out_bitmaps,
                         int64_t *out_count_dev, qsx_stream_t stream) {
  return lip_probe_blocks_impl(f, key_type, num_blocks, block_rows, block_keys, block_in_bitmaps, block_out_bitmaps, out_count_dev, nullptr, stream);
}
int qsx_lip_probe_blocks_coded(const qsx_lip_filter_t *f, int key_type, int64_t num_blocks, const int64_t *block_rows,
                               const void *const *block_keys, const qsx_key_coding_t *coding, const uint64_t *const *block_in_bitmaps,
                               uint64_t *const *block_out_bitmaps, int64_t *out_count_dev, qsx_stream_t stream) {
  return lip_probe_blocks_impl(f, key_type, num_blocks, block_rows, block_keys, block_in_bitmaps, block_out_bitmaps, out_count_dev, coding, stream);
}

int qsx_lip_filter_words(qsx_lip_filter_t *f, uint64_t **out_words_dev, int64_t *out_num_words) {
  if (f == nullptr || out_words_dev == nullptr || out_num_words == nullptr) return QSX_ERR_INVALID_ARGUMENT;
  *out_words_dev = reinterpret_cast<uint64_t *>(f->words);
  *out_num_words = f->num_words;
  return QSX_OK;
}

}  // extern "C"
