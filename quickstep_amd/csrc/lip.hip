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

namespace qsx {

constexpr int kLBlock = 256;

struct LipView {
  unsigned long long *words;  // LSB-first bit array
  long long cardinality;
  long long min_value;
  int exact;
  int is_anti;
};

template <typename KeyT>
__global__ __launch_bounds__(kLBlock) void lip_build_kernel(LipView f, const KeyT *__restrict__ keys, int64_t n,
                                                           const uint64_t *__restrict__ filter) {
  for (int64_t i = static_cast<int64_t>(blockIdx.x) * kLBlock + threadIdx.x; i < n;
       i += static_cast<int64_t>(gridDim.x) * kLBlock) {
    if (filter != nullptr && !msb_bit(filter[i >> 6], static_cast<int>(i & 63))) continue;
    const long long v = static_cast<long long>(keys[i]);
    unsigned long long bit;
    if (f.exact) {
      const long long off = v - f.min_value;
      if (off < 0 || off >= f.cardinality) continue;  // outside the declared [min, max]: cannot be represented
      bit = static_cast<unsigned long long>(off);
    } else {
      // value converted to size_t first: a negative key sign-extends (SingleIdentityHashFilter.hpp:156-169)
      bit = static_cast<unsigned long long>(v) % static_cast<unsigned long long>(f.cardinality);
    }
    const unsigned long long mask = 1ull << (bit & 63);
    unsigned long long *w = &f.words[bit >> 6];
    if ((__hip_atomic_load(w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & mask) == 0) atomicOr(w, mask);
  }
}

template <typename KeyT>
__global__ __launch_bounds__(kLBlock) void lip_probe_kernel(LipView f, const KeyT *__restrict__ keys, int64_t n,
                                                           const uint64_t *__restrict__ in_bitmap,
                                                           uint64_t *__restrict__ out_bitmap,
                                                           unsigned long long *__restrict__ out_count) {
  const int lane = lane_id();
  const int64_t num_words = (n + 63) >> 6;
  unsigned long long count = 0;
  for (int64_t w = static_cast<int64_t>(blockIdx.x) * (kLBlock / kWave) + (threadIdx.x >> 6); w < num_words;
       w += static_cast<int64_t>(gridDim.x) * (kLBlock / kWave)) {
    const int64_t row = (w << 6) + lane;
    bool live = row < n;
    if (live && in_bitmap != nullptr) live = msb_bit(in_bitmap[w], lane);
    bool hit = false;
    if (live) {
      const long long v = static_cast<long long>(keys[row]);
      if (f.exact) {
        const long long off = v - f.min_value;
        if (off < 0 || off >= f.cardinality) {
          hit = f.is_anti != 0;  // BitVectorExactFilter.hpp:158-172
        } else {
          const bool set = (f.words[off >> 6] >> (off & 63)) & 1ull;
          hit = f.is_anti ? !set : set;
        }
      } else {
        const unsigned long long bit = static_cast<unsigned long long>(v) % static_cast<unsigned long long>(f.cardinality);
        hit = (f.words[bit >> 6] >> (bit & 63)) & 1ull;
      }
    }
    const uint64_t word = msb_first(__ballot(live && hit));
    if (lane == 0) {
      out_bitmap[w] = word;
      count += __popcll(word);
    }
  }
  if (out_count != nullptr) {
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

extern "C" {

int qsx_lip_filter_create(int kind, int64_t cardinality, int64_t min_value, int is_anti, qsx_lip_filter_t **out) {
  QSX_REQUIRE_DEVICE();
  if (out == nullptr || cardinality < 1) return QSX_ERR_INVALID_ARGUMENT;
  if (kind != QSX_LIP_SINGLE_IDENTITY_HASH && kind != QSX_LIP_BITVECTOR_EXACT) return QSX_ERR_UNSUPPORTED;
  if (kind == QSX_LIP_SINGLE_IDENTITY_HASH && is_anti) return QSX_ERR_UNSUPPORTED;
  qsx_lip_filter *f = new qsx_lip_filter();
  f->kind = kind;
  f->cardinality = cardinality;
  f->min_value = min_value;
  f->is_anti = is_anti ? 1 : 0;
  f->num_words = (cardinality + 63) / 64;
  hipError_t err = hipMalloc(reinterpret_cast<void **>(&f->words), sizeof(unsigned long long) * f->num_words);
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
  (void)hipDeviceSynchronize();
  (void)hipFree(f->words);
  delete f;
  return QSX_OK;
}

int qsx_lip_build(qsx_lip_filter_t *f, int key_type, const void *keys_dev, int64_t n, const uint64_t *filter_dev,
                  qsx_stream_t stream) {
  QSX_REQUIRE_DEVICE();
  if (f == nullptr || n < 0 || (n > 0 && keys_dev == nullptr)) return QSX_ERR_INVALID_ARGUMENT;
  if (n == 0) return QSX_OK;
  const int grid = grid_for(n, kLBlock * 4);
  if (key_type == QSX_INT) {
    hipLaunchKernelGGL(lip_build_kernel<int32_t>, dim3(grid), dim3(kLBlock), 0, as_stream(stream), f->view(),
                       static_cast<const int32_t *>(keys_dev), n, filter_dev);
  } else if (key_type == QSX_LONG) {
    hipLaunchKernelGGL(lip_build_kernel<int64_t>, dim3(grid), dim3(kLBlock), 0, as_stream(stream), f->view(),
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
  const int64_t num_words = (n + 63) >> 6;
  const int grid = grid_for(num_words, (kLBlock / kWave) * 4);
  unsigned long long *count = reinterpret_cast<unsigned long long *>(out_count_dev);
  if (key_type == QSX_INT) {
    hipLaunchKernelGGL(lip_probe_kernel<int32_t>, dim3(grid), dim3(kLBlock), 0, s, f->view(),
                       static_cast<const int32_t *>(keys_dev), n, in_bitmap_dev, out_bitmap_dev, count);
  } else if (key_type == QSX_LONG) {
    hipLaunchKernelGGL(lip_probe_kernel<int64_t>, dim3(grid), dim3(kLBlock), 0, s, f->view(),
                       static_cast<const int64_t *>(keys_dev), n, in_bitmap_dev, out_bitmap_dev, count);
  } else {
    return QSX_ERR_UNSUPPORTED;
  }
  QSX_CHECK_LAUNCH();
  return QSX_OK;
}

int qsx_lip_filter_words(qsx_lip_filter_t *f, uint64_t **out_words_dev, int64_t *out_num_words) {
  if (f == nullptr || out_words_dev == nullptr || out_num_words == nullptr) return QSX_ERR_INVALID_ARGUMENT;
  *out_words_dev = reinterpret_cast<uint64_t *>(f->words);
  *out_num_words = f->num_words;
  return QSX_OK;
}

}  // extern "C"
