// lip_view.hpp — what a kernel needs of a LIP filter (lip.hip owns the filters; join.hip's fused probe tests them too).
// Reference: utility/lip_filter/SingleIdentityHashFilter.hpp:156-169 (bit = value % cardinality, the value converted to
// size_t first), BitVectorExactFilter.hpp:150-176 (bit = value - min; outside [min, max]: a miss — a hit for an anti filter).
#ifndef QSX_CSRC_LIP_VIEW_HPP_
#define QSX_CSRC_LIP_VIEW_HPP_

#include "common.hpp"

namespace qsx {

struct LipView {
  unsigned long long *words;  // LSB-first bit array
  long long cardinality;
  long long min_value;
  int exact;
  int is_anti;
};

// Filter membership of one key: index of the filter bit, or -1 when the key is outside an exact
// filter's range (then `out_of_range_hit` decides, BitVectorExactFilter.hpp:158-172).
__device__ __forceinline__ long long lip_bit_index(const LipView &f, long long v) {
  if (f.exact) {
    const long long off = v - f.min_value;
    return (off < 0 || off >= f.cardinality) ? -1 : off;
  }
  return static_cast<long long>(static_cast<unsigned long long>(v) % static_cast<unsigned long long>(f.cardinality));
}

#ifndef __HIPCC_RTC__
// The view of a filter object (lip.hip).
LipView lip_filter_view(const struct ::qsx_lip_filter *f);
#endif

}  // namespace qsx

#endif  // QSX_CSRC_LIP_VIEW_HPP_
