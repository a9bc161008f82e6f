// join_sliced.hpp — K4 for join tables that do not fit one XCD's L2: the XCD-sliced, compacting probe.
//
// What bounds a probe (tools/ubench/gather_floor.hip, 100 M random lookups, MI355X): 0.42 ms while the table is L2-resident
// (<= 3.5 MiB: a CU's vector cache takes about one divergent lane per clock, whatever the entry width or the number of loads
// in flight), 0.65 ms at 6 MiB, 0.89 ms at 8 MiB, 1.31 ms at 16 MiB (fabric line fetches).  A hashed table of 1 M INT keys is
// 16 MiB, the directly addressed table of the broadcast join at 8 GPUs 32 MiB.
//
// So the table is cut into S slices (S = 2, 4 or 8; a slice <= ~3 MiB) and workgroup b serves slice b % S only: blocks are
// dealt to the XCDs round-robin, every XCD's L2 ends up holding one slice, every lookup hits it (placement matters for speed
// only, any placement gives the same result).  Every slice's workgroups stream ALL probe keys — coalesced 16-byte loads,
// the repeats served by L2 / MALL — and keep the keys of their slice:
//
//   scan     a chunk of 2048 keys per workgroup: per key "is it mine?" (a multiply and a compare), survivors appended to a
//            ring in LDS as (key or table position, row) through wave ballots + ONE LDS atomic per wave and chunk;
//   round    once the ring holds 4096 entries every thread takes 16 of them and issues its 16 table reads at once — full
//            waves of lookups, which is what the first sliced kernel (join_dense.hpp) lacked: with one lane in eight
//            active its eight times more tile visits each cost a memory round trip;
//   emit     as dense_probe_kernel: wave ballots, ONE reservation on the output counter per round of 4096 lookups, pairs
//            stored straight from registers as contiguous runs.
//
// Hashed table (join.hip: open addressing, 16-byte units): a round reads ONE unit per entry; an entry whose probe sequence
// goes on (no empty slot seen yet) is appended to the ring again with its displacement + 1 and takes part in a later round as
// an ordinary lane — no per-lane walks, no wave-aggregated atomics.  When the build saw no duplicate key (build_kernel
// compares the occupant's key on every failed claim) a probe stops at its first match.
#ifndef QSX_CSRC_JOIN_SLICED_HPP_
#define QSX_CSRC_JOIN_SLICED_HPP_

namespace qsx {

constexpr int kSBlock = 256;
constexpr int kSWaves = kSBlock / kWave;
constexpr int kSChunk = 2048;                        // keys scanned between two looks at the ring
// lookups per thread and round: P::kPerThread (dense 16: 4-byte reads; hashed 8: 16-byte units — registers); a round is
// 256 x that many entries, the ring holds a round + a chunk (a chunk always fits behind an unfinished round): 48 / 32 KiB.

template <typename KeyT>
struct KeyVec;
template <>
struct KeyVec<int32_t> {
  using Raw = int __attribute__((ext_vector_type(4)));
  static constexpr int K = 4;
};
template <>
struct KeyVec<int64_t> {
  using Raw = long long __attribute__((ext_vector_type(2)));
  static constexpr int K = 2;
};

// ---- what a slice of each table kind is ---------------------------------------------------------------------------------
// mine() runs for every (probe key, slice) pair — S times per probe row — so it is a handful of 32-bit instructions.
template <typename KeyT>
struct DenseSlices {
  using Key = KeyT;
  using Entry = uint32_t;                       // ring entry: the head index
  static constexpr bool kRequeue = false;
  static constexpr int kPerThread = 16;
  DenseTableView t;
  uint64_t lo = 0, width = 0;                   // this workgroup's slice of head[]
  uint32_t min32 = 0, stride_mask = 0;          // INT keys: (key - min) in 32 bits
  __device__ void select(int slice, int num_slices) {
    const uint64_t per = (t.range + num_slices - 1) / num_slices;
    lo = per * slice < t.range ? per * slice : t.range;
    width = (lo + per < t.range ? lo + per : t.range) - lo;
    min32 = static_cast<uint32_t>(t.min_key);
    stride_mask = (1u << t.stride_shift) - 1u;
  }
  __device__ __forceinline__ bool mine(Key key, Entry &e) const {
    if constexpr (sizeof(Key) == 4) {
      // an INT key below min_key wraps to >= 2^32 + (INT32_MIN - min_key) > max_key - min_key: never inside the range
      // (launch_sliced only takes INT tables whose min_key fits 32 bits)
      const uint32_t d = static_cast<uint32_t>(key) - min32;
      e = d >> t.stride_shift;
      return static_cast<uint64_t>(e) - lo < width && (d & stride_mask) == 0u;
    } else {
      const uint64_t idx = dense_index(t, key);   // ~0: not a member (~0 - lo >= width)
      e = static_cast<uint32_t>(idx);
      return idx - lo < width;
    }
  }
};

template <typename Units>
struct HashedSlices {
  using Key = typename Units::Key;
  using Entry = Key;                            // ring entry: the key (the unit follows from it and the displacement)
  static constexpr bool kRequeue = true;
  static constexpr int kPerThread = 8;
  TableView t;
  const unsigned int *dup_flag;                 // != 0: some build key occurs twice
  int slice_shift = 0;                          // first unit >> slice_shift = slice
  uint64_t mine_slice = 0;
  __device__ void select(int slice, int num_slices) {
    const Units table(t);
    int log_units = 0;
    while ((1ull << log_units) <= table.unit_mask) ++log_units;
    int log_s = 0;
    while ((1 << log_s) < num_slices) ++log_s;
    slice_shift = log_units - log_s;
    mine_slice = static_cast<uint64_t>(slice);
  }
  __device__ __forceinline__ Units units() const { return Units(t); }
  __device__ __forceinline__ bool mine(Key key, Entry &e) const {
    const Units table(t);
    e = key;
    return (table.first_unit(key, t) >> slice_shift) == mine_slice;
  }
};

// MODE 0: pairs, 1: count.
// Ring entry = (Entry, row word).  Hashed tables: the row word carries the entry's displacement from its home unit in the
// bits above the row number (32 - row_bits of them: 5 for 100 M rows); a walk beyond that finishes lane by lane.
template <typename P, int MODE>
__global__ __launch_bounds__(kSBlock) void sliced_probe_kernel(P policy, const typename P::Key *__restrict__ keys, int64_t n,
                                                              int32_t probe_base_tid, const uint64_t *__restrict__ filter,
                                                              int32_t *__restrict__ out_probe, int32_t *__restrict__ out_build,
                                                              int64_t capacity_signed, unsigned long long *__restrict__ out_count,
                                                              int num_slices, int row_bits) {
  using Key = typename P::Key;
  using Entry = typename P::Entry;
  using Vec = KeyVec<Key>;
  using Raw = typename Vec::Raw;
  constexpr int K = Vec::K;
  constexpr int V = kSChunk / (kSBlock * K);     // 16-byte vectors per thread and chunk
  constexpr int R = P::kPerThread;
  constexpr int kSRound = kSBlock * R;
  constexpr int kSRing = kSRound + kSChunk;
  constexpr unsigned int kRing = kSRing;
  const unsigned long long capacity = static_cast<unsigned long long>(capacity_signed);
  __shared__ Entry s_entry[kSRing];
  __shared__ uint32_t s_row[kSRing];
  __shared__ unsigned int s_tail;
  __shared__ int s_wave_total[2][kSWaves];
  __shared__ unsigned long long s_base;
  const int lane = lane_id();
  const int wave = threadIdx.x >> 6;
  const int slice = static_cast<int>(blockIdx.x % num_slices);
  const int64_t first = blockIdx.x / num_slices, step = gridDim.x / num_slices;
  const int64_t num_chunks = (n + kSChunk - 1) / kSChunk;
  const uint32_t row_mask = row_bits >= 32 ? ~0u : (1u << row_bits) - 1u;
  const uint32_t disp_max = row_bits >= 32 ? 0u : (~0u >> row_bits);
  policy.select(slice, num_slices);
  if (threadIdx.x == 0) s_tail = 0;
  __syncthreads();
  unsigned int head = 0;       // entries consumed so far (the same in every thread)
  unsigned int head_w = 0;     // head % kRing
  unsigned long long local_count = 0;
  int parity = 0;

  bool unique = false;
  if constexpr (P::kRequeue) unique = sizeof(Key) == 4 && *policy.dup_flag == 0u;   // (LONG keys: the build cannot tell)

  // ---- one round: `cnt` entries from `head` on, one table read each ------------------------------------------------------
  auto round = [&](unsigned int cnt) {
    Entry entry[R];
    uint32_t roww[R];
    uint32_t live = 0;
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const unsigned int i = r * kSBlock + threadIdx.x;
      unsigned int p = head_w + (i < cnt ? i : 0u);
      p = p >= kRing ? p - kRing : p;
      entry[r] = s_entry[p];
      roww[r] = s_row[p];
      live |= i < cnt ? (1u << r) : 0u;
    }
    if (P::kRequeue || MODE == 1) __syncthreads();   // every wave has read its entries: the ring may be written again
    if constexpr (!P::kRequeue) {
      // ---- directly addressed table: the structure of dense_probe_kernel -----------------------------------------------
      const DenseTableView &t = policy.t;
      uint32_t h[R];
#pragma unroll
      for (int r = 0; r < R; ++r) {
        const uint32_t word = dense_head_word(t, (live >> r) & 1u ? entry[r] : static_cast<uint32_t>(policy.lo));   // unconditional read
        h[r] = (live >> r) & 1u ? word : 0u;
      }
      int total = 0;
      bool any_chain = false;
#pragma unroll
      for (int r = 0; r < R; ++r) {
        total += __popcll(__ballot(h[r] != 0u));
        any_chain = any_chain || (h[r] & kChainBit) != 0u;
      }
      const bool wave_has_chain = __any(any_chain);
      uint32_t next[R];
      if (MODE == 1) {
        if (lane == 0) local_count += total;
        if (wave_has_chain) {
#pragma unroll
          for (int r = 0; r < R; ++r) next[r] = (h[r] & kChainBit) ? t.ov[h[r] & ~kChainBit].y : 0u;
        }
      } else {
        if (lane == 0) s_wave_total[parity][wave] = total;
        __syncthreads();
        if (threadIdx.x == 0) {
          int all = 0;
#pragma unroll
          for (int w = 0; w < kSWaves; ++w) all += s_wave_total[parity][w];
          s_base = all != 0 ? atomicAdd(out_count, static_cast<unsigned long long>(all)) : 0ull;
        }
        __syncthreads();
        unsigned long long base = s_base;
        for (int w = 0; w < wave; ++w) base += s_wave_total[parity][w];
#pragma unroll
        for (int r = 0; r < R; ++r) {
          const uint64_t m = __ballot(h[r] != 0u);
          uint32_t tid = h[r] - 1u;
          next[r] = 0u;
          if (wave_has_chain && (h[r] & kChainBit)) {
            const uint2 e = t.ov[h[r] & ~kChainBit];
            tid = e.x;
            next[r] = e.y;
          }
          const unsigned long long o = base + rank_below(m);
          if (h[r] != 0u && o < capacity) {
            __builtin_nontemporal_store(static_cast<int32_t>(probe_base_tid + roww[r]), &out_probe[o]);
            __builtin_nontemporal_store(static_cast<int32_t>(tid), &out_build[o]);
          }
          base += __popcll(m);
        }
        parity ^= 1;
      }
      if (wave_has_chain) {   // duplicate build keys: walk the chains (rare)
#pragma unroll 1
        for (int r = 0; r < R; ++r) {
          uint32_t cur = next[r];
          while (__any(cur != 0u)) {
            uint32_t tid = cur - 1u, nxt = 0u;
            if (cur & kChainBit) {
              const uint2 e = t.ov[cur & ~kChainBit];
              tid = e.x;
              nxt = e.y;
            }
            if (MODE == 1) {
              local_count += cur != 0u ? 1u : 0u;
            } else {
              dense_emit_direct(cur != 0u, static_cast<int32_t>(probe_base_tid + roww[r]), static_cast<int32_t>(tid), out_probe, out_build,
                                capacity, out_count);
            }
            cur = nxt;
          }
        }
      }
    } else {
      // ---- hashed table: one 16-byte unit per entry and round ------------------------------------------------------------
      using Units = decltype(policy.units());
      const Units table = policy.units();
      using RawUnit = typename Units::Raw;
      uint32_t m0 = 0, m1 = 0, cont = 0;
      int32_t t0[R], t1[R];
      int total = 0;
      auto unit_of = [&](int r) {
        const uint32_t disp = row_bits >= 32 ? 0u : roww[r] >> row_bits;
        return (table.first_unit(entry[r], policy.t) + disp) & table.unit_mask;
      };
      // two batches of R / 2 unit reads: the 16-byte units are only live until they are inspected (registers)
#pragma unroll
      for (int half = 0; half < 2; ++half) {
        RawUnit raw[R / 2];
#pragma unroll
        for (int q = 0; q < R / 2; ++q) {
          const int r = half * (R / 2) + q;
          raw[q] = table.load((live >> r) & 1u ? unit_of(r) : (policy.mine_slice << policy.slice_shift));
        }
#pragma unroll
        for (int q = 0; q < R / 2; ++q) {
          const int r = half * (R / 2) + q;
          const UnitHits hit = table.inspect(raw[q], entry[r]);
          const bool alive = (live >> r) & 1u;
          const bool a = alive && hit.m0, b = alive && hit.m1;
          m0 |= a ? (1u << r) : 0u;
          m1 |= b ? (1u << r) : 0u;
          t0[r] = hit.t0;
          t1[r] = hit.t1;
          // the probe sequence goes on unless an empty slot was seen — or the key was found and no build key occurs twice
          const bool go_on = alive && !hit.end && !(unique && (a || b));
          cont |= go_on ? (1u << r) : 0u;
          total += __popcll(__ballot(a)) + __popcll(__ballot(b));
        }
      }
      if (MODE == 1) {
        if (lane == 0) local_count += total;
      } else {
        if (lane == 0) s_wave_total[parity][wave] = total;
        __syncthreads();
        if (threadIdx.x == 0) {
          int all = 0;
#pragma unroll
          for (int w = 0; w < kSWaves; ++w) all += s_wave_total[parity][w];
          s_base = all != 0 ? atomicAdd(out_count, static_cast<unsigned long long>(all)) : 0ull;
        }
        __syncthreads();
        unsigned long long base = s_base;
        for (int w = 0; w < wave; ++w) base += s_wave_total[parity][w];
#pragma unroll
        for (int r = 0; r < R; ++r) {
          const bool a = (m0 >> r) & 1u, b = (m1 >> r) & 1u;
          const uint64_t ba = __ballot(a), bb = __ballot(b);
          const int32_t ptid = static_cast<int32_t>(probe_base_tid + (roww[r] & row_mask));
          const unsigned long long oa = base + rank_below(ba);
          if (a && oa < capacity) {
            __builtin_nontemporal_store(ptid, &out_probe[oa]);
            __builtin_nontemporal_store(t0[r], &out_build[oa]);
          }
          base += __popcll(ba);
          if (bb != 0) {   // wave-uniform
            const unsigned long long ob = base + rank_below(bb);
            if (b && ob < capacity) {
              __builtin_nontemporal_store(ptid, &out_probe[ob]);
              __builtin_nontemporal_store(t1[r], &out_build[ob]);
            }
            base += __popcll(bb);
          }
        }
        parity ^= 1;
      }
      // entries whose walk goes on: back into the ring, one unit further
      if (__any(cont != 0u)) {
#pragma unroll
        for (int r = 0; r < R; ++r) {
          bool go_on = (cont >> r) & 1u;
          const uint32_t disp = row_bits >= 32 ? 0u : roww[r] >> row_bits;
          // a walk longer than the displacement bits can say (hundreds of duplicates of one key) finishes here, lane by lane
          bool walking = go_on && disp >= disp_max;
          if (__any(walking)) {   // wave-uniform
            go_on = go_on && !walking;
            uint64_t u = unit_of(r);
            while (__any(walking)) {
              u = (u + 1) & table.unit_mask;
              const RawUnit next_raw = table.load(walking ? u : (policy.mine_slice << policy.slice_shift));
              const UnitHits hit = table.inspect(next_raw, entry[r]);
              const bool a = walking && hit.m0, b = walking && hit.m1;
              if (MODE == 1) {
                local_count += (a ? 1u : 0u) + (b ? 1u : 0u);
              } else {
                const int32_t ptid = static_cast<int32_t>(probe_base_tid + (roww[r] & row_mask));
                dense_emit_direct(a, ptid, hit.t0, out_probe, out_build, capacity, out_count);
                dense_emit_direct(b, ptid, hit.t1, out_probe, out_build, capacity, out_count);
              }
              if (hit.end || (unique && (a || b))) walking = false;
            }
          }
          const uint64_t bc = __ballot(go_on);
          if (bc == 0) continue;   // wave-uniform
          const int leader = __ffsll(static_cast<long long>(bc)) - 1;
          unsigned int at = 0;
          if (lane == leader) at = atomicAdd(&s_tail, static_cast<unsigned int>(__popcll(bc)));
          at = __builtin_amdgcn_readfirstlane(__shfl(at, leader, kWave)) % kRing;
          if (go_on) {
            unsigned int p = at + rank_below(bc);
            p = p >= kRing ? p - kRing : p;
            s_entry[p] = entry[r];
            s_row[p] = roww[r] + (1u << row_bits);   // displacement + 1
          }
        }
      }
      __syncthreads();   // requeued entries and the new tail are visible
    }
    head += cnt;
    head_w += cnt;
    head_w = head_w >= kRing ? head_w - kRing : head_w;
  };

  // ---- scan: two chunks of keys in flight behind the one being looked at ----------------------------------------------------
  Raw cur[V], nx1[V], nx2[V];
  uint64_t cur_fw[V], nx1_fw[V], nx2_fw[V];
  auto request = [&](int64_t chunk, Raw (&v)[V], uint64_t (&fw)[V]) {
#pragma unroll
    for (int i = 0; i < V; ++i) {
      const int64_t row0 = chunk * kSChunk + static_cast<int64_t>(i) * (kSBlock * K) + threadIdx.x * K;
      if (row0 + K <= n) {
        v[i] = *reinterpret_cast<const Raw *>(keys + row0);   // plain loads: the other slices' workgroups read the same lines
      } else {   // the last rows of the stripe: no read past its end
#pragma unroll
        for (int j = 0; j < K; ++j) v[i][j] = row0 + j < n ? keys[row0 + j] : Key(0);
      }
      fw[i] = ~0ull;
      if (filter != nullptr && row0 < n) fw[i] = filter[row0 >> 6];
    }
  };
  if (first < num_chunks) request(first, cur, cur_fw);
  if (first + step < num_chunks) request(first + step, nx1, nx1_fw);
  for (int64_t chunk = first; chunk < num_chunks; chunk += step) {
    if (chunk + 2 * step < num_chunks) request(chunk + 2 * step, nx2, nx2_fw);
    Entry e[V * K];
    uint32_t mine_mask = 0;
    int total = 0;
#pragma unroll
    for (int i = 0; i < V; ++i) {
      const int64_t row0 = chunk * kSChunk + static_cast<int64_t>(i) * (kSBlock * K) + threadIdx.x * K;
#pragma unroll
      for (int j = 0; j < K; ++j) {
        bool ok = policy.mine(static_cast<Key>(cur[i][j]), e[i * K + j]);
        ok = ok && row0 + j < n;
        if (filter != nullptr) ok = ok && ((cur_fw[i] >> (63 - ((row0 + j) & 63))) & 1u);
        mine_mask |= ok ? (1u << (i * K + j)) : 0u;
        total += __popcll(__ballot(ok));
      }
    }
    unsigned int at = 0;
    if (lane == 0 && total != 0) at = atomicAdd(&s_tail, static_cast<unsigned int>(total));
    at = __builtin_amdgcn_readfirstlane(at) % kRing;   // wave-uniform: scalar arithmetic
#pragma unroll
    for (int i = 0; i < V; ++i) {
      const uint32_t row0 = static_cast<uint32_t>(chunk * kSChunk + static_cast<int64_t>(i) * (kSBlock * K) + threadIdx.x * K);
#pragma unroll
      for (int j = 0; j < K; ++j) {
        const bool ok = (mine_mask >> (i * K + j)) & 1u;
        const uint64_t b = __ballot(ok);
        if (b == 0) continue;   // wave-uniform
        if (ok) {
          unsigned int p = at + rank_below(b);
          p = p >= kRing ? p - kRing : p;
          s_entry[p] = e[i * K + j];
          s_row[p] = row0 + j;
        }
        at += __popcll(b);
        at = at >= kRing ? at - kRing : at;
      }
    }
#pragma unroll
    for (int i = 0; i < V; ++i) {
      cur[i] = nx1[i];
      cur_fw[i] = nx1_fw[i];
      nx1[i] = nx2[i];
      nx1_fw[i] = nx2_fw[i];
    }
    __syncthreads();
    unsigned int tail = s_tail;
    while (tail - head >= static_cast<unsigned int>(kSRound)) {
      round(kSRound);
      tail = s_tail;   // (hashed: the round appended the walks that go on, behind a barrier)
    }
  }
  // ---- drain ---------------------------------------------------------------------------------------------------------------
  __syncthreads();
  unsigned int tail = s_tail;
  while (tail != head) {
    const unsigned int cnt = tail - head < static_cast<unsigned int>(kSRound) ? tail - head : static_cast<unsigned int>(kSRound);
    round(cnt);
    tail = s_tail;
  }
  if (MODE == 1) {
    local_count = wave_reduce_add(local_count);
    if (lane == 0 && local_count != 0) atomicAdd(out_count, local_count);
  }
}

}  // namespace qsx

#endif  // QSX_CSRC_JOIN_SLICED_HPP_
