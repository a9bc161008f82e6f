// agg_common.hpp — device-side descriptors and per-row helpers shared by the
// aggregation kernels (aggregate.hip, agg_hash_update.hpp).
#ifndef QSX_CSRC_AGG_COMMON_HPP_
#define QSX_CSRC_AGG_COMMON_HPP_

#include "common.hpp"

namespace qsx {

constexpr int kABlock = 256;
constexpr uint64_t kEmptyCode = ~0ull;
constexpr int kMaxSums = QSX_MAX_AGGS;

struct DevOperand {
  int kind;
  int index;
};
// Internal operand kind (never in a qsx_agg_config_t): word `index` of a wide group-by key, see DevConfig::wide_words.
constexpr int kOpdKeyWord = 100;
constexpr int kMaxKeyWords = 3;
struct DevInstr {
  int op;
  int dst;
  DevOperand a, b;
};
// How an accumulator word combines (state columns hold 8-byte words).  MIN/MAX always work on
// int64: an INT/LONG argument as it is, a FLOAT/DOUBLE/expression argument through the
// order-preserving map of ordered_from_double(), so that one ds_min_i64 / global_atomic_smin_x2
// serves every type.
enum AccKind { kAccSumF64 = 0, kAccSumI64 = 1, kAccMinI64 = 2, kAccMaxI64 = 3 };

struct DevSum {
  DevOperand arg;
  int is_int;   // the argument is an INT/LONG column (read as integer)
  int kind;     // AccKind
  // Nullable inputs: bit s set = the argument is NULL when null slot s (DevConfig::null_column) is NULL in the row; the
  // row then contributes the accumulator's identity.  count_valid: this accumulator counts the rows whose mask is
  // clear (COUNT(x), the denominator of AVG(x), the "saw a value" test of SUM / MIN / MAX) instead of adding `arg`.
  unsigned null_mask;
  int count_valid;
};

// double <-> int64 with the same ordering (IEEE sign-magnitude -> two's complement)
__host__ __device__ constexpr long long ordered_from_bits(long long bits) {
  return bits < 0 ? bits ^ 0x7FFFFFFFFFFFFFFFll : bits;
}
__host__ __device__ constexpr long long acc_identity(int kind) {
  return kind == kAccMinI64 ? 0x7FFFFFFFFFFFFFFFll : (kind == kAccMaxI64 ? static_cast<long long>(0x8000000000000000ull) : 0ll);
}
constexpr int kRegDecoded = -2;   // DevConfig::lds_off of a compressed attribute whose values live in registers (plan_tile)
struct DevPred {
  int column;
  int op;
  unsigned long long literal;  // raw bits, typed like the column
};

// Interpreter plan (run-time configurations only; AOT plan shapes keep their temps in registers):
// operands pre-resolved to LDS byte offsets so that an interpreted instruction is one scalar
// decode + V ds_read_b64 per operand + V ds_write_b64, with no per-row type dispatch and no VGPR
// array of temps (which capped the interpreter at V = 2 rows per thread).
enum PlanMode { kPlanTileF64 = 0, kPlanTempF64 = 1, kPlanImm = 2, kPlanTileI32 = 3, kPlanTileI64 = 4, kPlanTileF32 = 5 };
struct PlanOperand {
  int mode;    // PlanMode
  int off;     // byte offset inside the staged tile / the temps area
  double imm;
};
struct PlanInstr {
  int op;
  int dst_off;  // byte offset of the result's slot in the temps area; -1: result never read
  PlanOperand a, b;
};
struct PlanSum {
  int is_int;   // integer column argument: read `width` bytes at tile + arg.off
  int width;
  PlanOperand arg;
};

struct DevConfig {
  int num_columns;
  int column_type[QSX_MAX_COLUMNS];
  int column_width[QSX_MAX_COLUMNS];
  int num_keys;
  int key_column[QSX_MAX_KEYS];
  int key_width[QSX_MAX_KEYS];
  int key_shift[QSX_MAX_KEYS];  // bit offset of the key inside its 64-bit word (the code itself unless wide_words != 0)
  // Group-by keys wider than 8 packed bytes (PackedPayloadHashTable takes any composite key,
  // storage/PackedPayloadHashTable.hpp:499-521): the components are packed into wide_words 64-bit words (key k in word
  // key_word[k]; a component never straddles two words), the table is keyed by a 64-bit mixing hash of the words, and
  // every word gets two hidden accumulators, MIN and MAX of the word over the group's rows.  Finalize reads the key back
  // from the MIN columns; MIN == MAX for every word of every group proves that no two keys shared a hash (anything else
  // is reported, never returned as a result).  wide_hash_mask: test hook that forces such collisions.
  int wide_words;
  int key_word[QSX_MAX_KEYS];
  unsigned long long wide_hash_mask;
  int num_instrs;
  DevInstr instrs[QSX_MAX_INSTRS];
  double consts[QSX_MAX_CONSTS];
  int num_sums;
  DevSum sums[kMaxSums];
  int num_pred;
  DevPred pred[QSX_MAX_PRED_TERMS];
  const void *cols[QSX_MAX_COLUMNS];
  // Compressed attributes (CompressedColumnStore): column c arrives as a stripe of code_width[c]-byte unsigned codes
  // (0 = plain values).  The codes are staged like any column (same DMA, code_off[c]); once the tile has landed
  // every thread decodes the codes of ITS rows into the column's value slots (lds_off[c]) — through dicts[c]
  // (dictionary of column_type values; per block, so per call) or, when that is null, as the value itself
  // (truncation).  HBM moves the codes, the value slots look like a plain column to everything downstream.
  int code_width[QSX_MAX_COLUMNS];
  const void *dicts[QSX_MAX_COLUMNS];
  int code_off[QSX_MAX_COLUMNS];   // byte offset of the staged codes inside a tile (-1: plain column)
  // LDS staging plan of the hash-strategy update kernel: byte offset of column
  // c inside a staged tile (-1: column not referenced, not staged), of the
  // filter words (-1: no filter) and the size of one tile buffer.
  int lds_off[QSX_MAX_COLUMNS];   // (-1: column not referenced; kRegDecoded: a compressed attribute decoded into registers)
  int filter_lds_off;
  int tile_bytes;
  // Nullable columns (qsx_agg_config_t::column_nullable) that the plan reads: slot s = column null_column[s]; its
  // null words of a tile are staged at null_lds_off[s] like the filter words.  row_null_mask: slots of the group-by key
  // and predicate columns — a row with one of them NULL is not aggregated at all.
  int num_null_cols;
  int null_column[QSX_MAX_COLUMNS];
  int null_lds_off[QSX_MAX_COLUMNS];
  unsigned row_null_mask;
  const unsigned long long *nulls[QSX_MAX_COLUMNS];   // per call, by slot; nullptr = no NULL in this block
  // interpreter plan (plan_interpreter() in aggregate.hip); temps_bytes == 0 for AOT shapes
  int temps_bytes;
  PlanInstr plan_instrs[QSX_MAX_INSTRS];
  PlanSum plan_sums[kMaxSums];
};

struct HashTableView {
  unsigned long long *keys;    // [cap + 1]
  unsigned long long *states;  // [(NS + 1)][cap + 1]
  unsigned long long cap;      // power of two
  int shift;                   // 64 - log2(cap)
  unsigned long long *ngroups; // groups inserted (sentinel slot not counted)
  int *overflow;               // set when a group found room neither in the table nor in the spill log
  // Spill log (growable states; log_cap == 0 otherwise): a group whose probe sequence is exhausted — the optimizer's
  // estimate was too low and the table has not been grown yet — gets a private record {code, col 0 .. col NC-1} here
  // instead of being dropped.  The host grows the table and folds the records back in before the next launch that sees
  // them and before every finalize / export (aggregate.hip: settle); the reference resizes in place under an exclusive
  // lock (storage/PackedPayloadHashTable.cpp:232-288, ThreadPrivateCompactKeyHashTable.cpp:159-201).
  unsigned long long *log;     // [log_cap][log_stride], every record pre-initialised with the columns' identities
  unsigned int *log_count;
  unsigned int log_cap;
  int log_stride;              // NC + 1 words
};

struct DenseView {
  unsigned long long *exist;
  unsigned long long *states;  // [ncols][E]
  long long num_entries;
  int has_count;               // col 0 is the row count
  int *error;                  // set when a key is outside [0, E)
};

// Combine two accumulator words.
__device__ __forceinline__ unsigned long long acc_combine(unsigned long long acc, unsigned long long inc, int kind) {
  switch (kind) {
    case kAccSumI64: return acc + inc;
    case kAccMinI64: return static_cast<long long>(inc) < static_cast<long long>(acc) ? inc : acc;
    case kAccMaxI64: return static_cast<long long>(inc) > static_cast<long long>(acc) ? inc : acc;
    default:
      return static_cast<unsigned long long>(__double_as_longlong(
          __longlong_as_double(static_cast<long long>(acc)) + __longlong_as_double(static_cast<long long>(inc))));
  }
}

// ---- global hash table --------------------------------------------------------
__device__ __forceinline__ unsigned long long code_slot(unsigned long long code, int shift) {
  return (mix64(code) * 0x9E3779B97F4A7C15ull) >> shift;
}

// Probing is bounded: below the load the host keeps the table at (<= 1/4 once it has seen the group count, aggregate.hip)
// a run of kGlobalMaxProbes occupied slots does not happen, and a table that filled up behind the host's back costs a
// miss 128 reads instead of cap.  A code that fails once fails every time (slots never free up), so its rows
// consistently go to the spill log.
constexpr unsigned long long kGlobalMaxProbes = 128;

// Returns where `code` accumulates: its slot (inserting it if new), cap for the sentinel code, cap + 1 + r for
// a fresh record r of the spill log, or ~0 when neither has room (overflow flag raised).
__device__ __forceinline__ unsigned long long global_find_or_insert(const HashTableView &g, unsigned long long code) {
  if (code == kEmptyCode) return g.cap;
  unsigned long long s = code_slot(code, g.shift);
  const unsigned long long limit = g.cap < kGlobalMaxProbes ? g.cap : kGlobalMaxProbes;
  for (unsigned long long probes = 0; probes < limit; ++probes) {
    unsigned long long k = __hip_atomic_load(&g.keys[s], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (k == code) return s;
    if (k == kEmptyCode) {
      k = atomicCAS(&g.keys[s], kEmptyCode, code);
      if (k == kEmptyCode) {
        atomicAdd(g.ngroups, 1ull);
        return s;
      }
      if (k == code) return s;
    }
    s = (s + 1) & (g.cap - 1);
  }
  if (g.log_cap != 0) {
    const unsigned int r = atomicAdd(g.log_count, 1u);
    if (r < g.log_cap) {
      g.log[static_cast<unsigned long long>(r) * g.log_stride] = code;
      return g.cap + 1 + r;
    }
  }
  atomicExch(g.overflow, 1);
  return ~0ull;
}

// One 64-bit global atomic on an accumulator word.
__device__ __forceinline__ void global_accumulate(unsigned long long *p, unsigned long long inc, int kind) {
  switch (kind) {
    case kAccSumI64:
      if (inc != 0) atomicAdd(p, inc);
      break;
    // MIN / MAX only ever move one way: a value that cannot move the accumulator any more is not sent (a stale read errs
    // on the side of sending it).  After a group's first rows that is nearly every row — and every row of a wide key's
    // hidden MIN / MAX proof, whose value is the same word each time: random global atomics complete at 24 G/s, reads of an
    // L2-resident line ten times faster (16-byte key, 10^5 groups: 25.5 -> 9 ms per 100 M rows).
    case kAccMinI64:
      if (static_cast<long long>(__hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) > static_cast<long long>(inc)) {
        atomicMin(reinterpret_cast<long long *>(p), static_cast<long long>(inc));
      }
      break;
    case kAccMaxI64:
      if (static_cast<long long>(__hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) < static_cast<long long>(inc)) {
        atomicMax(reinterpret_cast<long long *>(p), static_cast<long long>(inc));
      }
      break;
    default: atomic_add_f64(reinterpret_cast<double *>(p), __longlong_as_double(static_cast<long long>(inc))); break;
  }
}
__device__ __forceinline__ void global_add(const HashTableView &g, int col, unsigned long long slot,
                                           unsigned long long inc, int kind) {
  unsigned long long *p = slot <= g.cap ? g.states + static_cast<unsigned long long>(col) * (g.cap + 1) + slot
                                        : g.log + (slot - g.cap - 1) * g.log_stride + 1 + col;
  global_accumulate(as_global(p), inc, kind);   // (a select of two pointers is generic to the compiler: flat atomics otherwise)
}

// ---- group directory (mid-size group counts) ---------------------------------------
// Thousands of groups do not fit a replicated workgroup-private LDS *hash table* (keys + states), but their accumulators
// alone do when they are addressed by a dense group number: 10 k groups x (4-byte count + one 8-byte sum) = 120 KiB.
// The key -> group number mapping lives in this directory in HBM — 16-byte entries {key code, gid}, a few hundred KiB,
// L2-resident — and is read-mostly: after the first rows of an input every lookup is one 16-byte read.  gids are handed
// out in order of first insertion and stay valid for the life of the state, so every workgroup of every update call
// agrees on them.  The directory only accelerates: the global table (HashTableView) stays the state's content, rows
// whose group gets no gid below lds_gids take the per-row global path.
struct DirView {
  unsigned long long *entries;        // [dcap][2]: {code (kEmptyCode = free), gid (kEmptyCode = not published yet)}
  unsigned long long dmask;           // dcap - 1
  int dshift;                         // 64 - log2(dcap)
  unsigned long long *codes_by_gid;   // [lds_gids]
  unsigned int *ngids;
  unsigned int lds_gids;              // gids with an LDS accumulator
  // Key bounds of the build pass's rows (zeroed before every build pass): bounds[2k] = max of the k-th key's
  // order-preserving unsigned image, bounds[2k + 1] = max of that image's complement (i.e. ~min) — both grow from 0, so
  // one memset resets them and one atomicMax per wave updates them.  When the box they span has at most lds_gids cells the
  // accumulate pass numbers the groups by their position in the box (no directory read at all); rows outside the box —
  // the build pass only samples the input — take the per-row global path like rows without a gid.
  unsigned long long *bounds;         // [2 * QSX_MAX_KEYS]
  // The build pass reads every sample_stride-th tile, starting with tile sample_phase: a group that a sample of several
  // million rows misses has few rows, and those are aggregated through the global table.
  int sample_stride;
  int sample_phase;
  // The build pass is two launches over the same sample: 0 = key bounds only; 1 = the directory itself — which returns at
  // once when the bounds span a usable box (the accumulate pass will not look anything up).
  int build_step;
  // Wide keys (DevConfig::wide_words != 0): entries are four words {tag | gid, word 0, word 1, word 2} — tag = the upper
  // half of the key's hash code, gid in the lower half — and the key words of gid g are words_by_gid[3 g ..].  A lookup
  // compares all key words, so a gid's accumulators only ever see rows of ITS key: the hidden MIN / MAX accumulators that
  // prove a wide key elsewhere (DevConfig::wide_words) have no LDS planes in directory mode, the flush writes the key words
  // into those state columns once per group and workgroup.
  int wide_words;
  unsigned long long *words_by_gid;   // [lds_gids][kMaxKeyWords]
};
constexpr unsigned long long kSignBias = 1ull << 63;

// 64-bit mixing hash of the words of a wide key: the code the tables are keyed by (DevConfig::wide_words).
__device__ __forceinline__ unsigned long long wide_key_code(const unsigned long long (&w)[kMaxKeyWords], int wide_words,
                                                            unsigned long long hash_mask) {
  unsigned long long h = 0x9E3779B97F4A7C15ull;
#pragma unroll
  for (int i = 0; i < kMaxKeyWords; ++i) {
    if (i < wide_words) h = mix64(h ^ w[i]) * 0xD6E8FEB86659FD93ull + 0x2545F4914F6CDD1Dull;
  }
  return mix64(h) & hash_mask;
}

// Position of a row's group inside the key box of the build pass (dense numbering).
struct KeyBox {
  bool usable;                         // the box has at most lds_gids cells
  int cells;
  long long lo[QSX_MAX_KEYS];
  unsigned long long range[QSX_MAX_KEYS];
  unsigned int mult[QSX_MAX_KEYS];
};
__device__ __forceinline__ KeyBox key_box_of(const DirView &d, int num_keys) {
  KeyBox b;
  b.usable = d.bounds != nullptr && num_keys > 0;
  unsigned long long cells = 1;
#pragma unroll
  for (int k = 0; k < QSX_MAX_KEYS; ++k) {
    b.lo[k] = 0;
    b.range[k] = 1;
    b.mult[k] = 0;
    if (k < num_keys && b.usable) {
      const unsigned long long hi_image = d.bounds[2 * k];
      const unsigned long long lo_image = ~d.bounds[2 * k + 1];
      // an empty sample leaves both words 0: lo_image = ~0 > hi_image
      if (lo_image > hi_image || hi_image - lo_image >= d.lds_gids) {
        b.usable = false;
      } else {
        b.lo[k] = static_cast<long long>(lo_image ^ kSignBias);
        b.range[k] = hi_image - lo_image + 1;
        b.mult[k] = static_cast<unsigned int>(cells);
        cells *= b.range[k];
        if (cells > d.lds_gids) b.usable = false;
      }
    }
  }
  b.cells = b.usable ? static_cast<int>(cells) : 0;
  return b;
}
constexpr int kDirMaxProbes = 64;

// Within one update call the directory is written by one kernel and read by the next: on a multi-XCD part a
// device-coherent load (agent scope, sc1) bypasses the XCD's L2 — measured 59 ms per 200 M lookups against 1 ms for
// plain loads — and a plain load may return a stale line while another XCD inserts.  So an update call is two launches:
// the build pass collects the distinct key codes of its rows in a workgroup-private LDS set (agg_hash_update_body,
// kDirBuild) and inserts them here with device-scope atomics — a few thousand per workgroup, not one per row — and after
// the kernel boundary (which makes the directory visible to every L2) the accumulate pass only reads it.
// Update calls of other streams may overlap either pass.  That costs speed, not rows: an entry is never changed once its
// gid is published, a reader that sees it half-written or not at all (stale line) sends the row down the global path,
// and the flush fetches a gid's code with a device-scope load behind the release store that published the gid.

// Insert `code` if new (device-scope atomics; called per distinct code of a workgroup, not per row).
__device__ __forceinline__ void dir_insert(const DirView &d, unsigned long long code) {
  unsigned long long s = (mix64(code) * 0x9E3779B97F4A7C15ull) >> d.dshift;
  for (int probes = 0; probes < kDirMaxProbes; ++probes) {
    unsigned long long *entry = d.entries + 2 * s;
    unsigned long long k = __hip_atomic_load(entry, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (k == kEmptyCode) {
      k = atomicCAS(entry, kEmptyCode, code);
      if (k == kEmptyCode) {
        // the gid is read by the NEXT kernel only: no publication protocol inside this one
        const unsigned int g = atomicAdd(d.ngids, 1u);
        if (g < d.lds_gids) __hip_atomic_store(&d.codes_by_gid[g], code, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        // release: whoever sees the gid (an accumulate pass of ANOTHER stream's call may run next to this build pass) finds
        // the code behind it with a device-scope load
        __hip_atomic_store(entry + 1, static_cast<unsigned long long>(g), __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        return;
      }
    }
    if (k == code) return;
    s = (s + 1) & d.dmask;
  }
  // no room near the code's home: its rows take the global path in the accumulate pass
}

// gid of `code` (plain cached loads: the directory is read-only in this launch), or -1: not in the directory
// (full around its home) or without an LDS accumulator.  The lookup comes in two halves so that a caller can put other
// memory operations between the read of the home entry and its use (agg_hash_update.hpp: the next tile's DMA).
struct DirProbe {
  unsigned long long s;
  ulonglong2 a, b;   // the entry at s (b: words 2, 3 of a wide key's entry)
};
__device__ __forceinline__ DirProbe dir_first_probe(const DirView &d, unsigned long long code) {
  DirProbe p;
  p.s = (mix64(code) * 0x9E3779B97F4A7C15ull) >> d.dshift;
  if (d.wide_words != 0) {
    const ulonglong2 *entry = reinterpret_cast<const ulonglong2 *>(d.entries + 4 * p.s);
    p.a = entry[0];
    p.b = entry[1];
  } else {
    p.a = *reinterpret_cast<const ulonglong2 *>(d.entries + 2 * p.s);
    p.b = ulonglong2{0, 0};
  }
  return p;
}
__device__ __forceinline__ int dir_lookup_from(const DirView &d, unsigned long long code, DirProbe p) {
  for (int probes = 1;; ++probes) {
    if (p.a.x == code) return p.a.y < d.lds_gids ? static_cast<int>(p.a.y) : -1;
    if (p.a.x == kEmptyCode || probes >= kDirMaxProbes) return -1;
    p.s = (p.s + 1) & d.dmask;
    p.a = *reinterpret_cast<const ulonglong2 *>(d.entries + 2 * p.s);
  }
}
__device__ __forceinline__ int dir_lookup(const DirView &d, unsigned long long code) {
  return dir_lookup_from(d, code, dir_first_probe(d, code));
}

// Wide keys.  The directory only accelerates — a key that is not in it is aggregated through the global table, which is
// always right — so an INSERT may take "an entry with my tag is there" for "my key is there" (another key sharing the 32-bit
// tag merely stays out), while a LOOKUP answers with a gid only when every key word matches.
constexpr unsigned long long kDirTagMask = 0xFFFFFFFF00000000ull;
constexpr unsigned long long kDirGidPending = 0xFFFFFFFEull;   // claimed, gid not published yet (never < lds_gids)
__device__ __forceinline__ void dir_insert_wide(const DirView &d, unsigned long long code, const unsigned long long (&w)[kMaxKeyWords]) {
  const unsigned long long tag = code & kDirTagMask;
  unsigned long long s = (mix64(code) * 0x9E3779B97F4A7C15ull) >> d.dshift;
  for (int probes = 0; probes < kDirMaxProbes; ++probes) {
    unsigned long long *entry = d.entries + 4 * s;
    unsigned long long k = __hip_atomic_load(entry, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (k == kEmptyCode) {
      k = atomicCAS(entry, kEmptyCode, tag | kDirGidPending);
      if (k == kEmptyCode) {
        unsigned int g = atomicAdd(d.ngids, 1u);
        if (g > kDirGidPending - 1) g = static_cast<unsigned int>(kDirGidPending - 1);
#pragma unroll
        for (int i = 0; i < kMaxKeyWords; ++i) {
          __hip_atomic_store(entry + 1 + i, w[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          if (g < d.lds_gids) {
            __hip_atomic_store(&d.words_by_gid[static_cast<size_t>(g) * kMaxKeyWords + i], w[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          }
        }
        __hip_atomic_store(entry, tag | g, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        return;
      }
    }
    if ((k & kDirTagMask) == tag) return;
    s = (s + 1) & d.dmask;
  }
}
// (one aligned 32-byte read per probe; an entry seen half-written — the words still 0xFF.. behind a published gid cannot
// happen in program order, but the two halves of the read are not ordered — fails the comparison: global path)
__device__ __forceinline__ int dir_lookup_wide_from(const DirView &d, unsigned long long code, const unsigned long long (&w)[kMaxKeyWords],
                                                    DirProbe p) {
  const unsigned long long tag = code & kDirTagMask;
  for (int probes = 1;; ++probes) {
    if (p.a.x == kEmptyCode) return -1;
    if ((p.a.x & kDirTagMask) == tag) {
      const unsigned long long gid = p.a.x & ~kDirTagMask;
      return (gid < d.lds_gids && p.a.y == w[0] && p.b.x == w[1] && p.b.y == w[2]) ? static_cast<int>(gid) : -1;
    }
    if (probes >= kDirMaxProbes) return -1;
    p.s = (p.s + 1) & d.dmask;
    const ulonglong2 *entry = reinterpret_cast<const ulonglong2 *>(d.entries + 4 * p.s);
    p.a = entry[0];
    p.b = entry[1];
  }
}
__device__ __forceinline__ int dir_lookup_wide(const DirView &d, unsigned long long code, const unsigned long long (&w)[kMaxKeyWords]) {
  return dir_lookup_wide_from(d, code, w, dir_first_probe(d, code));
}

// ---- LDS table ------------------------------------------------------------------
// Probing is bounded (a group lives within kLdsMaxProbes slots of its home or not in LDS at
// all), so that a full table costs a miss 8 LDS reads, not S: rows of groups that do not fit
// go to the global table.
constexpr int kLdsMaxProbes = 8;
constexpr int kHashCtlWords = 8;   // control words behind the LDS accumulator planes of the hash update kernel

// ---- a run of blocks in ONE launch (qsx_agg_update_blocks) ------------------------------------------------------------
// The reference issues one work order per 2-4 MB storage block (~120 K Q1 rows: half a microsecond of HBM time behind
// ~16 us of launch): a GPU work order takes a RUN of blocks, each with its own stripes.  The update kernels walk the
// tiles of the run; a tile never straddles two blocks (every block's last tile is short).  The table travels through the
// kernels' `pieces` argument — every kernel family already has it — and announces itself with a magic first word.
// Layout in device memory (8-byte words): [0] kBlockRunMagic, [1] number of blocks, then the word offsets (from word 0) of
// [2] first_tile for 1024-row tiles (num_blocks + 1 words: tiles before block b), [3] the same for 512-row tiles,
// [4] rows per block, [5] column base pointers [block * QSX_MAX_COLUMNS + column], [6] filter bitmap per block (0: none),
// [7] tiles per block when all blocks but the last have the same number of 1024-row tiles (0: ragged run, binary search),
// [8] dictionaries of the compressed columns [block * QSX_MAX_COLUMNS + column] (0: the state has none; an entry is 0 for a
// truncated or uncompressed column).
constexpr long long kBlockRunMagic = -0x424C4B52554E31ll;
constexpr int kBlockRunHeaderWords = 10;
struct BlockRunView {
  long long uniform_tiles;   // > 0: every block but the last has this many tiles — block of tile t = t / uniform_tiles, no search
  long long num_blocks;
  const long long *first_tile;
  const long long *rows;
  const void *const *cols;
  const unsigned long long *const *filters;
  const void *const *dicts;
};
__device__ __forceinline__ bool is_block_run(const long long *pieces) { return pieces != nullptr && pieces[0] == kBlockRunMagic; }
__device__ __forceinline__ BlockRunView block_run_view(const long long *table, int tile_rows) {
  BlockRunView v;
  v.num_blocks = table[1];
  v.uniform_tiles = tile_rows >= 1024 ? table[7] : 0;
  // (word offsets from the table's own start, not addresses: every read then goes through the kernel's const __restrict__
  // argument and can be a scalar load — through a pointer that was itself loaded from memory the compiler issues vector
  // loads + readfirstlane, a ~1.5 us dependent chain in front of every tile's stage: Q1 5.3 instead of 3.4 ms per 600 M rows)
  v.first_tile = table + (tile_rows >= 1024 ? table[2] : table[3]);
  v.rows = table + table[4];
  v.cols = reinterpret_cast<const void *const *>(table + table[5]);
  v.filters = table[6] != 0 ? reinterpret_cast<const unsigned long long *const *>(table + table[6]) : nullptr;
  v.dicts = table[8] != 0 ? reinterpret_cast<const void *const *>(table + table[8]) : nullptr;
  return v;
}
// Block of tile t: the last b with first_tile[b] <= t (workgroup-uniform: ~10 scalar steps for a thousand blocks).
__device__ __forceinline__ long long block_of_tile(const BlockRunView &v, long long t) {
  if (v.uniform_tiles > 0) return t / v.uniform_tiles;
  long long lo = 0, hi = v.num_blocks;   // first_tile[lo] <= t < first_tile[hi]
  while (hi - lo > 1) {
    const long long mid = (lo + hi) >> 1;
    if (v.first_tile[mid] <= t) lo = mid; else hi = mid;
  }
  return lo;
}
__device__ __forceinline__ int lds_find_or_insert(unsigned long long *l_keys, int S, unsigned long long code) {
  int s = static_cast<int>(mix64(code) >> 40) & (S - 1);
  const int limit = S < kLdsMaxProbes ? S : kLdsMaxProbes;
  for (int probes = 0; probes < limit; ++probes) {
    unsigned long long k = __hip_atomic_load(&l_keys[s], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    if (k == code) return s;
    if (k == kEmptyCode) {
      k = atomicCAS(&l_keys[s], kEmptyCode, code);
      if (k == kEmptyCode || k == code) return s;
    }
    s = (s + 1) & (S - 1);
  }
  return -1;
}

// The same; *inserted = this call claimed the slot.
__device__ __forceinline__ int lds_find_or_insert_new(unsigned long long *l_keys, int S, unsigned long long code, bool *inserted) {
  int s = static_cast<int>(mix64(code) >> 40) & (S - 1);
  const int limit = S < kLdsMaxProbes ? S : kLdsMaxProbes;
  *inserted = false;
  for (int probes = 0; probes < limit; ++probes) {
    unsigned long long k = __hip_atomic_load(&l_keys[s], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    if (k == code) return s;
    if (k == kEmptyCode) {
      k = atomicCAS(&l_keys[s], kEmptyCode, code);
      if (k == kEmptyCode) {
        *inserted = true;
        return s;
      }
      if (k == code) return s;
    }
    s = (s + 1) & (S - 1);
  }
  return -1;
}

__device__ __forceinline__ void lds_add(unsigned long long *p, unsigned long long inc, int kind) {
  switch (kind) {
    case kAccSumI64: atomicAdd(p, inc); break;
    case kAccMinI64: atomicMin(reinterpret_cast<long long *>(p), static_cast<long long>(inc)); break;
    case kAccMaxI64: atomicMax(reinterpret_cast<long long *>(p), static_cast<long long>(inc)); break;
    default: unsafeAtomicAdd(reinterpret_cast<double *>(p), __longlong_as_double(static_cast<long long>(inc))); break;
  }
}

}  // namespace qsx

#endif  // QSX_CSRC_AGG_COMMON_HPP_
