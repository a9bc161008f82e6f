// agg_jit.hpp — run-time plan shapes: the aggregation update kernel specialised for one
// configuration with hipRTC when the state is created.
//
// The AOT plan shapes (agg_shapes.hpp) show what specialisation buys: with the translated
// configuration a compile-time constant the interpreter folds away and the same kernel body runs
// ~3x fewer instructions (Q1: 1.25 vs 3.3-3.7 ms / 200 M rows; the interpreter is bound by
// instruction issue — 711 scalar instructions per 128-row wave tile — not by HBM).  A query
// processor does not know its plans at build time, so the shape is compiled when the state is
// created: the translated DevConfig is printed as a constexpr object into a ~30-line wrapper around
// the bundled kernel sources (jit_bundle.inc) and built for gfx950 by hipRTC (2-3 s, cached per
// process by source text).  The role of the reference's per-query code path selection
// (AggregationOperationState's strategy dispatch) taken one step further; QSX_AGG_JIT=0 disables it,
// any hipRTC failure falls back to the interpreter kernel (same body, same results).
#ifndef QSX_CSRC_AGG_JIT_HPP_
#define QSX_CSRC_AGG_JIT_HPP_

#include "agg_common.hpp"
#include "agg_hash_update.hpp"

namespace qsx {

struct JitKernel;

// Request for the specialised kernel of `dev` (translated configuration with its tile plan for 1024-row tiles already
// filled in); nullptr when run-time compilation is off.  Requests are cached by source text.  synchronous = compile
// (or wait for whoever compiles it) before returning; otherwise a background thread compiles and the caller polls.
struct JitRequest;
// Launch geometry folded into the shape as constants (table slots, log2 of the accumulator replication, tile buffers,
// workgroup families): they are fixed per state and path, and as constants they fold the plane / slot arithmetic of the
// tile loop instead of occupying scalar registers.
struct JitGeometry {
  int S, rep_shift, nbuf, ranges;
  int dir_gids;   // != 0: the group-directory variant (agg_common.hpp DirView) with that many LDS accumulators per aggregate
  int runs;       // != 0: the rows are a run of blocks (agg_common.hpp BlockRunView behind the `pieces` argument)
  int reg_groups; // != 0: that many groups per wave accumulate in registers (agg_hash_update.hpp, REG; small hash tables only)
  int dir_rows;       // group directory / dense state in LDS: rows per thread (0 = 1); 2 = one 2048-row tile in the LDS two 1024-row
                      // buffers would take (dense states of few entries: 4)
  int waves_per_eu;   // != 0: the waves per SIMD the shape is to be built for (LDS admits that many workgroups per CU; a shape
                      // that spills for it is rebuilt without)
};
JitRequest *jit_agg_request(const DevConfig &dev, int num_sums, bool dense, const JitGeometry &geometry, bool synchronous);
// 0: still compiling, 1: ready (*kernel set), -1: failed (hipRTC error: the interpreter stays in use)
int jit_request_state(JitRequest *request, const JitKernel **kernel);

// Launches it: the argument list of agg_hash_update_body.
int jit_agg_launch(const JitKernel *k, int grid, size_t lds_bytes, hipStream_t stream, const ColumnPointers &cols,
                   const void *const *dict_table_dev, int64_t n, const uint64_t *filter, const HashTableView &g,
                   const DenseView &dense, bool is_dense, int S, int rep_shift, int nbuf, int ranges, const long long *pieces,
                   int block = kABlock,    // (1024 for a dense state in LDS: geometry.dir_gids != 0 of a dense state)
                   const unsigned long long *const *null_table_dev = nullptr);   // states over nullable columns: the call's null
                                                                                 // bitmaps by null slot, a table in device memory

// The group-directory variant (geometry.dir_gids != 0): the argument list of agg_dir_update_kernel.
int jit_agg_launch_dir(const JitKernel *k, int grid, size_t lds_bytes, hipStream_t stream, const ColumnPointers &cols,
                       const void *const *dict_table_dev, int64_t n, const uint64_t *filter, const HashTableView &g, const DirView &d,
                       const long long *pieces = nullptr,    // pieces: the run table of a geometry.runs shape
                       const unsigned long long *const *null_table_dev = nullptr);

// Workgroups of `block` threads with lds_bytes of dynamic LDS that one CU keeps resident (registers and LDS); 0: unknown.
int jit_resident_blocks(const JitKernel *k, int block, size_t lds_bytes);

// Rows of a tile per thread in the run-time shapes of the hash path (tile = 256 x that): 4, or 2 with QSX_AGG_JIT_ROWS=2.
int jit_rows_per_thread();
// true: shapes are built by the compiler driver in a child process (the calling thread never waits for a compile);
// false: by hipRTC inside this process, in the calling thread
bool jit_compiles_out_of_process();

}  // namespace qsx

#endif  // QSX_CSRC_AGG_JIT_HPP_
