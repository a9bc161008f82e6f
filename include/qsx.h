/*
 * qsx.h — C ABI of the MI355X (gfx950) execution kernel for Quickstep's
 * relational_operators hot path.
 *
 * This is the drop-in boundary.  The reference has no FFI on
 * this path: the five hot WorkOrder::execute() bodies call C++ templates in
 * storage/ and expressions/ directly.  Each entry point below names the
 * reference function (file:line under the Quickstep tree) whose inner loop it
 * replaces; a GPU work order's execute() calls exactly these and nothing else
 * (see INTEGRATION.md for the reference-side binding).
 *
 * Conventions
 *   - every pointer named *_dev / documented "device" is HBM (or pinned,
 *     device-visible host memory); "host" pointers are ordinary host memory;
 *   - sizes are in rows (tuples) unless a name says _bytes;
 *   - every call is ordered on `stream` (a hipStream_t passed as void*; NULL
 *     is the default stream) and returns without synchronising unless its
 *     comment says "synchronises";
 *   - return value: QSX_OK (0) or a negative qsx_status_t; qsx_status_string()
 *     describes it.  Nothing falls back to a CPU implementation: with no
 *     usable GPU every compute entry point returns QSX_ERR_NO_DEVICE.
 *   - bitmaps are TupleIdSequence-compatible: 64-bit words, bit i of the
 *     sequence is  word[i >> 6] & (1ull << (63 - (i & 63)))  (MSB first,
 *     utility/BitVector.hpp:893-935), trailing bits zero.
 */
#ifndef QSX_H_
#define QSX_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* 6: the entry points over runs of blocks (qsx_*_blocks); nothing older changed its signature */
/* 17: the block forms over compressed key stripes (qsx_key_coding_t, qsx_join_*_blocks_coded, qsx_lip_*_blocks_coded,
 *     qsx_join_key_pack_blocks_coded), qsx_join_probe_exists_lip, qsx_copy_segments; nothing older changed its signature */
/* 18: qsx_partition_scatter_blocks (K9 reading a run of blocks where they lie); nothing older changed its signature */
/* 19: qsx_lip_build_from_join_table; nothing older changed its signature */
#define QSX_ABI_VERSION 19

typedef void *qsx_stream_t;

typedef enum qsx_status {
  QSX_OK = 0,
  QSX_ERR_INVALID_ARGUMENT = -1,
  QSX_ERR_NO_DEVICE = -2,
  QSX_ERR_OUT_OF_MEMORY = -3,
  QSX_ERR_HIP = -4,
  QSX_ERR_CAPACITY = -5,     /* caller-provided output / table too small */
  QSX_ERR_UNSUPPORTED = -6,
  QSX_ERR_TOO_MANY_GROUPS = -7,
  QSX_ERR_HASH_COLLISION = -8, /* wide group-by key: two keys shared a 64-bit hash (see QSX_GROUPS_HASH_COLLISION) */
  QSX_ERR_COMM = -9            /* RCCL missing or a collective failed (see qsx_last_error) */
} qsx_status_t;

/* Value types; numbering follows types/TypeID.hpp:32-43 (kInt, kLong, kFloat,
 * kDouble, kChar, kVarChar, kDate) so a reference TypeID can be passed through unchanged.
 * QSX_CHAR columns are fixed-width byte strings; 1, 2, 4 or 8 bytes wide they are legal as
 * group-by key components (compact-key packing), any width as the left side of
 * qsx_select_cmp_char.
 * QSX_DATE values are the reference's 8-byte DateLit {int32 year; uint8 month; uint8 day;
 * 2 bytes of padding} (types/DatetimeLit.hpp:38-43), compared year, month, day
 * (:65-90); the padding bytes are never looked at.  Legal in qsx_select_cmp[_sorted|_columns],
 * as an aggregation state's predicate column and group-by key, and as a sort / distinct key. */
typedef enum qsx_type {
  QSX_INT = 0,    /* int32 */
  QSX_LONG = 1,   /* int64 */
  QSX_FLOAT = 2,  /* float */
  QSX_DOUBLE = 3, /* double */
  QSX_CHAR = 4,
  QSX_DATE = 6
} qsx_type_t;

/* Comparison ids; numbering follows types/operations/comparisons/ComparisonID.hpp:36-42. */
typedef enum qsx_cmp {
  QSX_EQ = 0, QSX_NE = 1, QSX_LT = 2, QSX_LE = 3, QSX_GT = 4, QSX_GE = 5
} qsx_cmp_t;

const char *qsx_status_string(int status);
int qsx_abi_version(void);
/* sizeof(qsx_agg_config_t) as compiled into the library: lets a foreign-language
 * binding (ctypes, cgo, JNI) verify its struct mirror before the first call. */
size_t qsx_abi_sizeof_agg_config(void);
/* Number of usable gfx950 devices (0 when there is none). Never fails. */
int qsx_device_count(void);
/* One process per GPU is the deployment this library is written for, but nothing in it is bound to device 0: every call
 * works on the device that is current in the calling thread (HIP's per-thread current device), objects live on the device
 * they were created on.  An engine that selects its GPU with hipSetDevice rather than HIP_VISIBLE_DEVICES tells threads it
 * creates itself which device that is through these two (the host layer's Worker threads do: they take the device that was
 * current in the thread calling ForemanSingleNode::run).  device = an index below qsx_device_count(). */
int qsx_current_device(int *out_device);
int qsx_set_current_device(int device);
/* Text of the last HIP error seen by this thread ("" if none). */
const char *qsx_last_error(void);

/* ---- device memory plumbing (used by the host StorageManager; tests and
 * bench.py use torch allocations instead) ------------------------------- */
int qsx_device_alloc(size_t bytes, void **out_dev);
int qsx_device_free(void *dev);
int qsx_copy_to_device(void *dst_dev, const void *src_host, size_t bytes, qsx_stream_t stream);
int qsx_copy_to_host(void *dst_host, const void *src_dev, size_t bytes, qsx_stream_t stream);
int qsx_copy_on_device(void *dst_dev, const void *src_dev, size_t bytes, qsx_stream_t stream);
/* num_segments device-to-device copies in ONE launch: segment i copies bytes[i] bytes from src_dev[i] to dst_dev[i] (host
 * arrays, consumed before return; segments must not overlap each other's destinations).  What a loop of qsx_copy_on_device
 * does — the stripes of a run of blocks laid end to end (PartitionExchangeOperator's pieces, storage/InsertDestination.cpp
 * bulkInsertTuples block after block) — without a launch and ~10 us per block: 1.1 GB in 55 pieces 0.45 instead of 0.88 ms. */
int qsx_copy_segments(int64_t num_segments, const void *const *src_dev, void *const *dst_dev, const int64_t *bytes, qsx_stream_t stream);
int qsx_memset_device(void *dst_dev, int byte, size_t bytes, qsx_stream_t stream);
int qsx_stream_synchronize(qsx_stream_t stream);
/* One stream per Worker thread (query_execution/Worker.cpp:54-99 runs work orders one at a
 * time per worker; concurrent workers = concurrent streams). */
int qsx_stream_create(qsx_stream_t *out_stream);
int qsx_stream_destroy(qsx_stream_t stream);
/* The library keeps, per (calling host thread, stream), a scratch arena, a pinned + device staging pair for host tables and
 * a few device slots, so that no allocator runs on the hot path.  They are given back when the thread exits, when the
 * stream goes through qsx_stream_destroy (the calling thread's entries for it) and here: everything the CALLING thread
 * holds, after waiting for its streams — the call an engine makes on its out-of-memory path before retrying an allocation
 * (the library does the same before it gives up on an allocation of its own).  out_bytes_released may be NULL. */
int qsx_trim_scratch(size_t *out_bytes_released);
/* The other direction: when a device allocation INSIDE the library fails for lack of memory, `hook(user)` is called — from the
 * failing thread, with no library lock held — and the allocation is attempted once more.  The engine's buffer manager gives
 * back what it pools there (the host layer of this repo: its output-block slabs and scratch caches).  NULL removes the hook.
 * The hook may call qsx_device_free and qsx_trim_scratch; it must not call back into the operation that is failing. */
int qsx_set_out_of_memory_hook(void (*hook)(void *user), void *user);

/* ======================================================================
 * Select: predicate + projection
 * ====================================================================== */

/* K1.  out_bitmap[i] = (col[i] OP *literal) [AND filter[i]].
 * Replaces LiteralUncheckedComparator::compareValueAccessorAndStaticValueHelper
 * (types/operations/comparisons/LiteralComparators-inl.hpp:317-388) as reached
 * from ComparisonPredicate::getAllMatches (expressions/predicate/
 * ComparisonPredicate.cpp:115-334) and StorageBlock::getMatchesForPredicate
 * (storage/StorageBlock.cpp:1053-1083).
 *   type        QSX_INT / QSX_LONG / QSX_FLOAT / QSX_DOUBLE / QSX_DATE (column and literal share it)
 *   col_dev     n values, densely packed (a BasicColumnStore stripe)
 *   literal     host pointer to one value of `type`
 *   filter_dev  optional existing TupleIdSequence (n bits) or NULL; only rows set
 *               in it can be set in the output (short-circuit semantics,
 *               LiteralComparators-inl.hpp:344-356)
 *   out_bitmap_dev  (n+63)/64 words, fully overwritten
 *   out_count_dev   optional int64 on device receiving popcount(out) (overwritten) */
int qsx_select_cmp(int type, const void *col_dev, int64_t n, int op,
                   const void *literal, const uint64_t *filter_dev,
                   uint64_t *out_bitmap_dev, int64_t *out_count_dev,
                   qsx_stream_t stream);

/* K1 over a run of blocks in one launch: for every block b < num_blocks,
 *   block_out_bitmaps[b][i] = (block_cols[b][i] OP *literal) [AND block_filters[b][i]],  i < block_rows[b],
 * exactly what num_blocks calls of qsx_select_cmp produce.  The operator decides how many blocks a work order covers
 * (RelationalOperator::getAllWorkOrders, relational_operators/RelationalOperator.hpp:117-119; SelectOperator.cpp:83-150
 * makes one SelectWorkOrder per block): at the reference's 2-4 MB blocks a launch per block is launch-bound on this
 * device, a run of blocks is not (DESIGN.md "Work-order granularity").
 *   block_rows         host array: rows of each block (>= 0)
 *   block_cols         host array of device pointers: each block's stripe of the attribute
 *   block_filters      NULL, or host array of device pointers (an entry may be NULL): existing TupleIdSequences
 *   block_out_bitmaps  host array of device pointers: (block_rows[b]+63)/64 words each, fully overwritten
 *   out_counts_dev     optional device array of num_blocks int64: popcount of every block's bitmap (overwritten)
 * The host arrays are consumed before the call returns. */
int qsx_select_cmp_blocks(int type, int64_t num_blocks, const int64_t *block_rows, const void *const *block_cols, int op,
                          const void *literal, const uint64_t *const *block_filters, uint64_t *const *block_out_bitmaps,
                          int64_t *out_counts_dev, qsx_stream_t stream);

/* K1 on a CHAR(width) attribute: out_bitmap[i] = (col[i] OP literal) [AND filter[i]] with the reference's string
 * comparison — both sides are C strings that end at their first NUL byte or at their maximum length, compared byte
 * by byte as unsigned chars, a proper prefix being smaller (AsciiStringUncheckedComparator::strcmpHelper,
 * types/operations/comparisons/AsciiStringComparators.hpp:218-251).
 *   col_dev      n values of `width` bytes each (1..255), densely packed (a column-store stripe of a CHAR(width) attribute)
 *   literal      host pointer to literal_length (0..64) bytes; a NUL inside ends the literal
 * Other arguments as qsx_select_cmp.  (TPC-H: c_mktsegment = 'BUILDING', l_shipmode IN (...), ...) */
#define QSX_MAX_CHAR_LITERAL 64
int qsx_select_cmp_char(const void *col_dev, int width, int64_t n, int op, const void *literal, int literal_length,
                        const uint64_t *filter_dev, uint64_t *out_bitmap_dev, int64_t *out_count_dev, qsx_stream_t stream);

/* qsx_select_cmp_char over a run of blocks in one launch (arguments as qsx_select_cmp_blocks / qsx_select_cmp_char). */
int qsx_select_cmp_char_blocks(int width, int64_t num_blocks, const int64_t *block_rows, const void *const *block_cols, int op,
                               const void *literal, int literal_length, const uint64_t *const *block_filters,
                               uint64_t *const *block_out_bitmaps, int64_t *out_counts_dev, qsx_stream_t stream);

/* qsx_select_cmp on the SORT COLUMN of a sorted column store (ascending, no NULLs in the first n rows): the matches are
 * one row range found by two searches, not a scan.  Replaces SortColumnPredicateEvaluator::
 * EvaluatePredicateForUncompressedSortColumn (storage/ColumnStoreUtil.cpp:40-280) as called from
 * BasicColumnStoreTupleStorageSubBlock::getMatchesForPredicate (storage/BasicColumnStoreTupleStorageSubBlock.cpp:
 * 587-610; predicate_cost::kBinarySearch) and CompressedColumnStoreTupleStorageSubBlock.cpp:381.  Same arguments and
 * result as qsx_select_cmp. */
int qsx_select_cmp_sorted(int type, const void *col_dev, int64_t n, int op, const void *literal,
                          const uint64_t *filter_dev, uint64_t *out_bitmap_dev, int64_t *out_count_dev,
                          qsx_stream_t stream);

/* K1 with a second column as right operand: out_bitmap[i] = (lhs[i] OP rhs[i])
 * [AND filter[i]].  Replaces LiteralUncheckedComparator::compareColumnVectors
 * (types/operations/comparisons/LiteralComparators-inl.hpp:52-125) as reached from
 * ComparisonPredicate::getAllMatches for attribute-vs-attribute predicates
 * (expressions/predicate/ComparisonPredicate.cpp:300-334).  With the operands
 * gathered for a list of joined pairs (qsx_gather) it evaluates a residual join
 * predicate on all pairs at once — what HashInnerJoinWorkOrder does pair by pair
 * through Predicate::matchesForJoinedTuples (relational_operators/
 * HashJoinOperator.cpp:510-524) — and the component check of a hashed composite
 * key (qsx_join_key_pack). */
int qsx_select_cmp_columns(int type, const void *lhs_dev, const void *rhs_dev, int64_t n, int op,
                           const uint64_t *filter_dev, uint64_t *out_bitmap_dev,
                           int64_t *out_count_dev, qsx_stream_t stream);

/* ---- compressed attributes (CompressedColumnStoreTupleStorageSubBlock) ----------------------
 * An attribute of a compressed column-store block is a stripe of 1/2/4-byte unsigned CODES: the value
 * itself for a truncated INT/LONG attribute, or an index into a sorted dictionary
 * (storage/CompressedBlockBuilder.cpp:508-566, 590-650; compression/CompressionDictionary.hpp).
 * A comparison with a literal is first rewritten into a comparison on codes by the caller
 * (CompressedAttributePredicateTransformer::TransformPredicateOnCompressedAttribute,
 * storage/CompressedStoreUtil.cpp:51-140, 425-616: ALL / NONE / {=, !=, <, >=} code / code range) and
 * then evaluated on the code stripe — a quarter to an eighth of the bytes of the uncompressed column. */
typedef enum qsx_code_cmp {
  QSX_CODE_EQ = 0,    /* getEqualCodes           code == first            */
  QSX_CODE_NE = 1,    /* getNotEqualCodes        code != first            */
  QSX_CODE_LT = 2,    /* getLessCodes            code <  first            */
  QSX_CODE_GE = 3,    /* getGreaterOrEqualCodes  code >= first            */
  QSX_CODE_RANGE = 4  /* getCodesInRange         first <= code < second   */
} qsx_code_cmp_t;

/* K1 on a code stripe.  Replaces the scan loops of storage/
 * CompressedColumnStoreTupleStorageSubBlock.cpp:420-760 (attributes other than the sort column).
 * filter / out_bitmap / out_count as in qsx_select_cmp. */
int qsx_select_codes(int code_width, const void *codes_dev, int64_t n, int op, uint32_t first,
                     uint32_t second, const uint64_t *filter_dev, uint64_t *out_bitmap_dev,
                     int64_t *out_count_dev, qsx_stream_t stream);

/* qsx_select_codes on the code stripe of the block's SORT column (codes ascend with the values: truncation keeps the
 * order, dictionaries are sorted): the matching rows are one range found by two searches instead of a scan — the
 * sort-column branches of CompressedColumnStoreTupleStorageSubBlock::get{Equal,NotEqual,Less,GreaterOrEqual}Codes /
 * getCodesInRange (storage/CompressedColumnStoreTupleStorageSubBlock.cpp:420-760).  lineitem, orders and partsupp
 * are sorted on their key in the reference's TPC-H DDL (benchmarks/tpch/create.sql:57-121).  Same arguments and result. */
int qsx_select_codes_sorted(int code_width, const void *codes_dev, int64_t n, int op, uint32_t first, uint32_t second,
                            const uint64_t *filter_dev, uint64_t *out_bitmap_dev, int64_t *out_count_dev,
                            qsx_stream_t stream);

/* qsx_select_codes over a run of blocks in one launch: the comparison rewritten on every block's own codes (the dictionaries
 * differ from block to block: CompressedTupleStorageSubBlock::getMatchesForPredicate, storage/CompressedTupleStorageSubBlock.cpp:
 * 160-250, runs per block), one code width for the run.  block_ops / block_first / block_second: QSX_CODE_* comparison and its
 * literals per block, as for qsx_select_codes ("every code" = QSX_CODE_GE 0, "no code" = QSX_CODE_LT 0). */
int qsx_select_codes_blocks(int code_width, int64_t num_blocks, const int64_t *block_rows, const void *const *block_codes,
                            const int32_t *block_ops, const uint32_t *block_first, const uint32_t *block_second,
                            const uint64_t *const *block_filters, uint64_t *const *block_out_bitmaps, int64_t *out_counts_dev,
                            qsx_stream_t stream);

/* K1 on the sort column of sorted column-store blocks, over a run of blocks in one launch: every block is sorted on its
 * own, so every block's matches are its own row range (one wave per block searches the bounds, then the bitmaps of the run
 * are written).  Same results as qsx_select_cmp_sorted / qsx_select_codes_sorted block by block.  The reference's TPC-H layout
 * sorts lineitem on l_shipdate and orders on o_orderdate (benchmarks/tpch/create.sql:69-121): the predicates of Q1 and Q3 are
 * of this kind.
 *   qsx_select_cmp_sorted_blocks      uncompressed sort column: block_cols = value stripes, one literal for all blocks
 *   qsx_select_codes_sorted_blocks    compressed sort column: block_codes = code stripes of `code_width` bytes, the comparison
 *                                     rewritten per block (QSX_CODE_* op, first, second as for qsx_select_codes: the
 *                                     dictionaries differ from block to block)
 *   out_counts_dev                    optional device array of num_blocks int64: matches per block */
int qsx_select_cmp_sorted_blocks(int type, int64_t num_blocks, const int64_t *block_rows, const void *const *block_cols, int op,
                                 const void *literal, const uint64_t *const *block_filters, uint64_t *const *block_out_bitmaps,
                                 int64_t *out_counts_dev, qsx_stream_t stream);
int qsx_select_codes_sorted_blocks(int code_width, int64_t num_blocks, const int64_t *block_rows, const void *const *block_codes,
                                   const int32_t *block_ops, const uint32_t *block_first, const uint32_t *block_second,
                                   const uint64_t *const *block_filters, uint64_t *const *block_out_bitmaps, int64_t *out_counts_dev,
                                   qsx_stream_t stream);

/* Decode a code stripe into values of value_width (4 or 8) bytes: out[i] = dictionary[codes[i]], or the
 * zero-extended code when dictionary_dev is NULL (truncated attribute).  What
 * CompressedTupleStorageSubBlock::getAttributeValue does per tuple (storage/
 * CompressedTupleStorageSubBlock.hpp:225-300) for operators that consume values (joins, aggregates,
 * projections). */
int qsx_decode_codes(int code_width, const void *codes_dev, int64_t n, const void *dictionary_dev,
                     int value_width, void *out_dev, qsx_stream_t stream);

/* Bitmap algebra on TupleIdSequences of n bits (storage/TupleIdSequence.hpp:
 * intersectWith / unionWith / invert). op: 0 = AND, 1 = OR, 2 = AND NOT,
 * 3 = NOT a (b ignored). */
int qsx_bitmap_combine(int op, const uint64_t *a_dev, const uint64_t *b_dev, int64_t n,
                       uint64_t *out_dev, qsx_stream_t stream);
int qsx_bitmap_count(const uint64_t *bitmap_dev, int64_t n, int64_t *out_count_dev,
                     qsx_stream_t stream);

/* Scratch bytes qsx_compact_gather / qsx_bitmap_to_tids need for n input rows. */
size_t qsx_compact_workspace_bytes(int64_t n);

/* K2.  Order-preserving compaction of the rows selected by `bitmap` for
 * `ncols` columns at once.  Replaces StorageBlock::selectSimple ->
 * BasicColumnStoreTupleStorageSubBlock::bulkInsertTuplesWithRemappedAttributes
 * (storage/StorageBlock.cpp:390-399, storage/BasicColumnStoreTupleStorageSubBlock.cpp:339-425).
 *   cols / out_cols  host arrays of ncols device pointers
 *   widths           host array of value widths in bytes (1, 2, 4 or 8)
 *   out_count_dev    int64 on device: number of rows written
 * Output columns must have room for popcount(bitmap) rows. */
int qsx_compact_gather(int ncols, const void *const *cols, const int32_t *widths,
                       const uint64_t *bitmap_dev, int64_t n, void *const *out_cols,
                       int64_t *out_count_dev, void *workspace_dev, size_t workspace_bytes,
                       qsx_stream_t stream);

/* K2 over a run of blocks in one launch: the rows selected by block_bitmaps[b] in every block b, block after block
 * and in row order, land in ONE output stripe per column — the way consecutive SelectWorkOrders fill the blocks of an
 * InsertDestination (relational_operators/SelectOperator.cpp:161-195, storage/InsertDestination.cpp:222-260).
 *   block_cols       host array [b * ncols + c] of device pointers: block b's stripe of column c
 *   block_bitmaps    host array of device pointers: (block_rows[b]+63)/64 words each
 *   block_base_tids  host array or NULL (run-global row numbers); only used for out_tids_dev
 *   out_tids_dev     optional: the tuple id (base + row) of every output row
 *   out_count_dev    int64 on device: rows written
 * workspace: qsx_compact_blocks_workspace_bytes(num_blocks, block_rows).  Host arrays are consumed before return. */
size_t qsx_compact_blocks_workspace_bytes(int64_t num_blocks, const int64_t *block_rows);
int qsx_compact_gather_blocks(int ncols, const int32_t *widths, int64_t num_blocks, const int64_t *block_rows,
                              const void *const *block_cols, const uint64_t *const *block_bitmaps,
                              const int32_t *block_base_tids, void *const *out_cols, int32_t *out_tids_dev,
                              int64_t *out_count_dev, void *workspace_dev, size_t workspace_bytes, qsx_stream_t stream);

/* TupleIdSequence -> ascending tuple-id list (base_tid + position). */
int qsx_bitmap_to_tids(const uint64_t *bitmap_dev, int64_t n, int32_t base_tid,
                       int32_t *out_tids_dev, int64_t *out_count_dev,
                       void *workspace_dev, size_t workspace_bytes, qsx_stream_t stream);

/* Tuple-id list -> TupleIdSequence of num_bits bits: bit (tids[i] - base_tid) is
 * set for every i < n (duplicates allowed); the bitmap is fully overwritten.
 * The semi/anti join with a residual predicate collects the probe tuples that
 * kept at least one pair this way (TupleIdSequence filter of
 * HashSemiJoinWorkOrder::executeWithResidualPredicate, relational_operators/
 * HashJoinOperator.cpp:735-760). */
int qsx_tids_to_bitmap(const int32_t *tids_dev, int64_t n, int32_t base_tid, int64_t num_bits,
                       uint64_t *out_bitmap_dev, qsx_stream_t stream);

/* K5.  dst[i] = src[tids[i]] for i < n (value width 1/2/4/8 bytes: one element per thread; any other width up to
 * 4096 — CHAR(n) — byte by byte); tids < 0
 * write zero bytes (outer-join NULL padding; the null bit is the caller's).
 * Replaces ScalarAttribute::getAllValuesForJoin (expressions/scalar/
 * ScalarAttribute.cpp:185-225) as used by HashInnerJoinWorkOrder
 * (relational_operators/HashJoinOperator.cpp:527-539). */
int qsx_gather(int width, const void *src_dev, const int32_t *tids_dev, int64_t n,
               void *dst_dev, qsx_stream_t stream);

/* K5 over a relation stored as several blocks: tids are relation-global row
 * numbers, segment s holds rows [segment_first_row[s], segment_first_row[s+1])
 * (the last one is open-ended) at segment_ptrs[s].  At most 16384 segments (beyond 64 the table
 * travels through device memory) — the blocks of a relation, or of a run handed to qsx_join_probe_blocks.
 * Counterpart of the per-build-block loop of HashInnerJoinWorkOrder
 * (relational_operators/HashJoinOperator.cpp:494-540). */
int qsx_gather_segmented(int width, int num_segments, const void *const *segment_ptrs,
                         const int64_t *segment_first_row, const int32_t *tids_dev, int64_t n,
                         void *dst_dev, qsx_stream_t stream);

/* The null bits that travel with gathered values (NativeColumnVector's null BitVector,
 * types/containers/ColumnVector.hpp:130-401, filled by ScalarAttribute::getAllValuesForJoin / by
 * bulkInsertTuplesWithRemappedAttributes for nullable attributes): bit i of out_bitmap (TupleIdSequence bit order,
 * 1 = NULL, trailing bits zero) = the null bit of row tids[i]; segment_bitmaps[s] is the null bitmap of segment s
 * (NULL pointer: the segment has no NULLs); a negative tid yields 1 (outer-join padding,
 * HashOuterJoinWorkOrder's fillWithNulls, relational_operators/HashJoinOperator.cpp:1077-1080).  One segment with
 * first row 0 = a plain block. */
int qsx_bitmap_gather_segmented(int num_segments, const uint64_t *const *segment_bitmaps,
                                const int64_t *segment_first_row, const int32_t *tids_dev, int64_t n,
                                uint64_t *out_bitmap_dev, qsx_stream_t stream);

/* ======================================================================
 * Hash join
 * ====================================================================== */

typedef struct qsx_join_table qsx_join_table_t;

/* JoinHashTable for one single-attribute INT or LONG key (the case
 * SimplifyHashTableImplTypeProto reduces to SimpleScalarSeparateChaining,
 * storage/HashTableFactory.cpp:58-68).  Duplicate keys are kept; the value is
 * a 32-bit build-side tuple reference chosen by the caller (see base_tid).
 * est_entries is the optimizer estimate (ExecutionGenerator.cpp:903-904); the
 * table grows on its own when the estimate is exceeded (counterpart of
 * HashTable::resize, storage/HashTable.hpp:1437-1440).  Synchronises. */
int qsx_join_table_create(int key_type, int64_t est_entries, qsx_join_table_t **out);
/* Same table, addressed directly by (key - min_key): one 4-byte head word per
 * key value instead of a hashed probe sequence, duplicates chained through an
 * overflow list.  For the caller that knows the build-side join attribute lies
 * exactly in [min_key, max_key] — the optimizer condition under which the
 * reference swaps the hash table for a BitVectorExactFilter
 * (query_optimizer/rules/InjectJoinFilters.cpp:130-150, exact min/max
 * statistics and a bounded value range; kMaxFilterSize there = 1e9) — but the
 * tuple references are kept, so every join flavour of this header works on it
 * (inner joins with build-side outputs and duplicate keys included).
 *   key_stride  1, or a power of two 2^s: the table holds the progression
 *               min_key, min_key + 2^s, ... only — what one hash partition
 *               (key & (P-1), catalog/PartitionSchemeHeader.hpp:200-214) of a
 *               dense key domain looks like after the multi-GPU shuffle, so the
 *               head array stays (max_key - min_key) / P words on every GPU.
 * A build key that is not a member (outside the range / off the stride) is a
 * broken precondition: the row is skipped and qsx_join_table_size reports
 * QSX_ERR_INVALID_ARGUMENT.  Probe keys that are not members simply do not
 * match.  Synchronises. */
int qsx_join_table_create_dense(int key_type, int64_t min_key, int64_t max_key, int64_t key_stride,
                                int64_t est_entries, qsx_join_table_t **out);
int qsx_join_table_destroy(qsx_join_table_t *table);
/* qsx_join_table_destroy WITHOUT its wait for the device: for a caller that knows every call that used the table has
 * completed — each followed by a qsx_stream_synchronize, as every work order of the host layer is — and that nothing else is
 * queued against it: DestroyHashOperator's place in the DAG behind the last HashJoin work order (relational_operators/
 * DestroyHashOperator.cpp:44-60, query_execution/QueryContext.hpp:352-360 destroyJoinHashTable).  qsx_join_table_destroy waits
 * for everything queued on the table's device first, the other operators' kernels included: with an aggregation still in
 * flight on other streams that wait was measured at 7 ms now and then, against 4 ms for the whole step (ABI 16). */
int qsx_join_table_release(qsx_join_table_t *table);

/* Composite join keys.  The reference keeps the key components in the bucket,
 * hashes them with a CombineHashes fold and compares hash + components on lookup
 * (HashTable::putValueAccessorCompositeKey / getAllFromValueAccessorCompositeKey,
 * storage/HashTable.hpp:1463-1575, 1835-1880; hashCompositeKey :2109-2119;
 * SeparateChainingHashTable::getNextEntryForCompositeKey .hpp:1033-1060).
 * Here the components are folded into ONE LONG key per row, which then goes
 * through the single-key table above:
 *   sum of widths <= 8 bytes: the components' bytes at running offsets of a zeroed
 *     64-bit word (little endian) — equal words <=> equal composite keys,
 *     *out_exact = 1;
 *   wider: the reference's composite hash (identity hash of every component,
 *     CombineHashes fold, utility/HashPair.hpp:47-58), *out_exact = 0: the pairs
 *     the table returns must be verified component by component
 *     (qsx_gather + qsx_select_cmp_columns(QSX_EQ) + qsx_compact_gather).
 *   types   host array: QSX_INT or QSX_LONG per component (at most QSX_MAX_KEYS)
 *   out_dev n int64 on device;  out_exact  host int */
int qsx_join_key_pack(int ncols, const void *const *cols, const int32_t *types, int64_t n,
                      int64_t *out_dev, int *out_exact, qsx_stream_t stream);
/* A CHAR(n <= 8) join key attribute as a LONG key: the bytes of the value (up to its first NUL) little-endian, zero behind
 * them — two CHAR values are equal strings exactly when their keys are equal, so the join tables, qsx_join_key_pack (as a
 * LONG component of a composite key) and the partition function take it from there.  The reference hashes the string
 * (types/TypedValue.hpp:626-633) and compares it on lookup; CHAR(n > 8) and VARCHAR keys are not supported. */
int qsx_join_key_pack_char(const void *col_dev, int width, int64_t n, int64_t *out_dev, qsx_stream_t stream);

/* qsx_join_key_pack over a run of blocks in one launch: the packed keys of all blocks in ONE stripe, block after block
 * (out_dev: sum of block_rows values) — what the run forms of build and probe then take as a single key stripe.
 *   block_cols   host array [b * ncols + k] of device pointers: block b's stripe of key component k */
int qsx_join_key_pack_blocks(int ncols, const int32_t *types, int64_t num_blocks, const int64_t *block_rows,
                             const void *const *block_cols, int64_t *out_dev, int *out_exact, qsx_stream_t stream);
/* The same over blocks that hold some components COMPRESSED (see qsx_key_coding_t below): block_code_widths[b * ncols + k] =
 * 0 (block_cols holds values), 1, 2 or 4 (it holds codes that wide), block_dictionaries[b * ncols + k] = the block's dictionary
 * of that component on the device, or NULL: truncated values.  Either array may be NULL (no component is compressed). */
int qsx_join_key_pack_blocks_coded(int ncols, const int32_t *types, int64_t num_blocks, const int64_t *block_rows,
                                   const void *const *block_cols, const int32_t *block_code_widths, const void *const *block_dictionaries,
                                   int64_t *out_dev, int *out_exact, qsx_stream_t stream);

/* Drop every entry, keep the allocation (a new query re-using the table;
 * counterpart of DestroyHashOperator + re-creation, relational_operators/
 * DestroyHashOperator.cpp:70-72).  Stream-ordered. */
int qsx_join_table_clear(qsx_join_table_t *table, qsx_stream_t stream);
/* Number of entries inserted so far.  Synchronises on `stream`. */
int qsx_join_table_size(qsx_join_table_t *table, int64_t *out_entries, qsx_stream_t stream);

/* K3.  Insert (keys[i] -> base_tid + i) for every row i < n that is set in
 * filter (all rows when filter_dev is NULL).  Safe to call concurrently from
 * several host threads / streams on the same table, like many
 * BuildHashWorkOrders sharing one JoinHashTable.  Replaces
 * HashTable::putValueAccessor (storage/HashTable.hpp:1358-1461) ->
 * SimpleScalarSeparateChainingHashTable::putInternal / locateBucketForInsertion
 * (storage/SimpleScalarSeparateChainingHashTable.hpp:1062-1113) as called from
 * BuildHashWorkOrder::execute (relational_operators/BuildHashOperator.cpp:162-207).
 * May synchronise when the table has to grow. */
int qsx_join_build(qsx_join_table_t *table, const void *keys_dev, int64_t n,
                   int32_t base_tid, const uint64_t *filter_dev, qsx_stream_t stream);

/* K3 over a run of build blocks in one launch: num_blocks calls of qsx_join_build (block b with base_tid =
 * block_base_tids[b], its own key stripe and filter) as one.  BuildHashOperator makes one BuildHashWorkOrder per block
 * (relational_operators/BuildHashOperator.cpp:70-130); how many blocks a work order covers is the operator's decision
 * (RelationalOperator.hpp:117-119).  Host arrays are consumed before return; may synchronise when the table grows. */
int qsx_join_build_blocks(qsx_join_table_t *table, int64_t num_blocks, const int64_t *block_rows, const void *const *block_keys,
                          const int32_t *block_base_tids, const uint64_t *const *block_filters, qsx_stream_t stream);

/* K4.  Inner-join probe: for every probe row i < n (set in filter) and every
 * table entry with an equal key, emit (probe_base_tid + i, build_tid).
 * Replaces HashTable::getAllFromValueAccessorImpl (storage/HashTable.hpp:
 * 2145-2181) + the pair collector (relational_operators/HashJoinOperator.cpp:
 * 76-130) inside HashInnerJoinWorkOrder::execute (:450-541).
 *   out_*_dev      arrays of `capacity` int32 each
 *   out_count_dev  int64 on device: total number of matches (also when it
 *                  exceeds capacity; pairs beyond capacity are not written —
 *                  the caller compares and re-runs with a larger buffer)
 * Pair order is unspecified (reference: unordered_map iteration order,
 * HashJoinOperator.cpp:480).  (A directly addressed table probed under a filter
 * runs count / scan / write passes and happens to emit the pairs in probe-row
 * order; QSX_JOIN_TWO_PASS=1 / 0 forces that form on / off.)  When the build key
 * is unique — what impliesUniqueAttributes tells the reference's optimizer — the
 * number of probe rows that pass the filter bounds the output, so no counting
 * call is needed to size the arrays. */
int qsx_join_probe(qsx_join_table_t *table, const void *keys_dev, int64_t n,
                   int32_t probe_base_tid, const uint64_t *filter_dev,
                   int32_t *out_probe_tid_dev, int32_t *out_build_tid_dev,
                   int64_t capacity, int64_t *out_count_dev, qsx_stream_t stream);

/* Number of matches only (no output written); same semantics as out_count_dev. */
int qsx_join_probe_count(qsx_join_table_t *table, const void *keys_dev, int64_t n,
                         const uint64_t *filter_dev, int64_t *out_count_dev,
                         qsx_stream_t stream);

/* K4 over a run of probe blocks in one launch: the pairs of num_blocks calls of qsx_join_probe (block b probed with
 * probe_base_tid = block_base_tids[b]) in one pair list with one counter.  HashJoinOperator makes one probe work order per
 * probe block (relational_operators/HashJoinOperator.cpp:203-260); how many blocks a work order covers is the operator's
 * decision (RelationalOperator.hpp:117-119) and a launch per 2-4 MB block is launch-bound on this device.
 *   block_rows / block_keys   host arrays: rows and key stripe (device pointer) of each block
 *   block_base_tids  host array, or NULL: block b's rows are then numbered from the rows of the blocks before it
 *                    (run-global row numbers — what qsx_gather_segmented takes); base + rows must fit int32
 *   block_filters    NULL, or host array of device pointers (entries may be NULL)
 * Other arguments and the capacity / count contract as qsx_join_probe.  The host arrays are consumed before return. */
int qsx_join_probe_blocks(qsx_join_table_t *table, int64_t num_blocks, const int64_t *block_rows, const void *const *block_keys,
                          const int32_t *block_base_tids, const uint64_t *const *block_filters, int32_t *out_probe_tid_dev,
                          int32_t *out_build_tid_dev, int64_t capacity, int64_t *out_count_dev, qsx_stream_t stream);

/* K4 + K5 in one pass: the OUTPUT RELATION of an inner join over a run of probe blocks, written by the probe itself.
 * HashInnerJoinWorkOrder collects (probe, build) tuple-id pairs and then materialises every output attribute from them
 * (relational_operators/HashJoinOperator.cpp:494-560: one ValueAccessor pass per build block, ScalarAttribute::
 * getAllValuesForJoin per attribute); here a matching lane reads the projected attributes of its probe row and of the build
 * tuple it found and stores them where it would have stored the pair — the pair list is never written, no launch per
 * attribute.  Output tuple i (i < min(*out_count, capacity)) holds column c at out_columns[c] + i * width[c]; tuples come
 * out in no particular order (like the pairs), all columns of a tuple at the same position.
 *   num_columns     1 .. QSX_MAX_PROJECTED output columns; width[c] = 1, 2, 4 or 8 bytes
 *   on_build[c]     0: column c is a probe-side attribute, its stripe in probe block b is probe_stripes[b * num_columns + c];
 *                   1: a build-side attribute, stored as num_build_segments blocks in build-tuple-id order: segment s holds
 *                   the tuples from build_first_tids[s] on (the ids the table was built with: base_tid + row) and its stripe is
 *                   build_stripes[s * num_columns + c].  Entries of the other side's columns are ignored.
 * block_rows / block_keys / block_filters, capacity and *out_count_dev as qsx_join_probe_blocks (probe tuple ids are not
 * reported, so there are no base tids).  Directly addressed tables — and hashed tables whose keys got a directly addressed
 * shadow — run the fused kernel; any other table is probed into a scratch pair list and gathered from it inside the call
 * (same result, the unfused cost).  All host arrays are consumed before return. */
#define QSX_MAX_PROJECTED 16
typedef struct qsx_join_projection {
  int32_t num_columns;
  int32_t width[QSX_MAX_PROJECTED];
  int32_t on_build[QSX_MAX_PROJECTED];
  const void *const *probe_stripes;     /* host array [num_blocks * num_columns] of device pointers */
  int32_t num_build_segments;
  const int64_t *build_first_tids;      /* host array [num_build_segments], ascending */
  const void *const *build_stripes;     /* host array [num_build_segments * num_columns] of device pointers */
  void *const *out_columns;             /* host array [num_columns] of device pointers, capacity values each */
} qsx_join_projection_t;
int qsx_join_probe_project_blocks(qsx_join_table_t *table, int64_t num_blocks, const int64_t *block_rows,
                                  const void *const *block_keys, const uint64_t *const *block_filters,
                                  const qsx_join_projection_t *projection, int64_t capacity, int64_t *out_count_dev,
                                  qsx_stream_t stream);

/* qsx_join_probe_count over a run of probe blocks: the number of pairs qsx_join_probe_blocks would emit. */
int qsx_join_probe_count_blocks(qsx_join_table_t *table, int64_t num_blocks, const int64_t *block_rows,
                                const void *const *block_keys, const uint64_t *const *block_filters,
                                int64_t *out_count_dev, qsx_stream_t stream);

/* Existence probe for semi / anti joins: out_bitmap bit i = filter[i] AND
 * (key i found) when anti == 0, filter[i] AND NOT found when anti != 0.
 * Replaces HashTable::runOverKeysFromValueAccessorIfMatch[Not]Found
 * (storage/HashTable.hpp:1979-2062) as used by HashSemiJoinWorkOrder /
 * HashAntiJoinWorkOrder (relational_operators/HashJoinOperator.cpp:795-816, 860-877). */
int qsx_join_probe_exists(qsx_join_table_t *table, const void *keys_dev, int64_t n,
                          const uint64_t *filter_dev, int anti,
                          uint64_t *out_bitmap_dev, int64_t *out_count_dev,
                          qsx_stream_t stream);

/* qsx_join_probe_exists over a run of probe blocks in one launch: block b's bitmap goes to block_out_bitmaps[b]
 * ((block_rows[b]+63)/64 words, fully overwritten); out_count_dev (optional) receives the number of set bits of the
 * whole run.  (One HashSemiJoinWorkOrder / HashAntiJoinWorkOrder per probe block in the reference,
 * relational_operators/HashJoinOperator.cpp:262-330.) */
int qsx_join_probe_exists_blocks(qsx_join_table_t *table, int64_t num_blocks, const int64_t *block_rows,
                                 const void *const *block_keys, const uint64_t *const *block_filters, int anti,
                                 uint64_t *const *block_out_bitmaps, int64_t *out_count_dev, qsx_stream_t stream);

/* The block forms over COMPRESSED key stripes.  A CompressedColumnStore block stores an INT / LONG attribute as values, as
 * truncated values of 1 / 2 / 4 bytes, or as 1 / 2 / 4-byte codes into the block's own sorted dictionary — each block decides
 * for itself (storage/CompressedBlockBuilder.cpp:508-566, 590-650) — and the reference's join reads the key through
 * CompressedTupleStorageSubBlock::getAttributeValue (storage/CompressedTupleStorageSubBlock.hpp:225-300: dictionary lookup
 * or widening, per tuple).  These forms take the stripes as they lie: block_keys[b] is block b's stripe of
 * coding->block_code_width[b]-byte codes (0: the stripe holds values of the table's key type, as in the plain forms) and
 * coding->block_dictionaries[b] the block's dictionary of key-type values on the device (NULL entry or NULL array: the codes
 * are truncated values, widened without sign).  coding == NULL or all widths 0: exactly the plain form.  No qsx_decode_codes
 * pass, no decoded stripe: the key costs its code width in HBM traffic.
 * qsx_join_probe_project_blocks_coded: a probe-side output column whose stripe pointers are the blocks' key stripes IS the
 * join key and comes out as its value (the column's width must be the key type's); on a table without a directly addressed
 * form such a column returns QSX_ERR_UNSUPPORTED (present its decoded stripe instead). */
typedef struct qsx_key_coding {
  const int32_t *block_code_width;        /* host array [num_blocks]: 0, 1, 2 or 4 */
  const void *const *block_dictionaries;  /* NULL, or host array [num_blocks] of device pointers (NULL entries: truncation) */
} qsx_key_coding_t;
int qsx_join_build_blocks_coded(qsx_join_table_t *table, int64_t num_blocks, const int64_t *block_rows, const void *const *block_keys,
                                const qsx_key_coding_t *coding, const int32_t *block_base_tids, const uint64_t *const *block_filters,
                                qsx_stream_t stream);
int qsx_join_probe_blocks_coded(qsx_join_table_t *table, int64_t num_blocks, const int64_t *block_rows, const void *const *block_keys,
                                const qsx_key_coding_t *coding, const int32_t *block_base_tids, const uint64_t *const *block_filters,
                                int32_t *out_probe_tid_dev, int32_t *out_build_tid_dev, int64_t capacity, int64_t *out_count_dev,
                                qsx_stream_t stream);
int qsx_join_probe_project_blocks_coded(qsx_join_table_t *table, int64_t num_blocks, const int64_t *block_rows,
                                        const void *const *block_keys, const qsx_key_coding_t *coding,
                                        const uint64_t *const *block_filters, const qsx_join_projection_t *projection,
                                        int64_t capacity, int64_t *out_count_dev, qsx_stream_t stream);
int qsx_join_probe_count_blocks_coded(qsx_join_table_t *table, int64_t num_blocks, const int64_t *block_rows,
                                      const void *const *block_keys, const qsx_key_coding_t *coding,
                                      const uint64_t *const *block_filters, int64_t *out_count_dev, qsx_stream_t stream);
int qsx_join_probe_exists_blocks_coded(qsx_join_table_t *table, int64_t num_blocks, const int64_t *block_rows,
                                       const void *const *block_keys, const qsx_key_coding_t *coding,
                                       const uint64_t *const *block_filters, int anti, uint64_t *const *block_out_bitmaps,
                                       int64_t *out_count_dev, qsx_stream_t stream);

/* ======================================================================
 * Aggregation
 * ====================================================================== */

typedef struct qsx_agg_state qsx_agg_state_t;

/* Strategy; chosen by the caller with the optimizer's rules
 * (query_optimizer/ExecutionGenerator.cpp:1922-1965,
 *  query_optimizer/cost_model/StarSchemaSimpleCostModel.cpp:614-776). */
typedef enum qsx_agg_strategy {
  QSX_AGG_SINGLE_STATE = 0,   /* no GROUP BY (AggregationOperationState.cpp:476-519) */
  QSX_AGG_COMPACT_KEY = 1,    /* ThreadPrivateCompactKeyHashTable (.cpp:203-304) */
  QSX_AGG_COLLISION_FREE = 2, /* CollisionFreeVectorTable (.hpp:530-645) */
  QSX_AGG_GENERIC = 3         /* PackedPayloadHashTable (.hpp:838-909); keys <= 8 bytes packed */
} qsx_agg_strategy_t;

typedef enum qsx_agg_fn {
  QSX_AGG_COUNT_STAR = 0, /* COUNT(*)            -> int64 */
  QSX_AGG_SUM = 1,        /* SUM(int|long) -> int64 ; SUM(float|double|expr) -> double */
  QSX_AGG_AVG = 2,        /* sum / (double)count -> double (AggregationHandleAvg.cpp:144-155) */
  QSX_AGG_MIN = 3,        /* AggregationHandleMin.cpp:45-120: result has the argument's type   */
  QSX_AGG_MAX = 4,        /* AggregationHandleMax.cpp:45-120  (DOUBLE for an expression)       */
  QSX_AGG_COUNT = 5       /* COUNT(x): rows whose argument is not NULL -> int64 (AggregationHandleCount.hpp:98-118,
                             the count_star = false, nullable_type = true instantiation) */
} qsx_agg_fn_t;

/* Operand of an expression instruction or an aggregate argument. */
typedef enum qsx_operand_kind {
  QSX_OPD_COLUMN = 0, /* index = input column */
  QSX_OPD_CONST = 1,  /* index = slot in consts[] */
  QSX_OPD_TEMP = 2    /* index = result of an earlier instruction (its dst) */
} qsx_operand_kind_t;

typedef struct qsx_operand {
  int32_t kind;
  int32_t index;
} qsx_operand_t;

typedef enum qsx_expr_op {
  QSX_EX_ADD = 0, QSX_EX_SUB = 1, QSX_EX_MUL = 2, QSX_EX_DIV = 3
} qsx_expr_op_t;

/* temp[dst] = a OP b, evaluated per row in IEEE double, operands converted
 * to double first (Quickstep promotes INT/LONG/FLOAT op DOUBLE to DOUBLE,
 * types/operations/binary_operations/ArithmeticBinaryOperators.hpp:203-340).
 * This replaces the temporary NativeColumnVectors of
 * ScalarBinaryExpression::getAllValues (expressions/scalar/
 * ScalarBinaryExpression.cpp:100-195) — fused, nothing is materialised. */
typedef struct qsx_expr_instr {
  int32_t op;
  int32_t dst; /* 0 .. QSX_MAX_TEMPS-1 */
  qsx_operand_t a;
  qsx_operand_t b;
} qsx_expr_instr_t;

typedef struct qsx_agg_desc {
  int32_t fn;        /* qsx_agg_fn_t */
  qsx_operand_t arg; /* ignored for COUNT(*); COLUMN or TEMP */
} qsx_agg_desc_t;

/* Comparison predicate held by the aggregation state (the reference keeps the
 * predicate in AggregationOperationState, .cpp:440-445); conjunction of
 * `column OP literal` terms evaluated inside the aggregation kernel. */
typedef struct qsx_pred_term {
  int32_t column;
  int32_t op;           /* qsx_cmp_t */
  union { int32_t i32; int64_t i64; float f32; double f64; } literal; /* typed like the column (QSX_DATE: the 8 DateLit bytes in i64) */
} qsx_pred_term_t;

#define QSX_MAX_COLUMNS 16
#define QSX_MAX_KEYS 4
#define QSX_MAX_AGGS 8
#define QSX_MAX_INSTRS 16
#define QSX_MAX_TEMPS 8
#define QSX_MAX_CONSTS 8
#define QSX_MAX_PRED_TERMS 4

/* K11 on its own: out[i] = value of `result` after running the program over row i — the projection of a scalar
 * expression by a SelectWorkOrder / HashJoinWorkOrder (Scalar::getAllValues:
 * expressions/scalar/ScalarBinaryExpression.cpp:100-195, ScalarAttribute.cpp:171-226, ScalarLiteral), one fused pass instead
 * of a NativeColumnVector per expression node.  Inside an aggregation the same program runs fused with the accumulation
 * (qsx_agg_config_t::instrs) and never comes here.
 *   cols / types  the num_columns input stripes (INT / LONG / FLOAT / DOUBLE), n rows each
 *   instrs        host array, at most QSX_MAX_INSTRS; consts: host array of QSX_MAX_CONSTS doubles (or NULL)
 *   result        the operand whose value is written (a column: its conversion to DOUBLE; a constant; a temp)
 *   out_dev       n doubles */
int qsx_eval_expression(int num_columns, const void *const *cols, const int32_t *types, int num_instrs,
                        const qsx_expr_instr_t *instrs, const double *consts, qsx_operand_t result, int64_t n,
                        double *out_dev, qsx_stream_t stream);

/* The same over INT / LONG operands in INTEGER arithmetic — what the reference's ArithmeticBinaryOperators compute for integer
 * argument types (types/operations/binary_operations/ArithmeticBinaryOperators.hpp:203-340): INT op INT is an INT (32-bit
 * wrap-around), an operation with a LONG operand a LONG; `/` truncates toward zero, x / 0 gives 0.  consts: host array of
 * QSX_MAX_CONSTS int64 (a constant that fits 32 bits counts as an INT operand).  out_width 4 or 8: the result stripe holds
 * INT or LONG values (the caller knows the expression's type by the same rule). */
int qsx_eval_expression_long(int num_columns, const void *const *cols, const int32_t *types, int num_instrs,
                             const qsx_expr_instr_t *instrs, const int64_t *consts, qsx_operand_t result, int64_t n, int out_width,
                             void *out_dev, qsx_stream_t stream);

typedef struct qsx_agg_config {
  int32_t strategy;                       /* qsx_agg_strategy_t */
  int32_t num_columns;                    /* columns handed to every qsx_agg_update */
  int32_t column_type[QSX_MAX_COLUMNS];   /* qsx_type_t */
  int32_t column_width[QSX_MAX_COLUMNS];  /* bytes; 4/8/4/8 for INT/LONG/FLOAT/DOUBLE, 1/2/4/8 for CHAR */
  int32_t num_keys;
  int32_t key_column[QSX_MAX_KEYS];       /* GROUP BY order; packed little-endian at running
                                             offsets into a 64-bit code (ThreadPrivateCompactKeyHashTable.cpp:216-232).
                                             A key wider than 8 bytes (hash strategies; up to 3 words of 8 bytes,
                                             PackedPayloadHashTable.hpp:499-521 takes any composite key) is grouped by a
                                             64-bit hash of its packed words and verified exactly: every word costs two
                                             of the QSX_MAX_AGGS accumulators of the state (QSX_ERR_UNSUPPORTED when
                                             they do not fit), see QSX_GROUPS_HASH_COLLISION */
  int32_t num_instrs;
  qsx_expr_instr_t instrs[QSX_MAX_INSTRS];
  double consts[QSX_MAX_CONSTS];
  int32_t num_aggs;
  qsx_agg_desc_t aggs[QSX_MAX_AGGS];
  int32_t num_pred_terms;
  qsx_pred_term_t pred[QSX_MAX_PRED_TERMS];
  int64_t est_groups;                     /* optimizer estimate (ExecutionGenerator.cpp:1922-1965).  The table is created with
                                             8x head-room and GROWS past it like the reference's tables
                                             (PackedPayloadHashTable::resize, ThreadPrivateCompactKeyHashTable::resize): groups
                                             that find no slot during an update go to a spill log inside the state (1 Mi
                                             records) instead of being dropped, and the table is enlarged 4x or more and the
                                             log folded back in before the next update launch that sees it and before every
                                             num_groups / finalize / export / merge.  Only when one update call spills more than
                                             the log holds (an estimate several orders of magnitude too low on a call of
                                             millions of rows) are rows lost: num_groups / finalize / export then return
                                             QSX_ERR_TOO_MANY_GROUPS and the caller re-runs with a larger estimate. */
  int64_t num_entries;                    /* COLLISION_FREE only: max_key + 1 (StarSchemaSimpleCostModel.cpp:707) */
  int32_t column_code_width[QSX_MAX_COLUMNS]; /* 0: the column arrives as values.  1 / 2 / 4: it arrives as a stripe of
                                             unsigned codes of that width (a compressed attribute of a
                                             CompressedColumnStoreTupleStorageSubBlock) and qsx_agg_update_coded decodes it
                                             while reading: INT / LONG / FLOAT / DOUBLE columns only */
  int32_t column_nullable[QSX_MAX_COLUMNS];  /* != 0: the attribute's type is nullable; qsx_agg_update_nullable hands its null
                                             bitmap in with every block.  Semantics are the reference's, applied inside the
                                             update kernel: a tuple with a NULL group-by key is skipped
                                             (PackedPayloadHashTable.hpp:861-867); a comparison with NULL is not true, so a NULL
                                             in a predicate column drops the tuple; an aggregate skips the tuples whose argument
                                             — or any operand of the expression it aggregates — is NULL while COUNT(*) still
                                             counts them (AggregationHandleSum.hpp:105-120 iterateUnaryInl,
                                             AggregationHandleCount.hpp:98-118); SUM / AVG / MIN / MAX of a group that saw no
                                             non-NULL argument finalize as NULL (AggregationHandleSum.cpp:100-120,
                                             AggregationHandleAvg.cpp:144-155) */
} qsx_agg_config_t;

/* Counterpart of the AggregationOperationState constructor
 * (storage/AggregationOperationState.cpp:74-252) + InitializeAggregation
 * (zeroing; CollisionFreeVectorTable.hpp:136-143).  Synchronises. */
int qsx_agg_state_create(const qsx_agg_config_t *config, qsx_agg_state_t **out);
int qsx_agg_state_destroy(qsx_agg_state_t *state);
/* Back to the freshly initialised state (InitializeAggregationOperator,
 * relational_operators/InitializeAggregationOperator.cpp:91), keeping the
 * allocation.  Stream-ordered. */
int qsx_agg_state_clear(qsx_agg_state_t *state, qsx_stream_t stream);

/* K6/K7/K8 (+K11 fused).  Accumulate n rows into the state.  Replaces
 * AggregationOperationState::aggregateBlock (storage/AggregationOperationState.cpp:
 * 428-474) and below: aggregateBlockSingleState (:476-519),
 * ThreadPrivateCompactKeyHashTable::upsertValueAccessorCompositeKey (.cpp:203-304),
 * CollisionFreeVectorTable::upsertValueAccessor* (.hpp:530-645),
 * PackedPayloadHashTable::upsertValueAccessorCompositeKey (.hpp:838-909).
 *   cols        host array of config.num_columns device pointers (n values each)
 *   filter_dev  optional TupleIdSequence restricting the rows (e.g. a LIP result)
 * Safe to call concurrently on one state from several host threads/streams,
 * like many AggregationWorkOrders sharing one state. */
int qsx_agg_update(qsx_agg_state_t *state, const void *const *cols, int64_t n,
                   const uint64_t *filter_dev, qsx_stream_t stream);

/* qsx_agg_update over a RUN of blocks in one launch.  The reference issues one AggregationWorkOrder per 2-4 MB storage
 * block (relational_operators/AggregationOperator.cpp:38-79) — about 120 K Q1 rows, half a microsecond of HBM time
 * behind ~16 us of launch; the GPU operator hands a run of blocks to ONE work order instead (the operator decides
 * work-order granularity, RelationalOperator.hpp:117-119), every block keeping its own stripes.
 *   block_rows     host array [num_blocks]: rows of each block (0 allowed)
 *   block_cols     host array [num_blocks * num_columns]: device stripe of column c of block b at [b * num_columns + c]
 *   block_filters  host array [num_blocks] of TupleIdSequence bitmaps (bit 0 = the block's first row; NULL entry = every
 *                  row of that block), or NULL: no block has one
 * States over compressed or nullable attributes: QSX_ERR_UNSUPPORTED (their per-block dictionaries / null bitmaps go
 * through qsx_agg_update_coded / qsx_agg_update_nullable, one call per block). */
int qsx_agg_update_blocks(qsx_agg_state_t *state, int num_blocks, const int64_t *block_rows, const void *const *block_cols,
                          const uint64_t *const *block_filters, qsx_stream_t stream);

/* qsx_agg_update_coded over a run of blocks in one launch: every block its own code stripes / value stripes and its own
 * dictionaries (the reference compresses block by block: CompressedBlockBuilder picks dictionary or truncation per block and
 * attribute, storage/CompressedBlockBuilder.cpp:120-260 — the state's column_code_width fixes the code width of a column for
 * all blocks it is handed; blocks that chose differently go through another state).
 *   block_cols           [b * num_columns + c]: code stripe where column_code_width[c] != 0, else the value stripe
 *   block_dictionaries   [b * num_columns + c]: dictionary of block b's column c (NULL: truncated or uncompressed) */
int qsx_agg_update_coded_blocks(qsx_agg_state_t *state, int num_blocks, const int64_t *block_rows, const void *const *block_cols,
                                const void *const *block_dictionaries, const uint64_t *const *block_filters, qsx_stream_t stream);

/* qsx_agg_update on a block with NULLs: null_bitmaps_dev[c] is the null bitmap of column c (TupleIdSequence bit
 * order, bit i set = tuple i is NULL, n bits; storage/BasicColumnStoreTupleStorageSubBlock.cpp:131-147 keeps one per
 * nullable attribute) or NULL when the block holds no NULL in that attribute; entries of columns not declared
 * column_nullable must be NULL.  qsx_agg_update on such a state = a block without NULLs. */
int qsx_agg_update_nullable(qsx_agg_state_t *state, const void *const *cols, const uint64_t *const *null_bitmaps_dev,
                            int64_t n, const uint64_t *filter_dev, qsx_stream_t stream);

/* qsx_agg_update on a block whose attributes with column_code_width != 0 are compressed: cols[c] is the code
 * stripe, dictionaries_dev[c] the block's dictionary for that attribute (values of the column's type, indexed by
 * code; storage/CompressedTupleStorageSubBlock.hpp:60-110 compression dictionaries) or NULL when the attribute is
 * truncation-compressed (value = code).  Replaces the per-value decode of CompressedColumnStoreValueAccessor::
 * getUntypedValue (storage/CompressedColumnStoreValueAccessor.hpp:90-150) in front of aggregateBlock: HBM is read
 * at the code width (Q1 over lineitem's compressed quantity / discount / tax: 13 instead of 34 bytes per row).
 * Entries of dictionaries_dev for plain columns are ignored; dictionaries_dev may be NULL when no coded column
 * uses a dictionary.
 * The same mechanism reads a column THROUGH a list of row numbers: declare code width 4, pass the row numbers (the
 * probe or build tids of a join) as the "codes" and the column itself as the "dictionary" — an aggregation right
 * behind a join then never materialises the join's output (Scalar::getAllValuesForJoin + bulkInsertTuples,
 * relational_operators/HashJoinOperator.cpp:529-541, followed by aggregateBlock). */
int qsx_agg_update_coded(qsx_agg_state_t *state, const void *const *cols, const void *const *dictionaries_dev,
                         int64_t n, const uint64_t *filter_dev, qsx_stream_t stream);

/* qsx_agg_update_coded with the dictionaries' sizes: dictionary_entries[c] = number of entries of dictionaries_dev[c] (0 or a
 * NULL array: unknown — the call behaves exactly like qsx_agg_update_coded).  A reference block knows them
 * (compression/CompressionDictionary.hpp:46-58: the dictionary begins with its number of codes).  With the size in hand the
 * plan shapes copy dictionaries of up to 64 entries into LDS once per workgroup and decode from there instead of through the
 * vector memory path (Q1 over lineitem: quantity 50, discount 11, tax 9 entries).  A code >= dictionary_entries[c] is
 * outside the contract (the unsized call would read past the dictionary): it decodes to 0 from the LDS copy.  A dictionary
 * with a NULL code (= num_codes, compression/CompressionDictionary.hpp:49-52) is passed with num_codes entries and the
 * attribute's NULL rows masked by its null bitmap / the filter, as the host layer does. */
int qsx_agg_update_coded_sized(qsx_agg_state_t *state, const void *const *cols, const void *const *dictionaries_dev,
                               const int32_t *dictionary_entries, int64_t n, const uint64_t *filter_dev, qsx_stream_t stream);

/* qsx_agg_update_coded_blocks with every block's dictionary sizes: block_dictionary_entries[b * num_columns + c] = number of
 * entries of block_dictionaries[b * num_columns + c] (0: unknown or no dictionary; a NULL array: exactly
 * qsx_agg_update_coded_blocks).  The reference builds a dictionary PER BLOCK (storage/CompressedBlockBuilder.cpp:300-368), so the
 * codes of one attribute mean different values from block to block: a state whose aggregates factor through the dictionary codes
 * (see qsx_agg_update_coded_sized; Q1 over lineitem: 13 B/row at 0.6 of the HBM peak instead of 0.45) keeps per-block
 * coefficient tables and settles its per-code counts at every block boundary.  Everything else behaves like the unsized call. */
int qsx_agg_update_coded_blocks_sized(qsx_agg_state_t *state, int num_blocks, const int64_t *block_rows, const void *const *block_cols,
                                      const void *const *block_dictionaries, const int32_t *block_dictionary_entries,
                                      const uint64_t *const *block_filters, qsx_stream_t stream);

/* BuildAggregationExistenceMapWorkOrder::execute (relational_operators/
 * BuildAggregationExistenceMapOperator.cpp:50-67, 177-208): sets the existence bit of every (selected)
 * key of a block in a COLLISION_FREE state, without touching the aggregate states — the left side of a
 * CrossReferenceCoalesceAggregate (query_optimizer/ExecutionGenerator.cpp:2054-2210: left outer join +
 * group-by fused; keys without right-side rows finalize as COUNT 0 / SUM 0,
 * CollisionFreeVectorTable.hpp:700-727).  key_type QSX_INT or QSX_LONG; other strategies:
 * QSX_ERR_UNSUPPORTED. */
int qsx_agg_mark_existence(qsx_agg_state_t *state, int key_type, const void *keys_dev, int64_t n,
                           const uint64_t *filter_dev, qsx_stream_t stream);

/* dst += src (same config; the two tables may have grown to different capacities).  Counterpart of
 * ThreadPrivateCompactKeyHashTable::mergeFrom (.cpp:306-363) and
 * AggregationOperationState::mergeGroupByHashTables (.cpp:831-843).  Synchronises on `stream` (the source is
 * brought to rest first). */
int qsx_agg_merge(qsx_agg_state_t *dst, qsx_agg_state_t *src, qsx_stream_t stream);

/* Raw partial state for transport between GPUs (RCCL all-gather /
 * reduce-scatter of partial aggregates).  Layout per strategy is described in
 * DESIGN.md.  The image of a hash strategy grows with its table: ask for the size (synchronises on `stream`, folds
 * spilled rows in) right before exporting; the size stays valid until the next update / merge into the state.
 * COLLISION_FREE and SINGLE_STATE images never change size. */
int qsx_agg_state_export_bytes(qsx_agg_state_t *state, size_t *out_bytes, qsx_stream_t stream);
/* How the image is laid out, for whoever reduces images across GPUs column by column (reduce-scatter of the dense
 * CollisionFreeVector state, storage/CollisionFreeVectorTable.hpp:192-208's key ranges): the image is `header_words`
 * 8-byte words (dense: the LSB-first existence bits; hash strategies: the key slots) followed by `num_columns` columns of
 * `words_per_column` words.  column_kinds[c] says how two partial values of column c combine:
 * QSX_ACC_SUM_F64 (0) f64 +, QSX_ACC_SUM_I64 (1) int64 +, QSX_ACC_MIN_I64 (2), QSX_ACC_MAX_I64 (3) on the int64 words
 * (MIN / MAX of doubles are stored as order-preserving int64 images).  Host-only, no device work. */
#define QSX_ACC_SUM_F64 0
#define QSX_ACC_SUM_I64 1
#define QSX_ACC_MIN_I64 2
#define QSX_ACC_MAX_I64 3
int qsx_agg_state_image_layout(qsx_agg_state_t *state, int *out_dense, int64_t *out_header_words, int64_t *out_words_per_column,
                               int *out_num_columns, int32_t *out_column_kinds, int kinds_capacity);
/* QSX_ERR_CAPACITY when the image no longer fits capacity_bytes. */
int qsx_agg_state_export(qsx_agg_state_t *state, void *out_dev, size_t capacity_bytes, qsx_stream_t stream);
/* dst += exported image (image_bytes long) of a state with the same config.  Stream-ordered. */
int qsx_agg_state_import_merge(qsx_agg_state_t *dst, const void *image_dev, size_t image_bytes, qsx_stream_t stream);

/* Upper bound on the number of groups a finalize of partition p can emit.
 * Synchronises on `stream`. */
int qsx_agg_num_groups(qsx_agg_state_t *state, int64_t *out_groups, qsx_stream_t stream);

/* K10.  Emit one row per group of finalize-partition `partition` of
 * `num_partitions`: key columns (width = the key column's width) and one
 * value column per aggregate (int64 for COUNT and SUM over INT/LONG, double for
 * the other SUMs and AVG; a MIN/MAX column has the argument's own type and
 * width, DOUBLE for an expression).  Replaces AggregationOperationState::finalizeAggregate
 * (storage/AggregationOperationState.cpp:641-948), incl.
 * CollisionFreeVectorTable::finalizeKey/finalizeState (.hpp:647-727; ascending
 * key order, partition = contiguous key range) and
 * ThreadPrivateCompactKeyHashTable::finalize (.cpp:365-421).
 *   out_null_dev  optional array of num_aggs byte columns; 1 = NULL result
 *                 (SUM/AVG over zero rows in SINGLE_STATE, AggregationHandleSum.cpp:45-120)
 *   out_groups_dev int64 on device: rows written — or QSX_GROUPS_HASH_COLLISION: the state groups by a key wider
 *                 than 8 bytes and two different keys shared their 64-bit hash (every group carries MIN and MAX of its
 *                 packed key words, MIN != MAX somewhere proves it).  Nothing of the output may be used then; the
 *                 caller reports QSX_ERR_HASH_COLLISION / re-runs the operator.  Expected once in ~2^64 / groups^2 states. */
#define QSX_GROUPS_HASH_COLLISION (-1ll)
int qsx_agg_finalize(qsx_agg_state_t *state, int partition, int num_partitions,
                     void *const *out_key_cols, void *const *out_val_cols,
                     uint8_t *const *out_null_cols, int64_t capacity,
                     int64_t *out_groups_dev, qsx_stream_t stream);

/* ======================================================================
 * LIP filters (utility/lip_filter/)
 * ====================================================================== */

typedef struct qsx_lip_filter qsx_lip_filter_t;

typedef enum qsx_lip_kind {
  QSX_LIP_SINGLE_IDENTITY_HASH = 0, /* bit = value % cardinality (SingleIdentityHashFilter.hpp:156-169) */
  QSX_LIP_BITVECTOR_EXACT = 1       /* bit = value - min, out of range = miss (BitVectorExactFilter.hpp:150-176) */
} qsx_lip_kind_t;

/* cardinality: filter bits (IDENTITY_HASH: max(64, 8*est build cardinality);
 * EXACT: max - min + 1).  min_value only used by EXACT.  At most 2^32 - 2 bits
 * (QSX_ERR_UNSUPPORTED beyond: bit positions are 32-bit in the probe kernel).
 * Synchronises. */
int qsx_lip_filter_create(int kind, int64_t cardinality, int64_t min_value, int is_anti,
                          qsx_lip_filter_t **out);
int qsx_lip_filter_destroy(qsx_lip_filter_t *f);
/* LIPFilterBuilder::insertValueAccessor (BuildHashOperator.cpp:187-190). */
int qsx_lip_build(qsx_lip_filter_t *f, int key_type, const void *keys_dev, int64_t n,
                  const uint64_t *filter_dev, qsx_stream_t stream);
/* LIPFilterAdaptiveProber::filterValueAccessor (LIPFilterAdaptiveProber.hpp:83-90):
 * out = in AND hit(filter, key).  in_bitmap_dev may be NULL (= all rows). */
int qsx_lip_probe(const qsx_lip_filter_t *f, int key_type, const void *keys_dev, int64_t n,
                  const uint64_t *in_bitmap_dev, uint64_t *out_bitmap_dev,
                  int64_t *out_count_dev, qsx_stream_t stream);
/* qsx_lip_build / qsx_lip_probe over a run of blocks in one launch each (every block its own key stripe and bitmaps): what
 * num_blocks calls produce.  LIPFilterBuilder::insertValueAccessor / LIPFilterAdaptiveProber::filterValueAccessor run once
 * per block inside the work orders (relational_operators/BuildHashOperator.cpp:187-190, SelectOperator.cpp:170-180); a work
 * order over a run of blocks (DESIGN.md "Work-order granularity") calls these instead.
 *   block_filters / block_in_bitmaps   NULL, or host arrays of device pointers (entries may be NULL = every row)
 *   block_out_bitmaps                  host array of device pointers, (block_rows[b]+63)/64 words each, fully overwritten
 *   out_count_dev                      optional: set bits of all output bitmaps together */
int qsx_lip_build_blocks(qsx_lip_filter_t *filter, int key_type, int64_t num_blocks, const int64_t *block_rows,
                         const void *const *block_keys, const uint64_t *const *block_filters, qsx_stream_t stream);
int qsx_lip_probe_blocks(const qsx_lip_filter_t *filter, int key_type, int64_t num_blocks, const int64_t *block_rows,
                         const void *const *block_keys, const uint64_t *const *block_in_bitmaps,
                         uint64_t *const *block_out_bitmaps, int64_t *out_count_dev, qsx_stream_t stream);

/* qsx_lip_build_blocks / qsx_lip_probe_blocks over compressed key stripes (qsx_key_coding_t, above). */
int qsx_lip_build_blocks_coded(qsx_lip_filter_t *filter, int key_type, int64_t num_blocks, const int64_t *block_rows,
                               const void *const *block_keys, const qsx_key_coding_t *coding, const uint64_t *const *block_filters,
                               qsx_stream_t stream);
int qsx_lip_probe_blocks_coded(const qsx_lip_filter_t *filter, int key_type, int64_t num_blocks, const int64_t *block_rows,
                               const void *const *block_keys, const qsx_key_coding_t *coding, const uint64_t *const *block_in_bitmaps,
                               uint64_t *const *block_out_bitmaps, int64_t *out_count_dev, qsx_stream_t stream);

/* An EXACT filter over the key a directly addressed join table was built on, taken from the table: the filter of a BuildHash work
 * order is built from the same keys as its table (relational_operators/BuildHashOperator.cpp:187-203), and such a table is an
 * existence map of its key range already — the bits are read off its head words (4 bytes streamed per key value of the range)
 * instead of one atomic per key.  Sets the bit of EVERY key the table holds by the time the stream gets here (a bit set twice does
 * no harm).  num_new_keys: how many keys the caller would hand qsx_lip_build instead (< 0: do it whatever it costs).
 * QSX_ERR_UNSUPPORTED — nothing was done, call qsx_lip_build — for a hashed table, a table over a strided key domain, a filter
 * that is not LIP_BITVECTOR_EXACT, or when reading the range costs more than num_new_keys atomics would. */
int qsx_lip_build_from_join_table(qsx_lip_filter_t *filter, qsx_join_table_t *table, int64_t num_new_keys, qsx_stream_t stream);

/* Raw bit array (for all-reduce(OR) across GPUs): 64-bit words, LSB-first. */
int qsx_lip_filter_words(qsx_lip_filter_t *f, uint64_t **out_words_dev, int64_t *out_num_words);

/* qsx_join_probe with the work order's LIP filters tested INSIDE the probe: HashInnerJoinWorkOrder::execute runs its
 * LIPFilterAdaptiveProber over the probe accessor first and probes the hash table with the survivors
 * (relational_operators/HashJoinOperator.cpp:450-470, utility/lip_filter/LIPFilterAdaptiveProber.hpp:113-228).  A row is probed
 * when it is set in filter_dev (NULL: every row) AND every filter reports a hit for its key; the pairs are those of
 * qsx_lip_probe (per filter) + qsx_join_probe under the resulting bitmap — from one pass over the keys when the table is
 * directly addressed (or answers from its shadow) and num_lip <= 2, else from exactly that sequence inside the call.
 *   lip_filters   host array of num_lip filters over the probe key (QSX_LIP_*; built before the probe: pipeline breaker) */
int qsx_join_probe_lip(qsx_join_table_t *table, const void *keys_dev, int64_t n, int32_t probe_base_tid,
                       const uint64_t *filter_dev, int num_lip, const qsx_lip_filter_t *const *lip_filters,
                       int32_t *out_probe_tid_dev, int32_t *out_build_tid_dev, int64_t capacity,
                       int64_t *out_count_dev, qsx_stream_t stream);
/* The semi join's form of qsx_join_probe_lip: out_bitmap bit i = filter[i] AND every LIP filter passes key i AND key i is in
 * the table (HashSemiJoinWorkOrder under its LIPFilterAdaptiveProber, relational_operators/HashJoinOperator.cpp:795-816 with
 * :462-470) — one pass over the key stripe where qsx_lip_probe + qsx_join_probe_exists make two. */
int qsx_join_probe_exists_lip(qsx_join_table_t *table, const void *keys_dev, int64_t n, const uint64_t *filter_dev, int num_lip,
                              const qsx_lip_filter_t *const *lip_filters, uint64_t *out_bitmap_dev, int64_t *out_count_dev,
                              qsx_stream_t stream);


/* ======================================================================
 * Hash partitioning (multi-GPU shuffle, partitioned aggregation)
 * ====================================================================== */

/* Partition id of a key = HashPartitionSchemeHeader::getPartitionId
 * (catalog/PartitionSchemeHeader.hpp:200-214) with the identity hash of
 * types/TypedValue.hpp:575-592: h = zero-extended bit pattern;
 * pid = P power of two ? h & (P-1) : (h >= P ? h % P : h). */
size_t qsx_partition_workspace_bytes(int64_t n, int num_partitions);

/* K9.  Stable scatter of n rows into num_partitions contiguous regions by the
 * partition id of keys[i]; moves `ncols` payload columns (the key column may
 * be one of them).  out_offsets_dev receives num_partitions + 1 int64 row
 * offsets (exclusive prefix of partition sizes).  Counterpart of
 * PartitionAwareInsertDestination routing (storage/InsertDestination.hpp:490-660). */
int qsx_partition_scatter(int key_type, const void *keys_dev, int64_t n, int num_partitions,
                          int ncols, const void *const *cols, const int32_t *widths,
                          void *const *out_cols, int64_t *out_offsets_dev,
                          void *workspace_dev, size_t workspace_bytes, qsx_stream_t stream);

/* K9 over a run of storage blocks: what qsx_partition_scatter leaves for the blocks' rows laid end to end (block 0's rows, then
 * block 1's, ...) without laying them so — the repartitioning Select in front of a partitioned join reads a stored relation
 * (SelectOperator.cpp:83-150 one work order per block, every tuple routed by PartitionAwareInsertDestination::bulkInsertTuples,
 * storage/InsertDestination.hpp:560-660); here a work order takes a run of blocks and the scatter reads every block's stripes
 * where they lie.  block_keys[b]: the key stripe of block b (INT / LONG values); block_cols[b * ncols + c]: column c's stripe of
 * block b; out_cols[c]: one stripe with room for all rows.  Blocks without rows are skipped (their pointers are not looked at).
 * Workspace: qsx_partition_blocks_workspace_bytes(all rows, num_blocks, num_partitions). */
size_t qsx_partition_blocks_workspace_bytes(int64_t n, int64_t num_blocks, int num_partitions);
int qsx_partition_scatter_blocks(int key_type, int64_t num_blocks, const int64_t *block_rows, const void *const *block_keys,
                                 int num_partitions, int ncols, const void *const *block_cols, const int32_t *widths,
                                 void *const *out_cols, int64_t *out_offsets_dev, void *workspace_dev, size_t workspace_bytes,
                                 qsx_stream_t stream);

/* ======================================================================
 * ORDER BY (the step after the aggregate in Q1 / Q3)
 * ====================================================================== */

size_t qsx_sort_workspace_bytes(int64_t n);

/* Stable sort permutation: out_tids[j] = row number (0-based) of the j-th row in ORDER BY order over
 * nkeys key columns (key 0 most significant; descending[k] != 0 = DESC; INT / LONG / FLOAT / DOUBLE).
 * Replaces the comparator sort of SortRunGenerationWorkOrder::execute (relational_operators/
 * SortRunGenerationOperator.cpp:88-105 -> StorageBlock::sort, storage/StorageBlock.cpp:561-640, ordering
 * from utility/SortConfiguration.hpp:51-130) and — applied to the concatenation of the runs, truncated
 * to top_k by the caller — the merge tree of SortMergeRunOperator (relational_operators/
 * SortMergeRunOperatorHelpers.cpp).  Rows with equal keys keep their input order.  The caller
 * materialises columns with qsx_gather(col, out_tids).  NULL ordering is out of scope (no NULL inputs). */
int qsx_sort_permutation(int nkeys, const void *const *key_cols, const int32_t *key_types,
                         const int32_t *descending, int64_t n, int32_t *out_tids_dev,
                         void *workspace_dev, size_t workspace_bytes, qsx_stream_t stream);

/* ORDER BY ... LIMIT k: the first min(k, n) entries of qsx_sort_permutation's output, written to
 * out_tids_dev (k entries), same tie order.  Replaces the top_k path of SortMergeRunOperator
 * (relational_operators/SortMergeRunOperator.hpp:92-118 `top_k`, SortMergeRunOperatorHelpers.cpp: the
 * merge stops after top_k tuples).  For k << n a histogram of the leading bits of key 0 selects the
 * candidate rows and only those are sorted.  Workspace = qsx_sort_workspace_bytes(n).  Synchronises the
 * stream once (candidate count). */
int qsx_sort_top_k(int nkeys, const void *const *key_cols, const int32_t *key_types,
                   const int32_t *descending, int64_t n, int64_t k, int32_t *out_tids_dev,
                   void *workspace_dev, size_t workspace_bytes, qsx_stream_t stream);

/* Distinctify: out_tids = row number of the first occurrence of every distinct tuple over ncols columns
 * (QSX_INT / LONG / FLOAT / DOUBLE, or 1-byte QSX_CHAR; at most QSX_MAX_KEYS), restricted to the rows of
 * filter_dev when non-NULL, listed in tuple order; *out_count_dev = number of distinct tuples.
 * Replaces the distinctify hash table of a DISTINCT aggregate — AggregationOperationState.cpp:172-207
 * (one table per DISTINCT aggregate, key = group-by values + argument), filled per block by
 * AggregationConcreteHandle::insertValueAccessorIntoDistinctifyHashTable (AggregationConcreteHandle.hpp:
 * 120-140, called at AggregationOperationState.cpp:522-528, 600-628) and drained at finalize by
 * aggregateOnDistinctifyHashTableFor{Single,GroupBy} (:652-670, :720-760) — by sort + run heads: the caller
 * gathers the tuples (qsx_gather) and feeds them to qsx_agg_update of the state that computes the
 * aggregate over distinct values.  Sort keys of type QSX_CHAR (1 byte) are also accepted by
 * qsx_sort_permutation / qsx_sort_top_k.  Workspace = qsx_sort_workspace_bytes(n).  Synchronises the stream. */
int qsx_distinct_rows(int ncols, const void *const *cols, const int32_t *types, int64_t n,
                      const uint64_t *filter_dev, int32_t *out_tids_dev, int64_t *out_count_dev,
                      void *workspace_dev, size_t workspace_bytes, qsx_stream_t stream);

/* ---------------------------------------------------------------------------
 * Multi-GPU: one process per GPU, GPU g = hash partition g of P = world (catalog/PartitionSchemeHeader.hpp:200-214),
 * RCCL over xGMI bound at run time (librccl.so.1).  The reference has no data-plane collective — its partitions share an
 * address space (storage/InsertDestination.hpp:490-660, BuildHashOperator.cpp:82-91, HashJoinOperator.cpp:220-231) — so
 * these are the two exchange steps the path gains when a partition becomes a GPU:
 *   join-key shuffle   qsx_partition_scatter (K9) -> qsx_exchange_counts -> qsx_alltoallv per column -> local build / probe
 *   partial aggregates qsx_agg_reduce_scatter (dense states) / qsx_agg_allgather_merge (hash states) before finalize
 * plus qsx_bitmap_allreduce_or for LIP filters (qsx_lip_filter_words) and qsx_allgather for broadcast build sides.
 * Bootstrap: rank 0 calls qsx_comm_unique_id, the engine carries the 128 bytes to the other processes (its own control
 * plane: TMB messages in the reference's distributed mode), every rank calls qsx_comm_create on its own device.
 * All collectives are stream-ordered; counts arrays of qsx_alltoallv are HOST arrays (rows per peer).
 * --------------------------------------------------------------------------- */
#define QSX_COMM_ID_BYTES 128
typedef struct qsx_comm qsx_comm_t;
int qsx_comm_unique_id(void *out_id_bytes);
int qsx_comm_create(int world, int rank, const void *id_bytes, qsx_comm_t **out);
int qsx_comm_destroy(qsx_comm_t *comm);
int qsx_comm_rank(const qsx_comm_t *comm, int *out_world, int *out_rank);
/* Failure agreement.  A rank that fails on its own between two collectives (an allocation, a validation) must not leave
 * its peers inside a collective nobody else will enter: before the first collective of a step every rank contributes the
 * status of its local preparation, and ALL ranks get the same verdict — QSX_OK when every rank contributed QSX_OK, else
 * the rank's own status when that is the failure, QSX_ERR_COMM ("rank r failed ...", qsx_last_error) on the others.  A
 * collective itself (one word per rank through a buffer the communicator owns: nothing is allocated); synchronises.
 * qsx_agg_reduce_scatter, qsx_agg_allgather_merge and qsx_bitmap_allreduce_or agree this way on their own scratch. */
int qsx_comm_agree(qsx_comm_t *comm, int local_status, qsx_stream_t stream);
/* Wait for `stream` under the communicator's watchdog (QSX_COMM_TIMEOUT_MS, default 600 000; 0 = wait for ever): a
 * collective whose peers never arrive does not block the caller indefinitely — at the deadline the communicator is
 * aborted (ncclCommAbort) and QSX_ERR_COMM returned; every later call on it returns QSX_ERR_COMM at once.  A transport
 * that has no ncclCommAbort gets no watchdog (as with 0): without the abort the stalled kernels stay on the stream and the
 * deadline would only move the stall into the next wait. */
int qsx_comm_synchronize(qsx_comm_t *comm, qsx_stream_t stream);
/* Abort the communicator (a rank that failed inside a step tells RCCL to give up instead of waiting in its kernels). */
int qsx_comm_abort(qsx_comm_t *comm);
/* recv_counts_dev[p] = send_counts_dev[p] of rank p's call (int64[world] device arrays): the row counts of a shuffle */
int qsx_exchange_counts(qsx_comm_t *comm, const int64_t *send_counts_dev, int64_t *recv_counts_dev, qsx_stream_t stream);
/* Rows of `width` bytes: send_rows[p] rows go to rank p (taken back to back from send_dev in rank order — the layout
 * qsx_partition_scatter leaves), recv_rows[p] rows arrive from rank p (placed back to back in rank order). */
int qsx_alltoallv(qsx_comm_t *comm, int width, const void *send_dev, const int64_t *send_rows, void *recv_dev,
                  const int64_t *recv_rows, qsx_stream_t stream);
/* recv_dev = the `bytes` of every rank, in rank order (world * bytes). */
int qsx_allgather(qsx_comm_t *comm, const void *send_dev, size_t bytes, void *recv_dev, qsx_stream_t stream);
/* words |= the words of every other rank (LIP bit vectors; RCCL has no bitwise reduction: gather + local OR). */
int qsx_bitmap_allreduce_or(qsx_comm_t *comm, uint64_t *words_dev, int64_t num_words, qsx_stream_t stream);
/* COLLISION_FREE state: afterwards this rank holds the MERGED groups of finalize partition `rank` of `world` and nothing
 * else — finalize with qsx_agg_finalize(state, rank, world, ...).  Replaces the shared atomics of
 * CollisionFreeVectorTable (storage/CollisionFreeVectorTable.hpp:530-645) across address spaces. */
int qsx_agg_reduce_scatter(qsx_comm_t *comm, qsx_agg_state_t *state, qsx_stream_t stream);
/* The key range [begin, end) that finalize partition `partition` of `num_partitions` owns in a COLLISION_FREE state of
 * `num_entries` keys (storage/CollisionFreeVectorTable.hpp:192-208: contiguous ranges of ceil(entries / partitions) keys),
 * the LSB-first existence words [first_word, last_word) covering it (both 0 for an empty range) and the masks that cut the
 * range out of its first and last word.  qsx_agg_finalize and qsx_agg_reduce_scatter split by this very function; a
 * caller that moves image ranges itself (quickstep_amd/distributed.py over torch.distributed) takes the split from here
 * instead of restating it.  Host arithmetic only: needs no device; any out pointer may be NULL. */
int qsx_agg_dense_partition_range(int64_t num_entries, int num_partitions, int partition, int64_t *out_begin, int64_t *out_end,
                                  int64_t *out_first_word, int64_t *out_last_word, uint64_t *out_first_mask,
                                  uint64_t *out_last_mask);
/* Hash-strategy state: afterwards every rank holds the whole merged table (AggregationOperationState.cpp:925-948's merge
 * of the thread-private tables, across ranks).  Synchronises on `stream`. */
int qsx_agg_allgather_merge(qsx_comm_t *comm, qsx_agg_state_t *state, qsx_stream_t stream);

#ifdef __cplusplus
} /* extern "C" */
#endif

#endif /* QSX_H_ */
