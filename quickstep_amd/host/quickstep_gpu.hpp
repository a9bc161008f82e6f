// quickstep_gpu.hpp — host side of the drop-in: the reference's operator /
// work-order interface for the hot path, with work orders that call the HIP
// execution kernel through the C ABI (include/qsx.h) and nothing else.
//
// Mirrors (same names, argument meaning and call protocol; paths in the Quickstep tree):
//   RelationalOperator            relational_operators/RelationalOperator.hpp:55-334
//   WorkOrder                     relational_operators/WorkOrder.hpp:53-335
//   WorkOrdersContainer           query_execution/WorkOrdersContainer.hpp:243
//   QueryContext                  query_execution/QueryContext.hpp:190-451
//   SelectOperator/WorkOrder      relational_operators/SelectOperator.hpp:69-386
//   BuildHashOperator/WorkOrder   relational_operators/BuildHashOperator.hpp:66-277
//   HashJoinOperator/WorkOrder    relational_operators/HashJoinOperator.hpp:66-438
//   AggregationOperator           relational_operators/AggregationOperator.hpp:53-187
//   FinalizeAggregationOperator   relational_operators/FinalizeAggregationOperator.hpp:52-163
//   DestroyHashOperator, DestroyAggregationStateOperator
//   ForemanSingleNode / Worker    query_execution/ForemanSingleNode.cpp:102-178, Worker.cpp:54-139
// What is NOT rebuilt: catalog persistence, buffer manager, TMB, protobuf
// (de)serialisation, optimizer.  CatalogRelation / StorageManager /
// InsertDestination here are the minimum those operators need: device-resident
// column-store blocks.
#ifndef QUICKSTEP_AMD_HOST_QUICKSTEP_GPU_HPP_
#define QUICKSTEP_AMD_HOST_QUICKSTEP_GPU_HPP_

#include <atomic>
#include <condition_variable>
#include <cstddef>
#include <cstdint>
#include <cstring>
#include <deque>
#include <functional>
#include <memory>
#include <mutex>
#include <stdexcept>
#include <string>
#include <thread>
#include <unordered_map>
#include <vector>

#include "../../include/qsx.h"

namespace tmb {
typedef std::uint32_t client_id;  // stand-in: scheduler messages never carry data
class MessageBus {};
}  // namespace tmb

namespace quickstep {

typedef int attribute_id;
typedef int tuple_id;
typedef int relation_id;
typedef std::uint64_t block_id;
typedef std::size_t partition_id;
typedef int numa_node_id;

constexpr attribute_id kInvalidAttributeID = -1;

// types/TypeID.hpp:32-43 (numbering shared with qsx_type_t)
enum TypeID { kInt = 0, kLong = 1, kFloat = 2, kDouble = 3, kChar = 4, kVarChar = 5, kDate = 6 };

// types/DatetimeLit.hpp:38-90: the value of a DATE attribute, 8 bytes in a column stripe; ordered by year, month, day.
struct DateLit {
  std::int32_t year;
  std::uint8_t month, day;
  std::uint8_t unused[2];
  static DateLit Create(std::int32_t year, std::uint8_t month, std::uint8_t day) { return DateLit{year, month, day, {0, 0}}; }
  bool operator<(const DateLit &r) const { return year != r.year ? year < r.year : (month != r.month ? month < r.month : day < r.day); }
  bool operator==(const DateLit &r) const { return year == r.year && month == r.month && day == r.day; }
};
static_assert(sizeof(DateLit) == 8, "DateLit occupies 8 bytes");
// types/operations/comparisons/ComparisonID.hpp:36-42
enum class ComparisonID { kEqual = 0, kNotEqual, kLess, kLessOrEqual, kGreater, kGreaterOrEqual };
// expressions/aggregation/AggregationID.hpp
enum class AggregationID { kCount, kSum, kAvg, kMin, kMax };

struct Type {
  TypeID id;
  int width;  // bytes
  bool nullable = false;  // blocks keep a null bitmap for the attribute (BasicColumnStoreTupleStorageSubBlock.cpp:131-147)
  static Type Int() { return {kInt, 4}; }
  static Type Long() { return {kLong, 8}; }
  static Type Float() { return {kFloat, 4}; }
  static Type Double() { return {kDouble, 8}; }
  static Type Char(int n) { return {kChar, n}; }
  static Type Date() { return {kDate, 8}; }
  Type getNullableVersion() const { Type t = *this; t.nullable = true; return t; }
};

// Error raised when the C ABI reports a failure: the reference aborts with
// LOG(FATAL) on invariant violations (BuildHashOperator.cpp:205-206); an
// exception keeps test processes alive.
class ExecutionError : public std::runtime_error {
 public:
  ExecutionError(const std::string &where, int status);
  int status() const { return status_; }
 private:
  int status_;
};
void CheckStatus(int status, const char *where);

// CPU plumbing mode (BASELINE config 1: "CPU reference WorkOrder, no GPU"): blocks are kept in
// host memory.  Set before any block is created; GPU work orders must not be used in this mode.
void UseHostMemoryForBlocks(bool on);

// The stream of the calling Worker thread (nullptr = default stream).
qsx_stream_t CurrentStream();
void SetCurrentStream(qsx_stream_t stream);

// ---------------------------------------------------------------------------
// catalog + storage (minimum)
// ---------------------------------------------------------------------------
class CatalogRelation {
 public:
  CatalogRelation(relation_id id, std::string name) : id_(id), name_(std::move(name)) {}
  attribute_id addAttribute(const std::string &name, Type type);
  relation_id getID() const { return id_; }
  const std::string &getName() const { return name_; }
  std::size_t size() const { return types_.size(); }
  const Type &getAttributeType(attribute_id a) const { return types_.at(a); }
  attribute_id getAttributeByName(const std::string &name) const;
  void addBlock(block_id b);   // unpartitioned relations; partition 0 of partitioned ones
  std::vector<block_id> getBlocksSnapshot() const;
  // HashPartitionSchemeHeader(num_partitions, {attribute}) + PartitionScheme (catalog/PartitionScheme.hpp,
  // PartitionSchemeHeader.hpp:200-214): blocks are registered per partition; the loader routes tuples.
  void setPartitionScheme(std::size_t num_partitions, attribute_id partition_attribute);
  bool hasPartitionScheme() const { return num_partitions_ > 0; }
  std::size_t getNumPartitions() const { return num_partitions_ > 0 ? num_partitions_ : 1; }
  attribute_id getPartitionAttribute() const { return partition_attribute_; }
  void addBlockToPartition(block_id b, partition_id part);
  std::vector<block_id> getBlocksInPartition(partition_id part) const;

 private:
  relation_id id_;
  std::string name_;
  std::vector<std::string> names_;
  std::vector<Type> types_;
  mutable std::mutex mutex_;
  std::vector<block_id> blocks_;
  std::size_t num_partitions_ = 0;  // 0: no partition scheme
  attribute_id partition_attribute_ = kInvalidAttributeID;
  std::vector<std::vector<block_id>> partition_blocks_;
};

// Block ids of an operator's input, per partition, with the per-partition count of work orders
// already generated (input_relation_block_ids_ / num_workorders_generated_ of the reference operators).
struct PartitionedBlockIds {
  std::vector<std::vector<block_id>> ids;
  std::vector<std::size_t> generated;
  explicit PartitionedBlockIds(std::size_t parts) : ids(parts), generated(parts, 0) {}
};

// How one attribute of a compressed column-store block is stored (CompressedBlockBuilder's choice,
// storage/CompressedBlockBuilder.cpp:508-566, 590-650): a stripe of 1/2/4-byte unsigned codes that are the
// values themselves (truncated non-negative INT/LONG) or indexes into a sorted dictionary.
struct CompressedAttribute {
  enum Kind { kUncompressed = 0, kTruncated = 1, kDictionary = 2 };
  Kind kind = kUncompressed;
  int code_width = 0;
  std::uint32_t num_codes = 0;          // dictionary entries
  void *codes = nullptr;                // device: num_tuples codes
  void *dictionary = nullptr;           // device copy of the dictionary (decode)
  std::vector<unsigned char> dictionary_host;  // sorted values, attribute width each (predicate transformation)
  int value_width = 0;                  // CHAR(n) / DATE dictionaries: bytes per entry
};

// Result of rewriting `attribute OP literal` into a comparison on codes
// (CompressedAttributePredicateTransformer::TransformPredicateOnCompressedAttribute,
// storage/CompressedStoreUtil.cpp:51-140, 425-616).
struct PredicateTransformResult {
  enum Type { kAll, kNone, kBasicComparison, kRangeComparison };
  Type type = kNone;
  qsx_code_cmp_t comp = QSX_CODE_EQ;    // kBasicComparison: =, !=, <, >= on codes
  std::uint32_t first_literal = 0, second_literal = 0;
};
struct TypedLiteral;
// CompressedBlockBuilder's per-attribute decision for n host values of a numeric type without NULLs: fills kind,
// code_width, num_codes and dictionary_host of *out and the code stripe (empty when the attribute stays
// uncompressed).  Pure host logic (storage/CompressedBlockBuilder.cpp:508-566, 590-650).
void CompressValues(TypeID type, const void *values, std::int64_t n, CompressedAttribute *out,
                    std::vector<unsigned char> *codes_host, int value_width = 0);   // value_width: CHAR(n) only
PredicateTransformResult TransformPredicateOnCompressedAttribute(const CompressedAttribute &attribute, TypeID type,
                                                                 ComparisonID comparison, const TypedLiteral &literal);

// One device-resident column-store block: a dense stripe per attribute
// (storage/BasicColumnStoreTupleStorageSubBlock.cpp:100-183), optionally compressed per attribute
// (storage/CompressedColumnStoreTupleStorageSubBlock.cpp).
class StorageBlock {
 public:
  StorageBlock(const CatalogRelation &relation, std::int64_t capacity, std::int64_t first_row, bool one_allocation = false);
  // A block that IS rows [first_tuple, first_tuple + num_tuples) of `parent` (one partition of a block that a
  // PartitionAwareInsertDestination has scattered): its stripes point into the parent's, which it keeps alive; null bitmaps
  // of nullable attributes are its own (zeroed; bit positions do not carry over from an unaligned row offset).
  StorageBlock(std::shared_ptr<StorageBlock> parent, std::int64_t first_tuple, std::int64_t num_tuples);
  // A block over memory that belongs to someone else (StorageManager::adoptBlockImage): nothing is freed with the block.
  StorageBlock(const CatalogRelation &relation, std::int64_t num_tuples, const std::vector<void *> &stripes,
               const std::vector<void *> &null_bitmaps);
  ~StorageBlock();
  const CatalogRelation &getRelation() const { return relation_; }
  std::int64_t numTuples() const { return num_tuples_; }
  void setNumTuples(std::int64_t n) { num_tuples_ = n; }
  std::int64_t capacity() const { return capacity_; }
  std::int64_t firstRow() const { return first_row_; }  // relation-global row number of tuple 0
  void setFirstRow(std::int64_t r) { first_row_ = r; }
  // The attribute's VALUES as a dense stripe.  A compressed attribute is decoded on first use (operators
  // that consume values: joins, aggregates, projections); predicates on it never come here, they scan the codes.
  void *stripe(attribute_id a) const;
  void copyAttributeToHost(attribute_id a, void *dst) const;
  const CompressedAttribute *compressedAttribute(attribute_id a) const {
    return compressed_.empty() || compressed_.at(a).kind == CompressedAttribute::kUncompressed ? nullptr : &compressed_.at(a);
  }
  // Replaces attribute a's stripe by its compressed form when CompressedBlockBuilder would (values on the host).
  void compressAttribute(attribute_id a, const void *host_values);
  bool valuesMaterialized(attribute_id a) const { return stripes_.at(a) != nullptr; }
  // An attribute of a block adopted from a reference CompressedColumnStore image (StorageManager::adoptBlockImage): its
  // code stripe and dictionary lie in the image (nothing is copied but the dictionary's host copy for the predicate
  // rewriting); the values are decoded on first use like those of a block this layer compressed itself.
  void adoptCompressedAttribute(attribute_id a, CompressedAttribute attribute);
  // The memory an adopted block points into (never freed with the block; what it allocates later — decoded stripes, the
  // null bitmap made from a dictionary's NULL code — is).
  void setExternalRange(const void *base, std::size_t bytes) { external_base_ = static_cast<const char *>(base); external_bytes_ = bytes; }
  void setNullBitmap(attribute_id a, void *bitmap) { null_bitmaps_.at(a) = bitmap; }
  // Sort column of a sorted column store (TupleStorageSubBlockDescription sort_attribute_id; the tuples of the block are
  // in ascending order of it): predicates on it are evaluated by binary search (predicate_cost::kBinarySearch).
  // kInvalidAttributeID = unsorted.  The loader of the block vouches for the order.
  void setSortColumn(attribute_id a) { sort_column_ = a; }
  attribute_id sortColumn() const { return sort_column_; }
  // Null bitmap of a nullable attribute (TupleIdSequence bit order, 1 = NULL; zeroed at creation),
  // nullptr for non-nullable attributes.
  std::uint64_t *nullBitmap(attribute_id a) const { return static_cast<std::uint64_t *>(null_bitmaps_.at(a)); }
  // dst: (numTuples() + 63) / 64 words; all zero for a non-nullable attribute.
  void copyNullBitmapToHost(attribute_id a, std::uint64_t *dst) const;

 private:
  const CatalogRelation &relation_;
  std::int64_t capacity_;
  std::int64_t num_tuples_;
  std::int64_t first_row_;
  attribute_id sort_column_ = kInvalidAttributeID;
  void *slab_ = nullptr;                         // != nullptr: the one allocation all stripes and null bitmaps live in
  std::size_t slab_bytes_ = 0, slab_granted_ = 0;   // granted: the pool's size class (0: a plain allocation)
  std::shared_ptr<StorageBlock> view_parent_;    // != nullptr: the stripes belong to it
  bool external_memory_ = false;                 // stripes and null bitmaps belong to the caller (an adopted block image)
  const char *external_base_ = nullptr;          // ... and lie in [external_base_, external_base_ + external_bytes_)
  std::size_t external_bytes_ = 0;               //     (0: every pointer of the block is the caller's)
  mutable std::vector<void *> stripes_;          // nullptr: compressed and not decoded yet
  std::vector<void *> null_bitmaps_;
  std::vector<CompressedAttribute> compressed_;  // empty or one per attribute
  mutable std::mutex decode_mutex_;
};
typedef std::shared_ptr<StorageBlock> BlockReference;

// Where the pieces of a reference block image lie (byte offsets from the start of the image).  The image is what the
// reference's StorageManager holds per block (storage/StorageBlock.cpp:81-195):
//   [int32 header length][StorageBlockHeader, protobuf wire format (StorageBlockLayout.proto:96-124)]
//   [tuple store sub-block of header.tuple_store_size bytes][index sub-blocks]
// with a BasicColumnStoreTupleStorageSubBlock as tuple store (storage/BasicColumnStoreTupleStorageSubBlock.cpp:100-183):
//   {int32 num_tuples; int32 nulls_in_sort_column}  (.hpp:188-191)
//   one null bitmap per nullable attribute, BitVector<false>::BytesNeeded(max_tuples) bytes each, MSB-first, 1 = NULL
//   one stripe per attribute at max_tuples * width bytes
//   max_tuples = (tuple_store_size - 8 - nullable_attributes * bitmap_bytes) / sum of the attribute widths   (:131-147)
struct ReferenceBlockLayout {
  std::int64_t num_tuples = 0, max_tuples = 0;
  attribute_id sort_attribute = kInvalidAttributeID;   // BasicColumnStoreTupleStorageSubBlockDescription::sort_attribute_id
  std::size_t tuple_store_offset = 0, tuple_store_size = 0;
  std::vector<std::size_t> null_bitmap_offset;          // per attribute; SIZE_MAX: not nullable / no bitmap in this block
  std::vector<std::size_t> stripe_offset;               // per attribute
  // A CompressedColumnStoreTupleStorageSubBlock (storage/CompressedColumnStoreTupleStorageSubBlock.cpp:755-798,
  // CompressedTupleStorageSubBlock.cpp:281-342): {int32 num_tuples; int32 info bytes; CompressedBlockInfo (protobuf)}, the
  // dictionaries back to back ({uint32 num_codes; uint32 null_code; values}, compression/CompressionDictionary.hpp:46-58),
  // a null bitmap of null_bitmap_bits bits per UNCOMPRESSED attribute that has NULLs, then one stripe per attribute at
  // max_tuples x attribute_size[a] bytes — codes where attribute_size differs from the type's width or a dictionary exists.
  bool compressed = false;
  std::vector<std::size_t> attribute_size;              // bytes per tuple in the stripe (compressed stores)
  std::vector<std::size_t> dictionary_offset;           // per attribute; SIZE_MAX: none.  Offset of the 8-byte dictionary header
  std::vector<std::size_t> dictionary_bytes;
  std::size_t null_bitmap_bits = 0;
};
// Pure host logic: `prefix` = the first prefix_bytes of the image (the block header and the 8-byte sub-block header must lie
// inside), image_bytes = its full size.  Throws ExecutionError(QSX_ERR_INVALID_ARGUMENT) for a malformed image
// (StorageBlock.cpp:108-131 MalformedBlock) and QSX_ERR_UNSUPPORTED for a tuple store that is neither a basic nor a compressed
// column store (row stores), or a compressed store with a variable-length attribute.
ReferenceBlockLayout ParseReferenceBlockImage(const CatalogRelation &relation, const void *prefix, std::size_t prefix_bytes,
                                              std::size_t image_bytes);

class StorageManager {
 public:
  StorageManager() = default;
  // A reference block image that already lies in device memory (the engine's buffer pool in HBM) becomes a block of
  // `relation` IN PLACE: the stripes and null bitmaps of the new block point into the image, nothing is copied; the image
  // belongs to the caller and must outlive the block.  What every kernel entry point then sees is (stripe, num_tuples) —
  // max_tuples only shows in the distance between two stripes.
  block_id adoptBlockImage(CatalogRelation *relation, void *image_dev, std::size_t image_bytes, partition_id part = 0);
  // Create an empty block with room for `capacity` tuples; first_row = rows already in the relation.
  block_id createBlock(CatalogRelation *relation, std::int64_t capacity);
  // Create a block from host columns (one pointer per attribute), copy to HBM and add it to the relation.
  // compress: per attribute, try to store it compressed (the attribute list of the block layout's
  // CompressedColumnStore description, storage/StorageBlockLayout.proto); nullptr = plain column store.
  // null_bitmaps: per attribute, the host null bitmap of a nullable attribute (TupleIdSequence bit order, 1 = NULL,
  // (num_tuples + 63) / 64 words) or nullptr = no NULLs in this block; the vector itself may be nullptr.
  block_id loadBlock(CatalogRelation *relation, const std::vector<const void *> &host_columns, std::int64_t num_tuples,
                     partition_id part = 0, const std::vector<bool> *compress = nullptr,
                     const std::vector<const std::uint64_t *> *null_bitmaps = nullptr);
  // Registers a view of rows [first_tuple, first_tuple + num_tuples) of block `parent` as a block of its own.
  block_id createViewBlock(block_id parent, std::int64_t first_tuple, std::int64_t num_tuples);
  BlockReference getBlock(block_id id) const;
  void deleteBlockOrBlobFile(block_id id);
  // Registers `num_tuples` more rows of `relation`; returns the relation-global row number of the first one.
  std::int64_t reserveRows(relation_id relation, std::int64_t num_tuples);

 private:
  mutable std::mutex mutex_;
  std::unordered_map<block_id, BlockReference> blocks_;
  std::unordered_map<relation_id, std::int64_t> rows_in_relation_;
  block_id next_id_ = 1;
};

// ---------------------------------------------------------------------------
// expressions (attribute-vs-literal comparisons, attribute projections)
// ---------------------------------------------------------------------------
struct TypedLiteral {
  TypeID type;
  union { std::int32_t i32; std::int64_t i64; float f32; double f64; } v;
  static TypedLiteral Int(std::int32_t x) { TypedLiteral l; l.type = kInt; l.v.i64 = 0; l.v.i32 = x; return l; }
  static TypedLiteral Long(std::int64_t x) { TypedLiteral l; l.type = kLong; l.v.i64 = x; return l; }
  static TypedLiteral Float(float x) { TypedLiteral l; l.type = kFloat; l.v.i64 = 0; l.v.f32 = x; return l; }
  static TypedLiteral Double(double x) { TypedLiteral l; l.type = kDouble; l.v.f64 = x; return l; }
  static TypedLiteral Date(std::int32_t year, int month, int day) {   // the DateLit bytes, padding zero
    TypedLiteral l;
    l.type = kDate;
    const DateLit d = DateLit::Create(year, static_cast<std::uint8_t>(month), static_cast<std::uint8_t>(day));
    std::memcpy(&l.v.i64, &d, 8);
    return l;
  }
  // a string literal for a CHAR(n) attribute (compared like the reference's strcmpHelper: AsciiStringComparators.hpp:218-251)
  static TypedLiteral Char(const std::string &text) { TypedLiteral l; l.type = kChar; l.v.i64 = 0; l.text = text; return l; }
  std::string text;
};

// ComparisonPredicate attr OP literal (expressions/predicate/ComparisonPredicate.cpp:115-334);
// a Predicate is a conjunction of those.  As the residual predicate of a join
// (HashJoinOperator.cpp:510-524, Predicate::matchesForJoinedTuples) a term also says which side each
// attribute comes from (Scalar::kLeftSide = probe, kRightSide = build) and may compare two attributes.
struct ComparisonPredicate {
  attribute_id attribute;
  ComparisonID comparison;
  TypedLiteral literal;
  bool on_build_side = false;                          // join residuals only: `attribute` is of the build relation
  attribute_id rhs_attribute = kInvalidAttributeID;    // != invalid: attribute OP rhs_attribute (literal unused)
  bool rhs_on_build_side = false;
  ComparisonPredicate() = default;
  ComparisonPredicate(attribute_id a, ComparisonID c, TypedLiteral l, bool build_side = false)
      : attribute(a), comparison(c), literal(l), on_build_side(build_side) {}
  static ComparisonPredicate Attributes(attribute_id lhs, bool lhs_on_build, ComparisonID c, attribute_id rhs, bool rhs_on_build) {
    ComparisonPredicate p(lhs, c, TypedLiteral::Long(0), lhs_on_build);
    p.rhs_attribute = rhs;
    p.rhs_on_build_side = rhs_on_build;
    return p;
  }
};
struct Predicate {
  std::vector<ComparisonPredicate> conjuncts;
  // TupleIdSequence of the matching tuples of `block` (StorageBlock::getMatchesForPredicate), restricted to
  // `filter` when given.  Returns a device bitmap owned by the caller (qsx_device_free) and the match count.
  void *getMatchesForBlock(const StorageBlock &block, std::int64_t *num_matches, const std::uint64_t *filter = nullptr) const;
};

// ---------------------------------------------------------------------------
// InsertDestination (storage/InsertDestination.hpp:75-340): output sink
// ---------------------------------------------------------------------------
class InsertDestination {
 public:
  InsertDestination(CatalogRelation *relation, StorageManager *storage_manager)
      : relation_(relation), storage_manager_(storage_manager) {}
  // PartitionAwareInsertDestination (storage/InsertDestination.hpp:490-660): every tuple goes to the partition
  // HashPartitionSchemeHeader::getPartitionId names for its value of `partition_attribute`
  // (catalog/PartitionSchemeHeader.hpp:200-214: identity hash of the INT / LONG value, h & (P - 1) for a power of two,
  // else h >= P ? h % P : h).  A work order still fills ONE block; returnBlock scatters it (K9, qsx_partition_scatter) and
  // registers one block per non-empty partition — what the reference's per-partition bulk inserts leave behind.
  InsertDestination(CatalogRelation *relation, StorageManager *storage_manager, std::size_t num_partitions, attribute_id partition_attribute)
      : relation_(relation), storage_manager_(storage_manager), num_partitions_(num_partitions), partition_attribute_(partition_attribute) {}
  bool isPartitionAware() const { return num_partitions_ > 0; }
  std::size_t getNumPartitions() const { return num_partitions_ > 0 ? num_partitions_ : 1; }
  const CatalogRelation &getRelation() const { return *relation_; }
  // A block with room for `capacity` tuples; hand it back with returnBlock once filled.
  BlockReference getBlockForInsertion(std::int64_t capacity, block_id *id);
  // input_partition: the partition of the work order that filled the block — the partition of the output block when the
  // destination does not repartition (the output relation keeps the input's scheme, RelationalOperator.hpp:311-320).
  void returnBlock(block_id id, std::int64_t num_tuples, partition_id input_partition = 0);
  // A partition-aware destination fed by a projection of EVERY tuple of a run of blocks (the repartitioning Select in front of a
  // partitioned join): output attribute i = attribute attributes[i] of the blocks.  K9 reads the blocks' stripes where they lie
  // (qsx_partition_scatter_blocks) — no output block filled first, then scattered.  false: not this case (a destination that
  // does not repartition, nullable attributes, attributes wider than 8 bytes) — nothing was done, fill a block and return it.
  bool insertRunRepartitioned(const std::vector<BlockReference> &blocks, const std::vector<attribute_id> &attributes);
  std::vector<block_id> getTouchedBlocks() const;
  struct TouchedBlock { block_id id; partition_id partition; };
  std::vector<TouchedBlock> getTouchedBlocksWithPartitions() const;
  // The blocks returned since the first `from` (what the Foreman feeds downstream each pass: kDataPipelineMessage).
  std::vector<TouchedBlock> getTouchedBlocksSince(std::size_t from) const;
  std::size_t numTouchedBlocks() const;
  // Called (outside the destination's lock) every time a block has been registered — the reference sends the Foreman a
  // kDataPipelineMessage from here (storage/InsertDestination.cpp:424-470); the Foreman of this layer sleeps until then.
  void setBlockReturnedCallback(std::function<void()> callback) { block_returned_ = std::move(callback); }

 private:
  void repartitionBlock(block_id id, std::int64_t num_tuples);
  CatalogRelation *relation_;
  StorageManager *storage_manager_;
  std::size_t num_partitions_ = 0;                        // 0: not partition-aware
  attribute_id partition_attribute_ = kInvalidAttributeID;
  mutable std::mutex mutex_;
  std::vector<TouchedBlock> touched_;
  std::function<void()> block_returned_;
};
typedef InsertDestination PartitionAwareInsertDestination;   // (one class: the second constructor makes it partition-aware)

// ---------------------------------------------------------------------------
// scalar expressions (expressions/scalar/): ScalarAttribute, ScalarLiteral, ScalarBinaryExpression over the four
// arithmetic operations (types/operations/binary_operations/ArithmeticBinaryOperators.hpp).  Evaluated per row in IEEE
// double, every node rounded on its own — what ScalarBinaryExpression::getAllValues produces for DOUBLE operands
// (ScalarBinaryExpression.cpp:100-195); TPC-H's DECIMAL columns are DOUBLE in the reference (parser/SqlParser.ypp:791-793).
// ---------------------------------------------------------------------------
enum class BinaryOperationID { kAdd = 0, kSubtract, kMultiply, kDivide };
class Scalar;
typedef std::shared_ptr<const Scalar> ScalarPtr;
class Scalar {
 public:
  enum Kind { kAttribute, kLiteral, kBinaryExpression };
  Kind kind = kAttribute;
  attribute_id attribute = kInvalidAttributeID;
  double literal = 0.0;
  TypeID literal_type = kDouble;   // kInt / kLong: an integer literal (ScalarLiteral of an INT / LONG TypedValue)
  BinaryOperationID operation = BinaryOperationID::kAdd;
  ScalarPtr left, right;
  static ScalarPtr Attribute(attribute_id a) { auto s = std::make_shared<Scalar>(); s->kind = kAttribute; s->attribute = a; return s; }
  static ScalarPtr Literal(double v) { auto s = std::make_shared<Scalar>(); s->kind = kLiteral; s->literal = v; return s; }
  // an INT literal when the value fits 32 bits, else LONG (|v| < 2^53: it travels as a double inside the program)
  static ScalarPtr IntLiteral(std::int64_t v) {
    auto s = std::make_shared<Scalar>();
    s->kind = kLiteral;
    s->literal = static_cast<double>(v);
    s->literal_type = v >= INT32_MIN && v <= INT32_MAX ? kInt : kLong;
    return s;
  }
  static ScalarPtr Binary(BinaryOperationID op, ScalarPtr l, ScalarPtr r) {
    auto s = std::make_shared<Scalar>();
    s->kind = kBinaryExpression; s->operation = op; s->left = std::move(l); s->right = std::move(r);
    return s;
  }
};
// The type of a Scalar over `relation` by the reference's rule for arithmetic (BinaryOperation::resultTypeForArgumentTypes,
// types/operations/binary_operations/ArithmeticBinaryOperation.hpp): DOUBLE as soon as a FLOAT / DOUBLE is involved, else LONG
// if a LONG is, else INT.  Integer-typed trees are evaluated in integer arithmetic (qsx_eval_expression_long) by the
// SelectOperator's general form; aggregate arguments are evaluated in double inside the aggregation kernel (exact below 2^53).
TypeID ScalarResultType(const ScalarPtr &scalar, const CatalogRelation &relation);
// Scalar trees flattened into one expression program (qsx_expr_instr_t[]): one instruction per distinct binary node —
// a subexpression shared by several scalars is computed once, the role of the reference's ColumnVectorCache — input
// attributes mapped to program columns through `column_of`.
class ExpressionFlattener {
 public:
  explicit ExpressionFlattener(std::function<int(attribute_id)> column_of) : column_of_(std::move(column_of)) {}
  qsx_operand_t add(const ScalarPtr &scalar);   // the operand holding the scalar's value
  const std::vector<qsx_expr_instr_t> &instrs() const { return instrs_; }
  const std::vector<double> &consts() const { return consts_; }
 private:
  std::function<int(attribute_id)> column_of_;
  std::vector<qsx_expr_instr_t> instrs_;
  std::vector<double> consts_;
};

// ---------------------------------------------------------------------------
// aggregation state description (storage/AggregationOperationState.cpp:74-252)
// ---------------------------------------------------------------------------
struct AggregateSpec {
  AggregationID function;
  attribute_id argument;  // kInvalidAttributeID for COUNT(*)
  bool is_distinct = false;   // serialization::Aggregate::is_distinct (AggregationOperationState.proto)
  // serialization::Aggregate::argument as a Scalar tree (AggregationOperationState.cpp:100-130); when set it replaces
  // `argument`: SUM(l_extendedprice * (1 - l_discount))
  ScalarPtr argument_expression;
  AggregateSpec() : function(AggregationID::kCount), argument(kInvalidAttributeID) {}
  AggregateSpec(AggregationID f, attribute_id a, bool distinct = false) : function(f), argument(a), is_distinct(distinct) {}
  AggregateSpec(AggregationID f, ScalarPtr expression) : function(f), argument(kInvalidAttributeID), argument_expression(std::move(expression)) {}
};
struct AggregationStateSpec {
  const CatalogRelation *input_relation = nullptr;
  std::vector<attribute_id> group_by;
  std::vector<AggregateSpec> aggregates;
  const Predicate *predicate = nullptr;  // the state owns the predicate in the reference (:440-445)
  qsx_agg_strategy_t strategy = QSX_AGG_GENERIC;
  std::int64_t estimated_num_groups = 16;
  std::int64_t collision_free_num_entries = 0;
};

class AggregationOperationState {
 public:
  explicit AggregationOperationState(const AggregationStateSpec &spec);
  ~AggregationOperationState();
  // :428-474; lip_filter = TupleIdSequence left by the LIPFilterAdaptiveProber (:440-460), or nullptr
  void aggregateBlock(const StorageBlock &block, const std::uint64_t *lip_filter = nullptr);
  // :418-426 — slice `state_partition_id` of the collision-free vector table set to its initial values.  The device state is
  // one allocation zeroed by one fill: slice 0 clears all of it (qsx_agg_state_clear; a state fresh from the constructor
  // is clear already), the other slices have nothing left to do.  Any other strategy: ExecutionError, as the reference's
  // LOG(FATAL) "is not supported by this aggregation".
  void initialize(std::size_t state_partition_id);
  // A run of blocks in one launch where the state allows it (plain, non-nullable attributes, every conjunct inside the
  // kernel, no DISTINCT aggregate); blocks that need the per-block path take it.  lip_filters[i]: block i's filter or nullptr.
  void aggregateBlocks(const std::vector<BlockReference> &blocks, const std::vector<const std::uint64_t *> &lip_filters);
  void finalizeAggregate(std::size_t partition, std::size_t num_partitions, InsertDestination *dest);  // :641-694
  // getCollisionFreeVectorTable()->getExistenceMap()->setBit(key) for every tuple of the block
  // (BuildAggregationExistenceMapOperator.cpp:177-208); the state must use QSX_AGG_COLLISION_FREE
  void buildExistenceMap(const StorageBlock &block, attribute_id build_attribute, const Type &type);
  // Partial aggregates of every rank -> merged (ExchangeAggregationStatesOperator).  Hash-table states: every rank ends
  // with the whole merged table (qsx_agg_allgather_merge); CollisionFreeVector states: rank r ends with the merged groups
  // of key range r (qsx_agg_reduce_scatter) — finalize with (partition = rank, num_partitions = world) either way.
  // A collective: every rank calls it for the same states in the same order.  DISTINCT aggregates are not supported.
  void mergeAcrossRanks(qsx_comm_t *comm);
  const AggregationStateSpec &spec() const { return spec_; }
  // blocks aggregated on their code stripes (qsx_agg_update_coded) rather than on decoded values
  std::int64_t numBlocksAggregatedOnCodes() const { return coded_blocks_.load(); }

 private:
  AggregationStateSpec spec_;
  qsx_agg_config_t config_;
  qsx_agg_state_t *state_ = nullptr;       // the non-DISTINCT aggregates; nullptr when every aggregate is DISTINCT (all_distinct_)
  // Blocks of a compressed column store: a second state of the same shape whose operand columns are declared as code
  // stripes (qsx_agg_update_coded reads the codes and decodes on the fly); created for the coding of the first such
  // block, blocks coded differently take the value path; merged into state_ before the first finalize.
  qsx_agg_state_t *coded_state_ = nullptr;
  qsx_agg_config_t coded_config_;
  bool coded_predicate_external_ = false;   // the coded state takes the predicate as a filter (externalizeCodedPredicate)
  void externalizeCodedPredicate();
  bool coded_merged_ = false;
  std::mutex coded_mutex_;
  std::atomic<std::int64_t> coded_blocks_{0};
  std::vector<attribute_id> column_attr_;  // config column -> input attribute
  // conjuncts the state's kernel does not evaluate itself (CHAR(n) comparisons, terms beyond QSX_MAX_PRED_TERMS): they are
  // evaluated per block like a SelectOperator's predicate and handed to the update as its filter
  Predicate external_predicate_;
  std::vector<int> main_agg_;              // spec aggregate -> aggregate of config_ (-1: DISTINCT)
  // one per DISTINCT aggregate: distinctify_hashtables_ (AggregationOperationState.cpp:172-207), kept as the distinct
  // (group-by..., argument) tuples of the blocks seen so far
  struct Distinctify;
  std::vector<std::unique_ptr<Distinctify>> distinctify_;
  void finalizeWithDistinct(InsertDestination *dest);
};

// ---------------------------------------------------------------------------
// QueryContext (query_execution/QueryContext.hpp:190-451)
// ---------------------------------------------------------------------------
class QueryContext {
 public:
  typedef std::uint32_t predicate_id;
  typedef std::uint32_t scalar_group_id;
  typedef std::uint32_t join_hash_table_id;
  typedef std::uint32_t aggregation_state_id;
  typedef std::uint32_t insert_destination_id;
  typedef std::uint32_t lip_filter_id;
  typedef std::uint32_t lip_deployment_id;
  static constexpr lip_deployment_id kInvalidLIPDeploymentId = static_cast<lip_deployment_id>(-1);
  static constexpr predicate_id kInvalidPredicateId = static_cast<predicate_id>(-1);
  static constexpr insert_destination_id kInvalidInsertDestinationId = static_cast<insert_destination_id>(-1);

  ~QueryContext();
  // LIP filters and their deployments (QueryContext.hpp:338-395, utility/lip_filter/LIPFilterDeployment.hpp,
  // LIPFilter.proto:24-63): a deployment names, for one operator, the filters it builds or probes and
  // the attribute each one sees.
  struct LIPFilterDeploymentEntry { lip_filter_id lip_filter; attribute_id attribute; };
  struct LIPFilterDeployment {
    std::vector<LIPFilterDeploymentEntry> build_entries, probe_entries;
  };
  lip_filter_id addLIPFilter(qsx_lip_kind_t kind, std::int64_t cardinality, std::int64_t min_value = 0, bool is_anti = false);
  qsx_lip_filter_t *getLIPFilterMutable(lip_filter_id id) const { return lip_filters_.at(id); }
  void destroyLIPFilter(lip_filter_id id);
  lip_deployment_id addLIPDeployment(LIPFilterDeployment deployment);
  const LIPFilterDeployment *getLIPDeployment(lip_deployment_id id) const {
    return id == kInvalidLIPDeploymentId ? nullptr : &lip_deployments_.at(id);
  }
  predicate_id addPredicate(Predicate p);
  scalar_group_id addScalarGroup(std::vector<attribute_id> attrs);  // attribute projections only
  // exact_key_range: exact min/max statistics of the build-side join attribute when the optimizer has them
  // (the condition of query_optimizer/rules/InjectJoinFilters.cpp:130-150) -> the directly addressed table
  // flavour (qsx_join_table_create_dense); nullptr -> the hashed table.
  struct ExactKeyRange { std::int64_t min_value, max_value; };
  // utility/SortConfiguration.hpp:51-130: ORDER BY attributes, ordering[i] true = ascending (no NULL inputs here, so
  // the null_ordering vector of the reference has nothing to order)
  struct SortConfiguration {
    std::vector<attribute_id> order_by;
    std::vector<bool> ordering;
  };
  typedef std::uint32_t sort_config_id;
  sort_config_id addSortConfig(SortConfiguration config) {
    sort_configs_.push_back(std::move(config));
    return static_cast<sort_config_id>(sort_configs_.size() - 1);
  }
  const SortConfiguration &getSortConfig(sort_config_id id) const { return sort_configs_.at(id); }
  join_hash_table_id addJoinHashTable(TypeID key_type, std::int64_t estimated_entries, std::size_t num_partitions = 1,
                                      const ExactKeyRange *exact_key_range = nullptr);
  aggregation_state_id addAggregationState(const AggregationStateSpec &spec, std::size_t num_partitions = 1);
  insert_destination_id addInsertDestination(CatalogRelation *relation, StorageManager *storage_manager);
  // A PartitionAwareInsertDestination for an output relation with a hash partition scheme (relation->setPartitionScheme).
  insert_destination_id addPartitionAwareInsertDestination(CatalogRelation *relation, StorageManager *storage_manager);

  const Predicate *getPredicate(predicate_id id) const { return id == kInvalidPredicateId ? nullptr : &predicates_.at(id); }
  const std::vector<attribute_id> &getScalarGroup(scalar_group_id id) const { return scalar_groups_.at(id); }
  qsx_join_table_t *getJoinHashTable(join_hash_table_id id, partition_id part = 0) const { return join_tables_.at(id).at(part); }
  void destroyJoinHashTable(join_hash_table_id id, partition_id part = 0);
  // The build operator records which build attributes form the key (the reference keeps the key
  // values inside the table's buckets; here a hashed composite key is verified against the build
  // relation's columns, so the probe side has to know them).
  void setJoinHashTableBuildKeyAttributes(join_hash_table_id id, const std::vector<attribute_id> &attrs) {
    if (join_table_build_keys_.size() <= id) join_table_build_keys_.resize(id + 1);
    join_table_build_keys_[id] = attrs;
  }
  const std::vector<attribute_id> &getJoinHashTableBuildKeyAttributes(join_hash_table_id id) const {
    return join_table_build_keys_.at(id);
  }
  AggregationOperationState *getAggregationState(aggregation_state_id id, partition_id part = 0) const {
    return agg_states_.at(id).at(part).get();
  }
  void destroyAggregationState(aggregation_state_id id, partition_id part = 0) { agg_states_.at(id).at(part).reset(); }
  InsertDestination *getInsertDestination(insert_destination_id id) const { return destinations_.at(id).get(); }

 private:
  std::vector<Predicate> predicates_;
  std::vector<std::vector<attribute_id>> scalar_groups_;
  std::vector<SortConfiguration> sort_configs_;
  std::vector<qsx_lip_filter_t *> lip_filters_;
  std::vector<LIPFilterDeployment> lip_deployments_;
  std::vector<std::vector<qsx_join_table_t *>> join_tables_;
  std::vector<std::vector<attribute_id>> join_table_build_keys_;
  std::vector<std::vector<std::unique_ptr<AggregationOperationState>>> agg_states_;
  std::vector<std::unique_ptr<InsertDestination>> destinations_;
};

// LIPFilterBuilder (utility/lip_filter/LIPFilterBuilder.hpp): inserts the build attributes of a block
// into the deployment's filters; owned by the work order that got it (BuildHashOperator.hpp:274).
class LIPFilterBuilder {
 public:
  LIPFilterBuilder(const QueryContext::LIPFilterDeployment &deployment, const QueryContext &query_context);
  // tuples of `block` selected by `filter` (nullptr = all)  (BuildHashOperator.cpp:187-190)
  void insertValueAccessor(const StorageBlock &block, const std::uint64_t *filter) const;
  // A run of blocks, one launch per filter (qsx_lip_build_blocks); false when an inserted attribute is nullable or
  // compressed in one of the blocks (nothing inserted: the caller goes block by block).
  // built_table != nullptr: the join table the caller has just put these blocks' tuples into (same stream), keyed on
  // attribute table_key — an exact filter over that attribute takes its bits from the table where that is cheaper than an
  // atomic per key (qsx_lip_build_from_join_table).
  bool insertBlocks(const std::vector<BlockReference> &blocks, qsx_join_table_t *built_table = nullptr,
                    attribute_id table_key = kInvalidAttributeID) const;
  // What insertBlocks would answer, without inserting anything.
  bool coversBlocks(const std::vector<BlockReference> &blocks) const;
 private:
  std::vector<std::pair<qsx_lip_filter_t *, attribute_id>> entries_;
};
// LIPFilterAdaptiveProber (utility/lip_filter/LIPFilterAdaptiveProber.hpp:60-243): the tuples of a block
// that pass every probe filter of the deployment, as a TupleIdSequence.  The reference reorders the
// filters by observed selectivity between batches; on the device every filter is one pass over a
// column and the order does not change the result, so they run in deployment order.
class LIPFilterAdaptiveProber {
 public:
  LIPFilterAdaptiveProber(const QueryContext::LIPFilterDeployment &deployment, const QueryContext &query_context);
  // Returns a device bitmap owned by the caller (qsx_device_free): `filter` AND all probes.  (:83-90)
  void *filterValueAccessor(const StorageBlock &block, const std::uint64_t *filter, std::int64_t *num_hits) const;
  // The same for a run of blocks, one launch per filter (qsx_lip_probe_blocks): bitmaps[b] receives block b's TupleIdSequence.
  // All bitmaps live in *storage (a device allocation owned by the caller: qsx_device_free).  Returns false — nothing
  // allocated — when a probed attribute is nullable or compressed in one of the blocks (the caller goes block by block).
  // in_bitmaps: block b's TupleIdSequence so far (the predicate's matches: SelectOperator.cpp:161-195 filters what the
  // predicate left) — only those tuples are looked up.
  bool filterBlocks(const std::vector<BlockReference> &blocks, void **storage, std::vector<const std::uint64_t *> *bitmaps,
                    std::int64_t *num_hits = nullptr,   // num_hits: tuples of the run that pass (synchronises)
                    const std::uint64_t *const *in_bitmaps = nullptr) const;
 private:
  std::vector<std::pair<qsx_lip_filter_t *, attribute_id>> entries_;
};
// CreateLIPFilter{Builder,AdaptiveProber}Helper (utility/lip_filter/LIPFilterUtil.hpp): nullptr without a
// deployment or without entries of that kind.
LIPFilterBuilder *CreateLIPFilterBuilderHelper(QueryContext::lip_deployment_id id, const QueryContext *query_context);
LIPFilterAdaptiveProber *CreateLIPFilterAdaptiveProberHelper(QueryContext::lip_deployment_id id, const QueryContext *query_context);

// ---------------------------------------------------------------------------
// WorkOrder / container / RelationalOperator
// ---------------------------------------------------------------------------
class WorkOrder {
 public:
  virtual ~WorkOrder() {}
  virtual void execute() = 0;                                            // WorkOrder.hpp:251
  const std::vector<int> &getPreferredNUMANodes() const { return preferred_numa_nodes_; }  // :260
  // The counterpart of the NUMA preference on this device: a work order whose speed depends on a table staying in the XCDs'
  // L2 (a probe of a table of a few MiB over millions of rows) wants the device to itself — next to a streaming aggregation
  // the table is evicted continuously and the probe runs at half speed while the aggregation gains nothing.  With
  // QSX_HOST_EXCLUSIVE_PROBES=1 the Foreman lets such work orders run only among themselves (ForemanSingleNode::workerMain;
  // off by default: no gain measured on the headline plan, see there).
  virtual bool prefersExclusiveDevice() const { return false; }
  std::size_t getQueryID() const { return query_id_; }
  partition_id getPartitionId() const { return partition_id_; }

 protected:
  explicit WorkOrder(std::size_t query_id, partition_id part_id = 0) : query_id_(query_id), partition_id_(part_id) {}
  const std::size_t query_id_;
  const partition_id partition_id_;
  std::vector<int> preferred_numa_nodes_;
};

class WorkOrdersContainer {
 public:
  explicit WorkOrdersContainer(std::size_t num_operators) : queues_(num_operators) {}
  void addNormalWorkOrder(WorkOrder *workorder, std::size_t operator_index);  // WorkOrdersContainer.hpp:243
  bool hasNormalWorkOrder(std::size_t operator_index) const;
  WorkOrder *getNormalWorkOrder(std::size_t operator_index);                  // caller owns the result
  std::size_t getNumNormalWorkOrders(std::size_t operator_index) const;

 private:
  mutable std::mutex mutex_;
  std::vector<std::deque<std::unique_ptr<WorkOrder>>> queues_;
};

class RelationalOperator {
 public:
  enum OperatorType { kAggregation = 0, kBuildAggregationExistenceMap, kBuildHash, kDestroyAggregationState, kDestroyHash, kFinalizeAggregation,
                      kInitializeAggregation, kInnerJoin, kSelect, kSortMergeRun, kSortRunGeneration, kMockOperator,
                      kPartitionExchange, kExchangeAggregationStates };   // (the last two: multi-GPU, no counterpart in the reference)
  virtual ~RelationalOperator() {}
  virtual OperatorType getOperatorType() const = 0;
  virtual std::string getName() const = 0;
  // Generates all work orders available right now; returns true when no more
  // will ever be produced (RelationalOperator.hpp:111-136).  May be called repeatedly.
  virtual bool getAllWorkOrders(WorkOrdersContainer *container, QueryContext *query_context,
                                StorageManager *storage_manager, const tmb::client_id scheduler_client_id,
                                tmb::MessageBus *bus) = 0;
  virtual void feedInputBlock(const block_id input_block_id, const relation_id input_relation_id,
                              const partition_id part_id) {}               // :174
  virtual void doneFeedingInputBlocks(const relation_id rel_id) { done_feeding_input_relation_ = true; }  // :198
  virtual QueryContext::insert_destination_id getInsertDestinationID() const {
    return QueryContext::kInvalidInsertDestinationId;
  }
  virtual relation_id getOutputRelationID() const { return -1; }
  // RelationalOperator.hpp:294-297
  void deployLIPFilters(const QueryContext::lip_deployment_id lip_deployment_index) { lip_deployment_index_ = lip_deployment_index; }
  void setOperatorIndex(std::size_t index) { op_index_ = index; }
  std::size_t getOperatorIndex() const { return op_index_; }
  std::size_t getQueryID() const { return query_id_; }
  std::size_t getNumPartitions() const { return num_partitions_; }
  bool hasRepartition() const { return has_repartition_; }                       // :278
  std::size_t getOutputNumPartitions() const { return output_num_partitions_; }  // :287
  // :164 — called by the query manager when the operator has finished; none of the operators here touches the catalog
  // at that point (the reference's overriders are the DDL / load operators)
  virtual void updateCatalogOnCompletion() {}
  // Multi-GPU: the operator's work order issues collectives (every rank must run it, and all ranks must run such operators
  // in the same order).  The Foreman lets a collective operator start only when every collective operator with a smaller
  // plan index has finished — all ranks build the same plan, so all issue the same sequence of collectives.
  virtual bool isCollective() const { return false; }

 protected:
  explicit RelationalOperator(std::size_t query_id, std::size_t num_partitions = 1, bool has_repartition = false,
                              std::size_t output_num_partitions = 1)
      : query_id_(query_id), num_partitions_(num_partitions), has_repartition_(has_repartition),
        output_num_partitions_(output_num_partitions) {}
  const std::size_t query_id_;
  const std::size_t num_partitions_;
  const bool has_repartition_;
  const std::size_t output_num_partitions_;
  bool done_feeding_input_relation_ = false;
  std::size_t op_index_ = 0;
  QueryContext::lip_deployment_id lip_deployment_index_ = QueryContext::kInvalidLIPDeploymentId;
};

// ---------------------------------------------------------------------------
// SelectOperator (SelectOperator.hpp:90-98,149-157; simple projection form)
// ---------------------------------------------------------------------------
class SelectOperator : public RelationalOperator {
 public:
  // on_gpu = false builds the CPU work order of BASELINE config 1 (plumbing only, no HIP).
  SelectOperator(std::size_t query_id, const CatalogRelation &input_relation, bool has_repartition,
                 const CatalogRelation &output_relation, QueryContext::insert_destination_id output_destination_index,
                 QueryContext::predicate_id predicate_index, std::vector<attribute_id> &&selection,
                 bool input_relation_is_stored, bool on_gpu = true);
  // The general form (SelectOperator.hpp:90-98: selection = a group of Scalars): attributes and arithmetic expressions;
  // an expression's output attribute has ScalarResultType(...): INT / LONG for integer operands, else DOUBLE.
  SelectOperator(std::size_t query_id, const CatalogRelation &input_relation, bool has_repartition,
                 const CatalogRelation &output_relation, QueryContext::insert_destination_id output_destination_index,
                 QueryContext::predicate_id predicate_index, std::vector<ScalarPtr> &&selection, bool input_relation_is_stored);
  OperatorType getOperatorType() const override { return kSelect; }
  std::string getName() const override { return "SelectOperator"; }
  bool getAllWorkOrders(WorkOrdersContainer *container, QueryContext *query_context, StorageManager *storage_manager,
                        const tmb::client_id scheduler_client_id, tmb::MessageBus *bus) override;
  void feedInputBlock(const block_id input_block_id, const relation_id, const partition_id) override {
    std::lock_guard<std::mutex> lock(mutex_);
    input_relation_block_ids_.push_back(input_block_id);
  }
  QueryContext::insert_destination_id getInsertDestinationID() const override { return output_destination_index_; }
  relation_id getOutputRelationID() const override { return output_relation_.getID(); }
  // How many input blocks one SelectWorkOrder covers (1 = the reference's one work order per block,
  // SelectOperator.cpp:83-150).  A run of blocks is one launch per predicate term plus one compaction into ONE output
  // block (qsx_select_cmp_blocks / qsx_compact_gather_blocks); blocks the run form does not cover (compressed, sorted,
  // CHAR or nullable attributes, expressions, LIP filters) are executed one by one inside the work order.
  void setBlocksPerWorkOrder(std::size_t blocks) { blocks_per_work_order_ = blocks > 0 ? blocks : 1; }

 private:
  std::size_t blocks_per_work_order_ = 1;
  const CatalogRelation &input_relation_;
  const CatalogRelation &output_relation_;
  const QueryContext::insert_destination_id output_destination_index_;
  const QueryContext::predicate_id predicate_index_;
  const std::vector<attribute_id> simple_selection_;
  const std::vector<ScalarPtr> selection_;   // non-empty: the general form
  const bool input_relation_is_stored_;
  const bool on_gpu_;
  std::mutex mutex_;
  std::vector<block_id> input_relation_block_ids_;
  std::size_t num_workorders_generated_ = 0;
  bool started_ = false;
};

class SelectWorkOrder : public WorkOrder {
 public:
  SelectWorkOrder(std::size_t query_id, block_id input_block_id, const Predicate *predicate,
                  const std::vector<attribute_id> &simple_selection, InsertDestination *output_destination,
                  StorageManager *storage_manager, bool on_gpu, LIPFilterAdaptiveProber *lip_filter_adaptive_prober = nullptr,
                  const std::vector<ScalarPtr> *selection = nullptr)
      : WorkOrder(query_id), input_block_id_(input_block_id), predicate_(predicate), simple_selection_(simple_selection),
        selection_(selection), output_destination_(output_destination), storage_manager_(storage_manager), on_gpu_(on_gpu),
        lip_filter_adaptive_prober_(lip_filter_adaptive_prober) {}
  // A run of input blocks (SelectOperator::setBlocksPerWorkOrder).
  SelectWorkOrder(std::size_t query_id, std::vector<block_id> &&input_block_ids, const Predicate *predicate,
                  const std::vector<attribute_id> &simple_selection, InsertDestination *output_destination,
                  StorageManager *storage_manager, LIPFilterAdaptiveProber *lip_filter_adaptive_prober = nullptr,
                  const std::vector<ScalarPtr> *selection = nullptr)
      : WorkOrder(query_id), input_block_id_(input_block_ids.front()), run_block_ids_(std::move(input_block_ids)),
        predicate_(predicate), simple_selection_(simple_selection), selection_(selection),
        output_destination_(output_destination), storage_manager_(storage_manager), on_gpu_(true),
        lip_filter_adaptive_prober_(lip_filter_adaptive_prober) {}
  void execute() override;  // SelectOperator.cpp:161-195

 private:
  void executeOnHost();
  void executeBlock(block_id input_block_id);
  bool executeRun();          // false: the run form does not cover these blocks
  const block_id input_block_id_;
  const std::vector<block_id> run_block_ids_;   // empty: the single block input_block_id_
  const Predicate *predicate_;
  const std::vector<attribute_id> &simple_selection_;
  const std::vector<ScalarPtr> *selection_;   // nullptr / empty: simple_selection_
  InsertDestination *output_destination_;
  StorageManager *storage_manager_;
  const bool on_gpu_;
  std::unique_ptr<LIPFilterAdaptiveProber> lip_filter_adaptive_prober_;  // SelectOperator.hpp:383
};

// ---------------------------------------------------------------------------
// BuildHashOperator (BuildHashOperator.hpp:86-93)
// ---------------------------------------------------------------------------
class BuildHashOperator : public RelationalOperator {
 public:
  BuildHashOperator(std::size_t query_id, const CatalogRelation &input_relation, bool input_relation_is_stored,
                    const std::vector<attribute_id> &join_key_attributes, bool any_join_key_attributes_nullable,
                    std::size_t num_partitions, QueryContext::join_hash_table_id hash_table_index,
                    QueryContext::predicate_id build_predicate_index = QueryContext::kInvalidPredicateId);
  OperatorType getOperatorType() const override { return kBuildHash; }
  std::string getName() const override { return "BuildHashOperator"; }
  bool getAllWorkOrders(WorkOrdersContainer *container, QueryContext *query_context, StorageManager *storage_manager,
                        const tmb::client_id scheduler_client_id, tmb::MessageBus *bus) override;
  void feedInputBlock(const block_id input_block_id, const relation_id, const partition_id part_id) override {
    std::lock_guard<std::mutex> lock(mutex_);
    if (is_broadcast_join_) {  // BuildHashOperator.hpp:146-152
      for (auto &ids : input_.ids) ids.push_back(input_block_id);
    } else {
      input_.ids.at(part_id).push_back(input_block_id);
    }
  }
  // How many build blocks one BuildHashWorkOrder covers (1 = the reference's one per block, BuildHashOperator.cpp:70-130).
  // A run is inserted with one launch (qsx_join_build_blocks) when the key is one non-nullable attribute and the
  // operator has neither a build predicate nor a LIP filter to fill; otherwise block by block inside the work order.
  void setBlocksPerWorkOrder(std::size_t blocks) { blocks_per_work_order_ = blocks > 0 ? blocks : 1; }

 private:
  std::size_t blocks_per_work_order_ = 1;
  const CatalogRelation &input_relation_;
  const bool input_relation_is_stored_;
  const std::vector<attribute_id> join_key_attributes_;
  const bool is_broadcast_join_;  // build side unpartitioned, probe side partitioned (BuildHashOperator.hpp:99)
  const QueryContext::join_hash_table_id hash_table_index_;
  const QueryContext::predicate_id build_predicate_index_;
  std::mutex mutex_;
  PartitionedBlockIds input_;
  bool started_ = false;
};

class BuildHashWorkOrder : public WorkOrder {
 public:
  BuildHashWorkOrder(std::size_t query_id, const CatalogRelation &input_relation,
                     const std::vector<attribute_id> &join_key_attributes, block_id build_block_id,
                     const Predicate *predicate, qsx_join_table_t *hash_table, StorageManager *storage_manager,
                     partition_id part_id = 0, LIPFilterBuilder *lip_filter_builder = nullptr)
      : WorkOrder(query_id, part_id), input_relation_(input_relation), join_key_attributes_(join_key_attributes),
        build_block_id_(build_block_id), predicate_(predicate), hash_table_(hash_table),
        storage_manager_(storage_manager), lip_filter_builder_(lip_filter_builder) {}
  void execute() override;  // BuildHashOperator.cpp:162-207
  // A run of build blocks (BuildHashOperator::setBlocksPerWorkOrder): build_block_id is the first of them.
  void setRun(std::vector<block_id> &&build_block_ids) { run_block_ids_ = std::move(build_block_ids); }

 private:
  void executeBlock(block_id build_block_id);
  bool executeRun();
  std::vector<block_id> run_block_ids_;
  const CatalogRelation &input_relation_;
  const std::vector<attribute_id> &join_key_attributes_;
  const block_id build_block_id_;
  const Predicate *predicate_;
  qsx_join_table_t *hash_table_;
  StorageManager *storage_manager_;
  std::unique_ptr<LIPFilterBuilder> lip_filter_builder_;  // BuildHashOperator.hpp:274
};

// ---------------------------------------------------------------------------
// HashJoinOperator, inner join (HashJoinOperator.hpp:126-141)
// ---------------------------------------------------------------------------
class HashJoinOperator : public RelationalOperator {
 public:
  enum class JoinType { kInnerJoin = 0, kLeftSemiJoin, kLeftAntiJoin, kLeftOuterJoin };
  // selection + is_selection_on_build: output attribute i is attribute selection[i] of the build
  // relation when is_selection_on_build[i], else of the probe relation.
  HashJoinOperator(std::size_t query_id, const CatalogRelation &build_relation, const CatalogRelation &probe_relation,
                   bool probe_relation_is_stored, const std::vector<attribute_id> &join_key_attributes,
                   bool any_join_key_attributes_nullable, std::size_t num_partitions, bool has_repartition,
                   const CatalogRelation &output_relation, QueryContext::insert_destination_id output_destination_index,
                   QueryContext::join_hash_table_id hash_table_index, QueryContext::predicate_id residual_predicate_index,
                   QueryContext::scalar_group_id selection_index, const std::vector<bool> *is_selection_on_build = nullptr,
                   JoinType join_type = JoinType::kInnerJoin);
  OperatorType getOperatorType() const override { return kInnerJoin; }
  std::string getName() const override { return "HashJoinOperator"; }
  bool getAllWorkOrders(WorkOrdersContainer *container, QueryContext *query_context, StorageManager *storage_manager,
                        const tmb::client_id scheduler_client_id, tmb::MessageBus *bus) override;
  void feedInputBlock(const block_id input_block_id, const relation_id, const partition_id part_id) override {
    std::lock_guard<std::mutex> lock(mutex_);
    probe_.ids.at(part_id).push_back(input_block_id);
  }
  QueryContext::insert_destination_id getInsertDestinationID() const override { return output_destination_index_; }
  relation_id getOutputRelationID() const override { return output_relation_.getID(); }
  // How many probe blocks one work order covers (1 = the reference's one probe work order per block,
  // HashJoinOperator.cpp:203-260).  A run is probed with one launch (qsx_join_probe_blocks) and its joined tuples go to
  // ONE output block; an inner join on one non-nullable key attribute without residual predicate or LIP filter takes the
  // run form, everything else is executed block by block inside the work order.
  void setBlocksPerWorkOrder(std::size_t blocks) { blocks_per_work_order_ = blocks > 0 ? blocks : 1; }

 private:
  std::size_t blocks_per_work_order_ = 1;
  const CatalogRelation &build_relation_;
  const CatalogRelation &probe_relation_;
  const bool probe_relation_is_stored_;
  const std::vector<attribute_id> join_key_attributes_;
  const CatalogRelation &output_relation_;
  const QueryContext::insert_destination_id output_destination_index_;
  const QueryContext::join_hash_table_id hash_table_index_;
  const QueryContext::predicate_id residual_predicate_index_;
  const QueryContext::scalar_group_id selection_index_;
  std::vector<bool> is_selection_on_build_;
  std::vector<attribute_id> build_key_attributes_;
  const JoinType join_type_;
  std::mutex mutex_;
  PartitionedBlockIds probe_;
  bool started_ = false;
};

// One work order class for the four join types (the reference has HashInnerJoinWorkOrder,
// HashSemiJoinWorkOrder, HashAntiJoinWorkOrder, HashOuterJoinWorkOrder; HashJoinOperator.cpp:450-1099).
class HashInnerJoinWorkOrder : public WorkOrder {
 public:
  HashInnerJoinWorkOrder(std::size_t query_id, const CatalogRelation &build_relation,
                         const CatalogRelation &probe_relation, const std::vector<attribute_id> &join_key_attributes,
                         const std::vector<attribute_id> &build_key_attributes,
                         block_id lookup_block_id, const Predicate *residual_predicate,
                         const std::vector<attribute_id> &selection, const std::vector<bool> &is_selection_on_build,
                         HashJoinOperator::JoinType join_type, qsx_join_table_t *hash_table,
                         InsertDestination *output_destination, StorageManager *storage_manager, partition_id part_id = 0,
                         LIPFilterAdaptiveProber *lip_filter_adaptive_prober = nullptr)
      : WorkOrder(query_id, part_id), lip_filter_adaptive_prober_(lip_filter_adaptive_prober),
        build_relation_(build_relation), probe_relation_(probe_relation),
        join_key_attributes_(join_key_attributes), build_key_attributes_(build_key_attributes),
        block_id_(lookup_block_id), residual_predicate_(residual_predicate),
        selection_(selection), is_selection_on_build_(is_selection_on_build), join_type_(join_type),
        hash_table_(hash_table), output_destination_(output_destination), storage_manager_(storage_manager) {}
  void execute() override;  // HashJoinOperator.cpp:450-541 (inner), :680-877 (semi / anti), :960-1099 (outer)
  // A run of probe blocks (HashJoinOperator::setBlocksPerWorkOrder): lookup_block_id is the first of them.
  void setRun(std::vector<block_id> &&probe_block_ids) { run_block_ids_ = std::move(probe_block_ids); }
  // a probe of millions of rows: its table wants to stay in L2 (called on the Foreman thread when the work order is queued)
  bool prefersExclusiveDevice() const override;

 private:
  void executeBlock(block_id probe_block_id);
  bool executeRun();          // false: the run form does not cover this join
  std::vector<block_id> run_block_ids_;
  std::unique_ptr<LIPFilterAdaptiveProber> lip_filter_adaptive_prober_;
  const CatalogRelation &build_relation_;
  const CatalogRelation &probe_relation_;
  const std::vector<attribute_id> &join_key_attributes_;
  const std::vector<attribute_id> &build_key_attributes_;
  const block_id block_id_;
  const Predicate *residual_predicate_;
  const std::vector<attribute_id> &selection_;
  const std::vector<bool> &is_selection_on_build_;
  const HashJoinOperator::JoinType join_type_;
  qsx_join_table_t *hash_table_;
  InsertDestination *output_destination_;
  StorageManager *storage_manager_;
};

class DestroyHashOperator : public RelationalOperator {
 public:
  DestroyHashOperator(std::size_t query_id, std::size_t num_partitions, QueryContext::join_hash_table_id hash_table_index)
      : RelationalOperator(query_id, num_partitions), hash_table_index_(hash_table_index) {}
  OperatorType getOperatorType() const override { return kDestroyHash; }
  std::string getName() const override { return "DestroyHashOperator"; }
  bool getAllWorkOrders(WorkOrdersContainer *container, QueryContext *query_context, StorageManager *storage_manager,
                        const tmb::client_id scheduler_client_id, tmb::MessageBus *bus) override;

 private:
  const QueryContext::join_hash_table_id hash_table_index_;
  bool work_generated_ = false;
};

// ---------------------------------------------------------------------------
// Aggregation (AggregationOperator.hpp:55-59, FinalizeAggregationOperator.hpp:54-61)
// ---------------------------------------------------------------------------
class AggregationOperator : public RelationalOperator {
 public:
  AggregationOperator(std::size_t query_id, const CatalogRelation &input_relation, bool input_relation_is_stored,
                      QueryContext::aggregation_state_id aggr_state_index, std::size_t num_partitions = 1);
  OperatorType getOperatorType() const override { return kAggregation; }
  std::string getName() const override { return "AggregationOperator"; }
  bool getAllWorkOrders(WorkOrdersContainer *container, QueryContext *query_context, StorageManager *storage_manager,
                        const tmb::client_id scheduler_client_id, tmb::MessageBus *bus) override;
  void feedInputBlock(const block_id input_block_id, const relation_id, const partition_id part_id) override {
    std::lock_guard<std::mutex> lock(mutex_);
    input_.ids.at(part_id).push_back(input_block_id);
  }
  // Work-order granularity (the operator's to decide, RelationalOperator.hpp:117-119): up to this many input blocks per
  // AggregationWorkOrder.  The reference's blocks are 2-4 MB — about 120 K Q1 rows, half a microsecond of HBM time behind
  // ~16 us of launch — so a GPU work order takes a run of them and aggregates it in one launch (qsx_agg_update_blocks).
  // Default 1: one work order per block, like the reference.
  void setBlocksPerWorkOrder(std::size_t blocks) { blocks_per_work_order_ = blocks > 0 ? blocks : 1; }

 private:
  const CatalogRelation &input_relation_;
  const bool input_relation_is_stored_;
  const QueryContext::aggregation_state_id aggr_state_index_;
  std::mutex mutex_;
  PartitionedBlockIds input_;
  bool started_ = false;
  std::size_t blocks_per_work_order_ = 1;
};

// relational_operators/BuildAggregationExistenceMapOperator.hpp:57-140: marks the keys of the left relation of a
// CrossReferenceCoalesceAggregate in the collision-free table of the aggregation state, so that keys without
// right-side rows still finalize (COUNT 0).  Runs after InitializeAggregation, before the AggregationOperator.
class BuildAggregationExistenceMapOperator : public RelationalOperator {
 public:
  BuildAggregationExistenceMapOperator(std::size_t query_id, const CatalogRelation &input_relation, attribute_id build_attribute,
                                       bool input_relation_is_stored, QueryContext::aggregation_state_id aggr_state_index,
                                       std::size_t num_partitions = 1);
  OperatorType getOperatorType() const override { return kBuildAggregationExistenceMap; }
  std::string getName() const override { return "BuildAggregationExistenceMapOperator"; }
  bool getAllWorkOrders(WorkOrdersContainer *container, QueryContext *query_context, StorageManager *storage_manager,
                        const tmb::client_id scheduler_client_id, tmb::MessageBus *bus) override;
  void feedInputBlock(const block_id input_block_id, const relation_id, const partition_id part_id) override {
    std::lock_guard<std::mutex> lock(mutex_);
    input_.ids.at(part_id).push_back(input_block_id);
  }

 private:
  const CatalogRelation &input_relation_;
  const attribute_id build_attribute_;
  const bool input_relation_is_stored_;
  const QueryContext::aggregation_state_id aggr_state_index_;
  std::mutex mutex_;
  PartitionedBlockIds input_;
};

class FinalizeAggregationOperator : public RelationalOperator {
 public:
  FinalizeAggregationOperator(std::size_t query_id, QueryContext::aggregation_state_id aggr_state_index,
                              std::size_t num_partitions, bool has_repartition, std::size_t aggr_state_num_partitions,
                              const CatalogRelation &output_relation,
                              QueryContext::insert_destination_id output_destination_index)
      : RelationalOperator(query_id, num_partitions, has_repartition), aggr_state_index_(aggr_state_index),
        aggr_state_num_partitions_(aggr_state_num_partitions), output_relation_(output_relation),
        output_destination_index_(output_destination_index) {}
  OperatorType getOperatorType() const override { return kFinalizeAggregation; }
  std::string getName() const override { return "FinalizeAggregationOperator"; }
  bool getAllWorkOrders(WorkOrdersContainer *container, QueryContext *query_context, StorageManager *storage_manager,
                        const tmb::client_id scheduler_client_id, tmb::MessageBus *bus) override;
  QueryContext::insert_destination_id getInsertDestinationID() const override { return output_destination_index_; }
  relation_id getOutputRelationID() const override { return output_relation_.getID(); }
  // One process per GPU: after ExchangeAggregationStatesOperator every rank holds the merged state (hash tables) or the
  // merged groups of its key range (CollisionFreeVector); construct the operator with aggr_state_num_partitions = world and
  // let rank r emit finalize partition r only — every group leaves the job exactly once.
  void setRankSlice(std::size_t rank) { rank_slice_ = static_cast<std::int64_t>(rank); }

 private:
  const QueryContext::aggregation_state_id aggr_state_index_;
  const std::size_t aggr_state_num_partitions_;
  const CatalogRelation &output_relation_;
  const QueryContext::insert_destination_id output_destination_index_;
  std::int64_t rank_slice_ = -1;   // >= 0: only this finalize partition of every state
  bool started_ = false;
};

// InitializeAggregationOperator.hpp:52-99: num_partitions x aggr_state_num_init_partitions work orders, each
// AggregationOperationState::initialize(state_partition_id) (.cpp:38-62, 91-93).  The optimizer adds it in front of an
// aggregation over a CollisionFreeVectorTable (ExecutionGenerator.cpp:204-208 sizes the slices).
class InitializeAggregationOperator : public RelationalOperator {
 public:
  InitializeAggregationOperator(std::size_t query_id, QueryContext::aggregation_state_id aggr_state_index,
                                std::size_t num_partitions, std::size_t aggr_state_num_init_partitions)
      : RelationalOperator(query_id, num_partitions), aggr_state_index_(aggr_state_index),
        aggr_state_num_init_partitions_(aggr_state_num_init_partitions) {}
  OperatorType getOperatorType() const override { return kInitializeAggregation; }
  std::string getName() const override { return "InitializeAggregationOperator"; }
  bool getAllWorkOrders(WorkOrdersContainer *container, QueryContext *query_context, StorageManager *storage_manager,
                        const tmb::client_id scheduler_client_id, tmb::MessageBus *bus) override;

 private:
  const QueryContext::aggregation_state_id aggr_state_index_;
  const std::size_t aggr_state_num_init_partitions_;
  bool started_ = false;
};

class DestroyAggregationStateOperator : public RelationalOperator {
 public:
  DestroyAggregationStateOperator(std::size_t query_id, QueryContext::aggregation_state_id aggr_state_index,
                                  std::size_t num_partitions = 1)
      : RelationalOperator(query_id, num_partitions), aggr_state_index_(aggr_state_index) {}
  OperatorType getOperatorType() const override { return kDestroyAggregationState; }
  std::string getName() const override { return "DestroyAggregationStateOperator"; }
  bool getAllWorkOrders(WorkOrdersContainer *container, QueryContext *query_context, StorageManager *storage_manager,
                        const tmb::client_id scheduler_client_id, tmb::MessageBus *bus) override;

 private:
  const QueryContext::aggregation_state_id aggr_state_index_;
  bool work_generated_ = false;
};

// ---------------------------------------------------------------------------
// ORDER BY: SortRunGenerationOperator (SortRunGenerationOperator.hpp:86-108) sorts every input block into a run,
// SortMergeRunOperator (SortMergeRunOperator.hpp:99-133) merges the runs into the output relation, optionally only
// the first top_k tuples.  On the device a merge of sorted runs is the same radix sort over their concatenation
// (qsx_sort_permutation), so the merge operator issues ONE work order once all runs have arrived; merge_factor and
// the intermediate run relation of the reference's multi-pass merge tree are accepted and unused.
// ---------------------------------------------------------------------------
class SortRunGenerationOperator : public RelationalOperator {
 public:
  SortRunGenerationOperator(std::size_t query_id, const CatalogRelation &input_relation, const CatalogRelation &output_relation,
                            QueryContext::insert_destination_id output_destination_index,
                            QueryContext::sort_config_id sort_config_index, bool input_relation_is_stored)
      : RelationalOperator(query_id), input_relation_(input_relation), output_relation_(output_relation),
        output_destination_index_(output_destination_index), sort_config_index_(sort_config_index),
        input_relation_is_stored_(input_relation_is_stored) {
    if (input_relation_is_stored) input_relation_block_ids_ = input_relation.getBlocksSnapshot();
  }
  OperatorType getOperatorType() const override { return kSortRunGeneration; }
  std::string getName() const override { return "SortRunGenerationOperator"; }
  bool getAllWorkOrders(WorkOrdersContainer *container, QueryContext *query_context, StorageManager *storage_manager,
                        const tmb::client_id scheduler_client_id, tmb::MessageBus *bus) override;
  void feedInputBlock(const block_id input_block_id, const relation_id, const partition_id) override {
    std::lock_guard<std::mutex> lock(mutex_);
    input_relation_block_ids_.push_back(input_block_id);
  }
  QueryContext::insert_destination_id getInsertDestinationID() const override { return output_destination_index_; }
  relation_id getOutputRelationID() const override { return output_relation_.getID(); }
  // ORDER BY ... LIMIT k: the merge behind this operator keeps k tuples (SortMergeRunOperator's top_k), so no run needs more than
  // its first k — a run is then a selection of k tuples (qsx_sort_top_k) instead of a sort of the whole block.  0 = whole runs.
  void setTopK(std::size_t top_k) { top_k_ = top_k; }

 private:
  const CatalogRelation &input_relation_;
  const CatalogRelation &output_relation_;
  const QueryContext::insert_destination_id output_destination_index_;
  const QueryContext::sort_config_id sort_config_index_;
  const bool input_relation_is_stored_;
  std::size_t top_k_ = 0;
  std::mutex mutex_;
  std::vector<block_id> input_relation_block_ids_;
  std::size_t num_workorders_generated_ = 0;
};

class SortMergeRunOperator : public RelationalOperator {
 public:
  SortMergeRunOperator(std::size_t query_id, const CatalogRelation &input_relation, const CatalogRelation &output_relation,
                       QueryContext::insert_destination_id output_destination_index, const CatalogRelation &run_relation,
                       QueryContext::insert_destination_id run_block_destination_index,
                       QueryContext::sort_config_id sort_config_index, std::size_t merge_factor, std::size_t top_k,
                       bool input_relation_is_stored)
      : RelationalOperator(query_id), input_relation_(input_relation), output_relation_(output_relation),
        output_destination_index_(output_destination_index), sort_config_index_(sort_config_index), top_k_(top_k),
        input_relation_is_stored_(input_relation_is_stored) {
    (void)run_relation; (void)run_block_destination_index; (void)merge_factor;
    if (input_relation_is_stored) input_relation_block_ids_ = input_relation.getBlocksSnapshot();
  }
  OperatorType getOperatorType() const override { return kSortMergeRun; }
  std::string getName() const override { return "SortMergeRunOperator"; }
  bool getAllWorkOrders(WorkOrdersContainer *container, QueryContext *query_context, StorageManager *storage_manager,
                        const tmb::client_id scheduler_client_id, tmb::MessageBus *bus) override;
  void feedInputBlock(const block_id input_block_id, const relation_id, const partition_id) override {
    std::lock_guard<std::mutex> lock(mutex_);
    input_relation_block_ids_.push_back(input_block_id);
  }
  QueryContext::insert_destination_id getInsertDestinationID() const override { return output_destination_index_; }
  relation_id getOutputRelationID() const override { return output_relation_.getID(); }

 private:
  const CatalogRelation &input_relation_;
  const CatalogRelation &output_relation_;
  const QueryContext::insert_destination_id output_destination_index_;
  const QueryContext::sort_config_id sort_config_index_;
  const std::size_t top_k_;    // 0 = all tuples
  const bool input_relation_is_stored_;
  std::mutex mutex_;
  std::vector<block_id> input_relation_block_ids_;
  bool work_generated_ = false;
};

// ---------------------------------------------------------------------------
// Multi-GPU: one process per GPU, every process runs the SAME plan under its own ForemanSingleNode over its own
// StorageManager and QueryContext; GPU (rank) r owns the partitions p with p % world == r of every hash-partitioned
// relation.  The reference's partitions share one address space — PartitionAwareInsertDestination routes a tuple to the
// block of its partition (storage/InsertDestination.hpp:490-660) and the per-partition work orders of BuildHash / HashJoin /
// Aggregation read them where they lie (BuildHashOperator.cpp:82-91, HashJoinOperator.cpp:220-231,
// AggregationOperator.cpp:49-61); its distributed mode moves whole blocks between the StorageManagers of its nodes on
// demand (storage/DataExchangerAsync.cpp, DataExchange.proto:22-34: Pull by block id).  Here a partition that is not this
// rank's leaves as one exchange step per plan edge: PartitionExchangeOperator behind the repartitioning producer
// (qsx_exchange_counts + qsx_alltoallv per attribute = RCCL all-to-all over xGMI), and ExchangeAggregationStatesOperator in
// front of FinalizeAggregation (qsx_agg_reduce_scatter / qsx_agg_allgather_merge).  Collectives are issued by exactly one
// work order per such operator, and the Foreman runs those operators one after the other in plan order
// (RelationalOperator::isCollective), so all ranks issue the same sequence of collectives.
// ---------------------------------------------------------------------------
class RankGroup {
 public:
  // Rank 0 makes the id; the caller's control plane carries the QSX_COMM_ID_BYTES to the other ranks (the reference's
  // control plane is the TMB; tests use a file).
  static std::vector<unsigned char> MakeUniqueId();
  RankGroup(int world, int rank, const void *id_bytes);   // qsx_comm_create on the calling thread's device
  ~RankGroup();
  RankGroup(const RankGroup &) = delete;
  RankGroup &operator=(const RankGroup &) = delete;
  int world() const { return world_; }
  int rank() const { return rank_; }
  qsx_comm_t *comm() const { return comm_; }
  std::size_t ownerOf(partition_id part) const { return part % static_cast<std::size_t>(world_); }
  bool owns(partition_id part) const { return ownerOf(part) == static_cast<std::size_t>(rank_); }
  // Failure agreement (qsx_comm_agree).  `prepare` is this rank's own, collective-free part of a step — the work that can
  // fail on one rank alone (fetching blocks, allocating the output block and the send buffers, validating what arrived).
  // Either it succeeded on EVERY rank and agreeOn returns on every rank, or it throws on every rank (the rank whose
  // preparation threw rethrows that exception, the others an ExecutionError with QSX_ERR_COMM): no rank walks into a
  // collective its peers will never enter.  A collective itself: all ranks call it at the same point of the step.
  void agreeOn(const std::function<void()> &prepare, const char *where);
  // Wait for the calling thread's stream under the communicator's watchdog (qsx_comm_synchronize, QSX_COMM_TIMEOUT_MS).
  void synchronize();
  // A rank that failed INSIDE a step's collectives gives the communicator up: its peers' collectives end with an error.
  void abort() noexcept;

 private:
  int world_, rank_;
  qsx_comm_t *comm_ = nullptr;
};

// The blocks of `input_relation` on this rank -> the ranks that own their partitions.
//   partitioned input (a relation with a partition scheme, filled by a PartitionAwareInsertDestination or stored): the
//     tuples of partition p go to rank p % world; the output relation has the same attributes and partition scheme, its
//     blocks — one per owned partition and round, tuples of rank 0 first, every rank's in their local order — are
//     registered under their partition and streamed to the consumers with it (kDataPipelineMessage's partition id);
//   unpartitioned input + broadcast = true: every rank receives the tuples of all ranks (rank order) as ONE block of the
//     (unpartitioned) output relation — the build side of a broadcast join (BuildHashOperator.hpp:99, 146-152).
// One work order at a time, each over the blocks that have arrived since the last (a ROUND of collectives: the producer keeps
// filling blocks under it, and the whole shuffled relation never has to be held at once); every rank runs the same number of
// rounds — the counts exchange carries "more of mine will follow", and the round in which no rank says so is the last everywhere.
// NULL bitmaps of nullable attributes travel as word-aligned bitmaps per (source rank, partition) and are re-packed on arrival.
class PartitionExchangeOperator : public RelationalOperator {
 public:
  PartitionExchangeOperator(std::size_t query_id, const CatalogRelation &input_relation, bool input_relation_is_stored,
                            const CatalogRelation &output_relation, QueryContext::insert_destination_id output_destination_index,
                            RankGroup *ranks, bool broadcast = false);
  OperatorType getOperatorType() const override { return kPartitionExchange; }
  bool isCollective() const override { return true; }
  std::string getName() const override { return "PartitionExchangeOperator"; }
  bool getAllWorkOrders(WorkOrdersContainer *container, QueryContext *query_context, StorageManager *storage_manager,
                        const tmb::client_id scheduler_client_id, tmb::MessageBus *bus) override;
  void feedInputBlock(const block_id input_block_id, const relation_id, const partition_id part_id) override {
    std::lock_guard<std::mutex> lock(mutex_);
    input_.ids.at(broadcast_ ? 0 : part_id).push_back(input_block_id);
  }
  QueryContext::insert_destination_id getInsertDestinationID() const override { return output_destination_index_; }
  relation_id getOutputRelationID() const override { return output_relation_.getID(); }
  // bytes this rank sent to other ranks / received from them (after the work order has run)
  std::uint64_t bytesSentToPeers() const { return bytes_sent_.load(); }
  std::size_t numRounds() const { return rounds_; }

 private:
  friend class PartitionExchangeWorkOrder;
  const CatalogRelation &input_relation_;
  const bool input_relation_is_stored_;
  const CatalogRelation &output_relation_;
  const QueryContext::insert_destination_id output_destination_index_;
  RankGroup *ranks_;
  const bool broadcast_;
  void roundFinished(bool last);   // by the round's work order: the next one may be issued / the operator is done
  std::mutex mutex_;
  PartitionedBlockIds input_;
  std::vector<std::size_t> consumed_;     // per partition: blocks of input_ the rounds so far have taken
  std::atomic<std::uint64_t> bytes_sent_{0};
  bool round_in_flight_ = false, finished_ = false;
  std::size_t rounds_ = 0;
};

// The partial aggregation states of all ranks merged, in front of FinalizeAggregationOperator (a pipeline breaker after
// the AggregationOperator): one work order, state partitions in order, AggregationOperationState::mergeAcrossRanks each.
// The counterpart of merging the partitions' / threads' tables at finalize (storage/AggregationOperationState.cpp:831-843,
// 925-948) when a partition is a GPU.
class ExchangeAggregationStatesOperator : public RelationalOperator {
 public:
  ExchangeAggregationStatesOperator(std::size_t query_id, QueryContext::aggregation_state_id aggr_state_index, std::size_t num_partitions,
                                    RankGroup *ranks)
      : RelationalOperator(query_id, num_partitions), aggr_state_index_(aggr_state_index), ranks_(ranks) {}
  OperatorType getOperatorType() const override { return kExchangeAggregationStates; }
  bool isCollective() const override { return true; }
  std::string getName() const override { return "ExchangeAggregationStatesOperator"; }
  bool getAllWorkOrders(WorkOrdersContainer *container, QueryContext *query_context, StorageManager *storage_manager,
                        const tmb::client_id scheduler_client_id, tmb::MessageBus *bus) override;

 private:
  const QueryContext::aggregation_state_id aggr_state_index_;
  RankGroup *ranks_;
  bool work_generated_ = false;
};

// ---------------------------------------------------------------------------
// Scheduler: ForemanSingleNode + Worker (query_execution/ForemanSingleNode.cpp:102-178,
// Worker.cpp:54-139), reduced to what the hot path needs: a DAG of operators,
// pipeline-breaking and streaming edges, N worker threads with one HIP stream each.
// ---------------------------------------------------------------------------
class QueryPlan {
 public:
  std::size_t addRelationalOperator(RelationalOperator *op);  // takes ownership
  // is_pipeline_breaker: consumer may not start before producer has finished entirely
  // (BuildHash -> HashJoin, Aggregation -> Finalize; ExecutionGenerator.cpp:1110-1124, 2027-2046);
  // otherwise the producer's filled output blocks are fed to the consumer as they appear.
  void addDirectDependency(std::size_t consumer, std::size_t producer, bool is_pipeline_breaker);
  std::size_t size() const { return operators_.size(); }
  RelationalOperator *getOperator(std::size_t i) const { return operators_.at(i).get(); }
  struct Edge { std::size_t producer; bool breaker; };
  const std::vector<Edge> &dependencies(std::size_t consumer) const { return deps_.at(consumer); }

 private:
  std::vector<std::unique_ptr<RelationalOperator>> operators_;
  std::vector<std::vector<Edge>> deps_;
};

struct WorkOrderTimeEntry {  // --profile_and_report_workorder_perf (PolicyEnforcerBase.cpp:66-68)
  std::size_t worker_id;
  std::size_t operator_index;
  std::uint64_t start_us;
  std::uint64_t end_us;
};

class ForemanSingleNode {
 public:
  ForemanSingleNode(QueryPlan *plan, QueryContext *query_context, StorageManager *storage_manager,
                    std::size_t num_workers);
  void run();  // admits the query, dispatches work orders until every operator has finished
  const std::vector<WorkOrderTimeEntry> &getWorkOrderProfilingResults() const { return profile_; }

 private:
  void workerMain(std::size_t worker_id);
  QueryPlan *plan_;
  QueryContext *query_context_;
  StorageManager *storage_manager_;
  std::size_t num_workers_;
  tmb::MessageBus bus_;

  std::mutex mutex_;
  std::condition_variable cv_work_;
  std::condition_variable cv_done_;
  struct Item { WorkOrder *wo; std::size_t op; bool exclusive; };
  std::size_t exclusive_running_ = 0, shared_running_ = 0;   // work orders on Workers right now, by prefersExclusiveDevice()
  std::deque<Item> ready_;
  std::uint64_t events_ = 0;              // work orders finished + blocks returned to a destination (what the Foreman sleeps on)
  std::vector<std::size_t> outstanding_;  // dispatched but unfinished work orders per operator
  std::vector<std::size_t> executing_;    // of those: on a Worker right now
  bool shutting_down_ = false;
  std::string worker_error_;
  std::vector<WorkOrderTimeEntry> profile_;
};

}  // namespace quickstep

#endif  // QUICKSTEP_AMD_HOST_QUICKSTEP_GPU_HPP_
