// query_context.cpp — predicates, insert destinations, the aggregation state, QueryContext, LIP filters, the work-order container (see quickstep_gpu.hpp; what the files share: quickstep_gpu_internal.hpp)
#include "quickstep_gpu_internal.hpp"

namespace quickstep {

// ---------------------------------------------------------------------------
// Predicate
// ---------------------------------------------------------------------------
void *Predicate::getMatchesForBlock(const StorageBlock &block, std::int64_t *num_matches, const std::uint64_t *filter) const {
  const std::int64_t n = block.numTuples();
  const std::size_t words = static_cast<std::size_t>((n + 63) / 64);
  void *current = nullptr, *next = nullptr, *count = nullptr;
  CheckStatus(qsx_device_alloc(words * 8 + 8, &current), "qsx_device_alloc(bitmap)");
  CheckStatus(qsx_device_alloc(words * 8 + 8, &next), "qsx_device_alloc(bitmap)");
  CheckStatus(qsx_device_alloc(8, &count), "qsx_device_alloc(count)");
  bool first = true;
  bool recount = false;
  for (const ComparisonPredicate &term : conjuncts) {
    const Type &t = block.getRelation().getAttributeType(term.attribute);
    const std::uint64_t *in = first ? filter : static_cast<const std::uint64_t *>(current);
    // conjunctions chain the filter through their children (short-circuit)
    if (const CompressedAttribute *c = block.compressedAttribute(term.attribute)) {
      // CompressedTupleStorageSubBlock::getMatchesForPredicate (storage/CompressedTupleStorageSubBlock.cpp:160-250):
      // rewrite to a comparison on codes, scan the code stripe
      const PredicateTransformResult r = TransformPredicateOnCompressedAttribute(*c, t.id, term.comparison, term.literal);
      if (r.type == PredicateTransformResult::kAll || r.type == PredicateTransformResult::kNone) {
        // every code / no code: code >= 0 resp. code < 0
        CheckStatus(qsx_select_codes(c->code_width, c->codes, n, r.type == PredicateTransformResult::kAll ? QSX_CODE_GE : QSX_CODE_LT, 0, 0,
                                     in, static_cast<std::uint64_t *>(next), static_cast<std::int64_t *>(count), CurrentStream()),
                    "qsx_select_codes");
      } else if (term.attribute == block.sortColumn()) {
        // the sort column of a compressed store: the codes ascend, the matches are one range (the sort-column branches of
        // CompressedColumnStoreTupleStorageSubBlock.cpp:420-760)
        CheckStatus(qsx_select_codes_sorted(c->code_width, c->codes, n, r.comp, r.first_literal, r.second_literal, in,
                                            static_cast<std::uint64_t *>(next), static_cast<std::int64_t *>(count), CurrentStream()),
                    "qsx_select_codes_sorted");
      } else {
        CheckStatus(qsx_select_codes(c->code_width, c->codes, n, r.comp, r.first_literal, r.second_literal, in,
                                     static_cast<std::uint64_t *>(next), static_cast<std::int64_t *>(count), CurrentStream()),
                    "qsx_select_codes");
      }
    } else if (t.id == kChar) {
      // CHAR(n) OP string literal (AsciiStringUncheckedComparator, AsciiStringComparators.hpp:218-251)
      CheckStatus(qsx_select_cmp_char(block.stripe(term.attribute), t.width, n, static_cast<int>(term.comparison), term.literal.text.data(),
                                      static_cast<int>(term.literal.text.size()), in, static_cast<std::uint64_t *>(next),
                                      static_cast<std::int64_t *>(count), CurrentStream()), "qsx_select_cmp_char");
    } else if (term.attribute == block.sortColumn()) {
      // the block is sorted on this attribute: SortColumnPredicateEvaluator (storage/ColumnStoreUtil.cpp:40-280)
      CheckStatus(qsx_select_cmp_sorted(t.id, block.stripe(term.attribute), n, static_cast<int>(term.comparison), &term.literal.v, in,
                                        static_cast<std::uint64_t *>(next), static_cast<std::int64_t *>(count), CurrentStream()),
                  "qsx_select_cmp_sorted");
    } else {
      CheckStatus(qsx_select_cmp(t.id, block.stripe(term.attribute), n, static_cast<int>(term.comparison), &term.literal.v, in,
                                 static_cast<std::uint64_t *>(next), static_cast<std::int64_t *>(count), CurrentStream()),
                  "qsx_select_cmp");
    }
    if (block.nullBitmap(term.attribute) != nullptr && n > 0) {
      // a comparison with NULL is not true (LiteralComparators-inl.hpp:330-370: the nullable variants test the value
      // pointer first): the stripe holds an arbitrary value under a NULL, so its match is taken back
      CheckStatus(qsx_bitmap_combine(2, static_cast<const std::uint64_t *>(next), block.nullBitmap(term.attribute), n,
                                     static_cast<std::uint64_t *>(next), CurrentStream()), "qsx_bitmap_combine");
      recount = true;
    }
    std::swap(current, next);
    first = false;
  }
  if (recount) {
    CheckStatus(qsx_bitmap_count(static_cast<const std::uint64_t *>(current), n, static_cast<std::int64_t *>(count), CurrentStream()),
                "qsx_bitmap_count");
  }
  if (first) {  // empty conjunction: every tuple (of the filter) matches
    CheckStatus(qsx_memset_device(next, 0xFF, words * 8, CurrentStream()), "qsx_memset_device");
    CheckStatus(qsx_bitmap_combine(0, static_cast<const std::uint64_t *>(next),
                                   filter != nullptr ? filter : static_cast<const std::uint64_t *>(next), n,
                                   static_cast<std::uint64_t *>(current), CurrentStream()), "qsx_bitmap_combine");
    CheckStatus(qsx_bitmap_count(static_cast<const std::uint64_t *>(current), n, static_cast<std::int64_t *>(count),
                                 CurrentStream()), "qsx_bitmap_count");
  }
  *num_matches = ReadCount(count);
  qsx_device_free(next);
  qsx_device_free(count);
  return current;
}

// ---------------------------------------------------------------------------
// A predicate over a RUN of blocks (the SelectOperator's work orders over runs; the aggregation over runs of compressed blocks)
// ---------------------------------------------------------------------------
// Can the run forms evaluate `predicate` over these blocks in one launch per term?  Terms against a literal on attributes
// without NULLs, every block coding and ordering a term's attribute like the first non-empty one.
bool RunPredicateCovers(const Predicate &predicate, const std::vector<BlockReference> &blocks) {
  const StorageBlock *reference_block = nullptr;   // the first non-empty block: what the others have to agree with
  for (const BlockReference &block : blocks) {
    const StorageBlock &b = *block;
    for (const ComparisonPredicate &term : predicate.conjuncts) {
      const Type &t = b.getRelation().getAttributeType(term.attribute);
      if (term.rhs_attribute != kInvalidAttributeID || b.nullBitmap(term.attribute) != nullptr) return false;
      if (b.numTuples() == 0) continue;                 // (an empty block has neither codes nor an order to agree on)
      // a term on the blocks' sort column is a per-block binary search (also on the code stripe of a compressed sort column), a
      // term on a compressed attribute a scan of the code stripes with the comparison rewritten per block
      if (t.id == kChar && b.compressedAttribute(term.attribute) == nullptr && term.attribute == b.sortColumn()) return false;
      if (reference_block == nullptr) reference_block = &b;
      const StorageBlock &f = *reference_block;
      if ((term.attribute == b.sortColumn()) != (term.attribute == f.sortColumn())) return false;
      const CompressedAttribute *cb = b.compressedAttribute(term.attribute), *cf = f.compressedAttribute(term.attribute);
      if ((cb != nullptr) != (cf != nullptr) || (cb != nullptr && cb->code_width != cf->code_width)) return false;
    }
  }
  return true;
}

// The TupleIdSequences of the run's blocks under `predicate` (RunPredicateCovers said yes), every term one launch over the run,
// chained like a conjunction behind `in_filters` (per block, entries may be null; or nullptr): out->bitmaps[b] is block b's
// bitmap, out->counts the per-block match counts of the last term (device memory).
void RunPredicateMatches(const Predicate &predicate, const std::vector<BlockReference> &blocks, const std::vector<std::int64_t> &rows,
                         const std::uint64_t *const *in_filters, RunMatches *out) {
  const std::size_t nb = blocks.size();
  std::size_t bitmap_words = 0;
  const StorageBlock *reference_block = blocks.empty() ? nullptr : blocks.front().get();
  for (std::size_t b = 0; b < nb; ++b) {
    bitmap_words += static_cast<std::size_t>((rows[b] + 63) / 64) + 1;
    if (reference_block != nullptr && reference_block->numTuples() == 0 && blocks[b]->numTuples() != 0) reference_block = blocks[b].get();
  }
  // two sets of per-block bitmaps in two allocations, the terms ping-pong between them
  out->set_a.reset(new DeviceBuffer(bitmap_words * 8 + 8));
  out->set_b.reset(new DeviceBuffer(bitmap_words * 8 + 8));
  out->counts.reset(new DeviceBuffer(nb * 8 + 8));
  std::vector<std::uint64_t *> cur(nb), nxt(nb);
  std::size_t at = 0;
  for (std::size_t b = 0; b < nb; ++b) {
    cur[b] = static_cast<std::uint64_t *>(out->set_a->ptr) + at;
    nxt[b] = static_cast<std::uint64_t *>(out->set_b->ptr) + at;
    at += static_cast<std::size_t>((rows[b] + 63) / 64) + 1;
  }
  std::vector<const void *> stripes(nb);
  bool first = true;
  for (const ComparisonPredicate &term : predicate.conjuncts) {
    const Type &t = blocks.front()->getRelation().getAttributeType(term.attribute);
    const StorageBlock &ref = *reference_block;
    if (ref.compressedAttribute(term.attribute) == nullptr) {   // (a compressed sort column is searched on its codes)
      for (std::size_t b = 0; b < nb; ++b) stripes[b] = blocks[b]->stripe(term.attribute);
    }
    const std::uint64_t *const *in = first ? in_filters : reinterpret_cast<const std::uint64_t *const *>(cur.data());
    const bool on_sort_column = term.attribute == ref.sortColumn();
    if (on_sort_column && ref.compressedAttribute(term.attribute) != nullptr) {
      // the sort column of compressed blocks: the comparison rewritten on every block's own codes
      // (CompressedTupleStorageSubBlock::getMatchesForPredicate), then one search per block on the code stripes
      std::vector<std::int32_t> ops(nb);
      std::vector<std::uint32_t> firsts(nb), seconds(nb);
      for (std::size_t b = 0; b < nb; ++b) {
        const CompressedAttribute *c = blocks[b]->compressedAttribute(term.attribute);
        if (c == nullptr) {   // an empty block
          stripes[b] = nullptr;
          ops[b] = QSX_CODE_LT;
          firsts[b] = seconds[b] = 0;
          continue;
        }
        const PredicateTransformResult r = TransformPredicateOnCompressedAttribute(*c, t.id, term.comparison, term.literal);
        stripes[b] = c->codes;
        if (r.type == PredicateTransformResult::kAll || r.type == PredicateTransformResult::kNone) {
          ops[b] = r.type == PredicateTransformResult::kAll ? QSX_CODE_GE : QSX_CODE_LT;   // every code / no code
          firsts[b] = seconds[b] = 0;
        } else {
          ops[b] = r.comp;
          firsts[b] = r.first_literal;
          seconds[b] = r.second_literal;
        }
      }
      CheckStatus(qsx_select_codes_sorted_blocks(ref.compressedAttribute(term.attribute)->code_width, static_cast<std::int64_t>(nb),
                                                 rows.data(), stripes.data(), ops.data(), firsts.data(), seconds.data(), in, nxt.data(),
                                                 static_cast<std::int64_t *>(out->counts->ptr), CurrentStream()), "qsx_select_codes_sorted_blocks");
    } else if (ref.compressedAttribute(term.attribute) != nullptr) {
      // a compressed attribute: every block's code stripe scanned with the comparison rewritten on that block's codes
      std::vector<std::int32_t> ops(nb);
      std::vector<std::uint32_t> firsts(nb), seconds(nb);
      for (std::size_t b = 0; b < nb; ++b) {
        const CompressedAttribute *c = blocks[b]->compressedAttribute(term.attribute);
        if (c == nullptr) {   // an empty block
          stripes[b] = nullptr;
          ops[b] = QSX_CODE_LT;
          firsts[b] = seconds[b] = 0;
          continue;
        }
        const PredicateTransformResult r = TransformPredicateOnCompressedAttribute(*c, t.id, term.comparison, term.literal);
        stripes[b] = c->codes;
        if (r.type == PredicateTransformResult::kAll || r.type == PredicateTransformResult::kNone) {
          ops[b] = r.type == PredicateTransformResult::kAll ? QSX_CODE_GE : QSX_CODE_LT;
          firsts[b] = seconds[b] = 0;
        } else {
          ops[b] = r.comp;
          firsts[b] = r.first_literal;
          seconds[b] = r.second_literal;
        }
      }
      CheckStatus(qsx_select_codes_blocks(ref.compressedAttribute(term.attribute)->code_width, static_cast<std::int64_t>(nb), rows.data(),
                                          stripes.data(), ops.data(), firsts.data(), seconds.data(), in, nxt.data(),
                                          static_cast<std::int64_t *>(out->counts->ptr), CurrentStream()), "qsx_select_codes_blocks");
    } else if (t.id == kChar) {
      // CHAR(n) OP string literal on plain stripes (AsciiStringUncheckedComparator, AsciiStringComparators.hpp:218-251)
      CheckStatus(qsx_select_cmp_char_blocks(t.width, static_cast<std::int64_t>(nb), rows.data(), stripes.data(), static_cast<int>(term.comparison),
                                             term.literal.text.data(), static_cast<int>(term.literal.text.size()), in, nxt.data(),
                                             static_cast<std::int64_t *>(out->counts->ptr), CurrentStream()), "qsx_select_cmp_char_blocks");
    } else if (on_sort_column) {
      // SortColumnPredicateEvaluator (storage/ColumnStoreUtil.cpp:40-280), one search per block
      CheckStatus(qsx_select_cmp_sorted_blocks(t.id, static_cast<std::int64_t>(nb), rows.data(), stripes.data(), static_cast<int>(term.comparison),
                                               &term.literal.v, in, nxt.data(), static_cast<std::int64_t *>(out->counts->ptr), CurrentStream()),
                  "qsx_select_cmp_sorted_blocks");
    } else {
      CheckStatus(qsx_select_cmp_blocks(t.id, static_cast<std::int64_t>(nb), rows.data(), stripes.data(), static_cast<int>(term.comparison),
                                        &term.literal.v, in, nxt.data(), static_cast<std::int64_t *>(out->counts->ptr), CurrentStream()),
                  "qsx_select_cmp_blocks");
    }
    std::swap(cur, nxt);
    first = false;
  }
  out->bitmaps = cur;
}

// ---------------------------------------------------------------------------
// InsertDestination
// ---------------------------------------------------------------------------
BlockReference InsertDestination::getBlockForInsertion(std::int64_t capacity, block_id *id) {
  *id = storage_manager_->createBlock(relation_, capacity);
  return storage_manager_->getBlock(*id);
}
void InsertDestination::returnBlock(block_id id, std::int64_t num_tuples, partition_id input_partition) {
  if (isPartitionAware()) {
    repartitionBlock(id, num_tuples);
    return;
  }
  BlockReference block = storage_manager_->getBlock(id);
  block->setNumTuples(num_tuples);
  block->setFirstRow(storage_manager_->reserveRows(relation_->getID(), num_tuples));
  {
    std::lock_guard<std::mutex> lock(mutex_);
    touched_.push_back(TouchedBlock{id, input_partition});
    if (relation_->hasPartitionScheme()) {
      relation_->addBlockToPartition(id, input_partition);   // the output keeps the input's partitioning (no repartition)
    } else {
      relation_->addBlock(id);
    }
  }
  if (block_returned_) block_returned_();
}

// bulkInsertTuples of a PartitionAwareInsertDestination over every tuple of a run of input blocks: one K9 pass that finds the
// blocks' stripes through a table — against a copy of the run into one block (two more trips through HBM) and K9 over that.
bool InsertDestination::insertRunRepartitioned(const std::vector<BlockReference> &blocks, const std::vector<attribute_id> &attributes) {
  if (!isPartitionAware() || blocks.empty() || attributes.size() != relation_->size() || attributes.size() > QSX_MAX_COLUMNS) return false;
  const std::size_t P = num_partitions_;
  const Type &key_type = relation_->getAttributeType(partition_attribute_);
  if ((key_type.id != kInt && key_type.id != kLong) || P > 64) return false;   // (repartitionBlock reports these)
  for (std::size_t a = 0; a < relation_->size(); ++a) {
    const Type &t = relation_->getAttributeType(static_cast<attribute_id>(a));
    if (t.nullable || (t.width != 1 && t.width != 2 && t.width != 4 && t.width != 8)) return false;
  }
  std::vector<std::int64_t> rows;
  std::vector<const void *> keys, cols;
  std::vector<std::int32_t> widths;
  std::int64_t total = 0;
  for (const BlockReference &b : blocks) {
    for (attribute_id a : attributes) {
      if (b->nullBitmap(a) != nullptr) return false;
    }
    rows.push_back(b->numTuples());
    total += b->numTuples();
    keys.push_back(b->numTuples() > 0 ? b->stripe(attributes[partition_attribute_]) : nullptr);
    for (attribute_id a : attributes) cols.push_back(b->numTuples() > 0 ? b->stripe(a) : nullptr);
  }
  if (total == 0) return true;
  for (std::size_t a = 0; a < relation_->size(); ++a) widths.push_back(relation_->getAttributeType(static_cast<attribute_id>(a)).width);
  block_id scattered_id;
  BlockReference scattered = getBlockForInsertion(total, &scattered_id);
  scattered->setNumTuples(total);
  std::vector<void *> outs;
  for (std::size_t a = 0; a < relation_->size(); ++a) outs.push_back(scattered->stripe(static_cast<attribute_id>(a)));
  const std::size_t ws_bytes = qsx_partition_blocks_workspace_bytes(total, static_cast<std::int64_t>(blocks.size()), static_cast<int>(P));
  DeviceBuffer ws(ws_bytes + 8), offsets_dev((P + 1) * 8);
  CheckStatus(qsx_partition_scatter_blocks(key_type.id, static_cast<std::int64_t>(blocks.size()), rows.data(), keys.data(), static_cast<int>(P),
                                           static_cast<int>(widths.size()), cols.data(), widths.data(), outs.data(),
                                           static_cast<std::int64_t *>(offsets_dev.ptr), ws.ptr, ws_bytes, CurrentStream()),
              "qsx_partition_scatter_blocks");
  std::vector<std::int64_t> offsets(P + 1);
  CheckStatus(qsx_copy_to_host(offsets.data(), offsets_dev.ptr, (P + 1) * 8, CurrentStream()), "qsx_copy_to_host");
  CheckStatus(qsx_stream_synchronize(CurrentStream()), "qsx_stream_synchronize");   // (the work order's wait; ws goes out of scope)
  std::vector<std::pair<block_id, partition_id>> made;
  for (std::size_t p = 0; p < P; ++p) {
    const std::int64_t first = offsets[p], rows_p = offsets[p + 1] - offsets[p];
    if (rows_p == 0) continue;
    const block_id view_id = storage_manager_->createViewBlock(scattered_id, first, rows_p);
    storage_manager_->getBlock(view_id)->setFirstRow(storage_manager_->reserveRows(relation_->getID(), rows_p));
    made.emplace_back(view_id, p);
  }
  storage_manager_->deleteBlockOrBlobFile(scattered_id);      // (the views keep the scattered block alive)
  {
    std::lock_guard<std::mutex> lock(mutex_);
    for (const auto &m : made) {
      touched_.push_back(TouchedBlock{m.first, m.second});
      relation_->addBlockToPartition(m.first, m.second);
    }
  }
  if (block_returned_ && !made.empty()) block_returned_();
  return true;
}

// bulkInsertTuples of a PartitionAwareInsertDestination (storage/InsertDestination.hpp:560-660), on a whole block at once:
// K9 scatters the columns of 1 / 2 / 4 / 8 bytes and a row-number column by the partition of the partition attribute; wider
// columns (CHAR(n)) and the null bits follow through the scattered row numbers; the scattered block is then cut into one
// block per partition (views: no copy).
void InsertDestination::repartitionBlock(block_id id, std::int64_t num_tuples) {
  BlockReference src = storage_manager_->getBlock(id);
  src->setNumTuples(num_tuples);
  const std::size_t P = num_partitions_;
  const Type &key_type = relation_->getAttributeType(partition_attribute_);
  if (key_type.id != kInt && key_type.id != kLong) {
    throw ExecutionError("PartitionAwareInsertDestination: the partition attribute must be INT or LONG", QSX_ERR_UNSUPPORTED);
  }
  if (P > 64) throw ExecutionError("PartitionAwareInsertDestination: more than 64 partitions", QSX_ERR_UNSUPPORTED);
  if (num_tuples == 0) {
    storage_manager_->deleteBlockOrBlobFile(id);
    return;
  }
  block_id scattered_id;
  BlockReference scattered = getBlockForInsertion(num_tuples, &scattered_id);
  scattered->setNumTuples(num_tuples);
  bool need_rows = false;
  std::vector<const void *> cols;
  std::vector<void *> outs;
  std::vector<std::int32_t> widths;
  for (std::size_t a = 0; a < relation_->size(); ++a) {
    const Type &t = relation_->getAttributeType(static_cast<attribute_id>(a));
    if (t.nullable) need_rows = true;
    if (t.width == 1 || t.width == 2 || t.width == 4 || t.width == 8) {
      cols.push_back(src->stripe(static_cast<attribute_id>(a)));
      outs.push_back(scattered->stripe(static_cast<attribute_id>(a)));
      widths.push_back(t.width);
    } else {
      need_rows = true;
    }
  }
  std::unique_ptr<DeviceBuffer> rows, rows_scattered;
  if (need_rows) {
    // row numbers 0 .. n-1: the tuple ids of an all-ones TupleIdSequence (NOT of a zeroed one: trailing bits stay zero)
    const std::size_t words = static_cast<std::size_t>((num_tuples + 63) / 64) + 1;
    DeviceBuffer zero(words * 8), ones(words * 8), count(8);
    CheckStatus(qsx_memset_device(zero.ptr, 0, words * 8, CurrentStream()), "qsx_memset_device");
    CheckStatus(qsx_bitmap_combine(3, static_cast<const std::uint64_t *>(zero.ptr), nullptr, num_tuples, static_cast<std::uint64_t *>(ones.ptr),
                                   CurrentStream()), "qsx_bitmap_combine");
    rows.reset(new DeviceBuffer(static_cast<std::size_t>(num_tuples) * 4 + 8));
    rows_scattered.reset(new DeviceBuffer(static_cast<std::size_t>(num_tuples) * 4 + 8));
    const std::size_t tws = qsx_compact_workspace_bytes(num_tuples);
    DeviceBuffer tw(tws + 8);
    CheckStatus(qsx_bitmap_to_tids(static_cast<const std::uint64_t *>(ones.ptr), num_tuples, 0, static_cast<std::int32_t *>(rows->ptr),
                                   static_cast<std::int64_t *>(count.ptr), tw.ptr, tws, CurrentStream()), "qsx_bitmap_to_tids");
    cols.push_back(rows->ptr);
    outs.push_back(rows_scattered->ptr);
    widths.push_back(4);
    CheckStatus(qsx_stream_synchronize(CurrentStream()), "qsx_stream_synchronize");   // (zero / ones / tw go out of scope)
  }
  const std::size_t ws_bytes = qsx_partition_workspace_bytes(num_tuples, static_cast<int>(P));
  DeviceBuffer ws(ws_bytes + 8), offsets_dev((P + 1) * 8);
  CheckStatus(qsx_partition_scatter(key_type.id, src->stripe(partition_attribute_), num_tuples, static_cast<int>(P), static_cast<int>(cols.size()),
                                    cols.data(), widths.data(), outs.data(), static_cast<std::int64_t *>(offsets_dev.ptr), ws.ptr, ws_bytes,
                                    CurrentStream()), "qsx_partition_scatter");
  std::vector<std::int64_t> offsets(P + 1);
  CheckStatus(qsx_copy_to_host(offsets.data(), offsets_dev.ptr, (P + 1) * 8, CurrentStream()), "qsx_copy_to_host");
  for (std::size_t a = 0; a < relation_->size(); ++a) {
    const Type &t = relation_->getAttributeType(static_cast<attribute_id>(a));
    if (t.width == 1 || t.width == 2 || t.width == 4 || t.width == 8) continue;
    CheckStatus(qsx_gather(t.width, src->stripe(static_cast<attribute_id>(a)), static_cast<const std::int32_t *>(rows_scattered->ptr), num_tuples,
                           scattered->stripe(static_cast<attribute_id>(a)), CurrentStream()), "qsx_gather");
  }
  std::vector<std::pair<block_id, partition_id>> made;
  for (std::size_t p = 0; p < P; ++p) {
    const std::int64_t first = offsets[p], rows_p = offsets[p + 1] - offsets[p];
    if (rows_p == 0) continue;
    const block_id view_id = storage_manager_->createViewBlock(scattered_id, first, rows_p);
    BlockReference view = storage_manager_->getBlock(view_id);
    for (std::size_t a = 0; a < relation_->size(); ++a) {
      std::uint64_t *dst = view->nullBitmap(static_cast<attribute_id>(a));
      if (dst == nullptr) continue;
      const std::uint64_t *bits = src->nullBitmap(static_cast<attribute_id>(a));
      const std::int64_t zero_row = 0;
      CheckStatus(qsx_bitmap_gather_segmented(1, &bits, &zero_row, static_cast<const std::int32_t *>(rows_scattered->ptr) + first, rows_p, dst,
                                              CurrentStream()), "qsx_bitmap_gather_segmented");
    }
    view->setFirstRow(storage_manager_->reserveRows(relation_->getID(), rows_p));
    made.emplace_back(view_id, p);
  }
  CheckStatus(qsx_stream_synchronize(CurrentStream()), "qsx_stream_synchronize");   // before the source block goes
  storage_manager_->deleteBlockOrBlobFile(id);
  storage_manager_->deleteBlockOrBlobFile(scattered_id);      // (the views keep the scattered block alive)
  {
    std::lock_guard<std::mutex> lock(mutex_);
    for (const auto &m : made) {
      touched_.push_back(TouchedBlock{m.first, m.second});
      relation_->addBlockToPartition(m.first, m.second);
    }
  }
  if (block_returned_ && !made.empty()) block_returned_();
}
std::vector<block_id> InsertDestination::getTouchedBlocks() const {
  std::lock_guard<std::mutex> lock(mutex_);
  std::vector<block_id> ids;
  for (const TouchedBlock &t : touched_) ids.push_back(t.id);
  return ids;
}
std::vector<InsertDestination::TouchedBlock> InsertDestination::getTouchedBlocksWithPartitions() const {
  std::lock_guard<std::mutex> lock(mutex_);
  return touched_;
}
std::vector<InsertDestination::TouchedBlock> InsertDestination::getTouchedBlocksSince(std::size_t from) const {
  std::lock_guard<std::mutex> lock(mutex_);
  if (from >= touched_.size()) return {};
  return std::vector<TouchedBlock>(touched_.begin() + static_cast<std::ptrdiff_t>(from), touched_.end());
}
std::size_t InsertDestination::numTouchedBlocks() const {
  std::lock_guard<std::mutex> lock(mutex_);
  return touched_.size();
}

// ---------------------------------------------------------------------------
// AggregationOperationState
// ---------------------------------------------------------------------------
struct AggregationOperationState::Distinctify {
  std::size_t agg_index = 0;            // position in spec.aggregates
  std::vector<attribute_id> attrs;      // group-by..., argument (kInvalidAttributeID: the argument is `expression`)
  std::vector<Type> types;
  ScalarPtr expression;                 // DISTINCT over an arithmetic expression: evaluated per block into a DOUBLE column
  std::mutex mutex;                     // many AggregationWorkOrders append concurrently
  struct Chunk {
    std::vector<std::unique_ptr<DeviceBuffer>> cols;
    std::int64_t rows = 0;
  };
  std::vector<Chunk> chunks;
};

namespace {
qsx_agg_fn_t AggFn(AggregationID id) {
  switch (id) {
    case AggregationID::kCount: return QSX_AGG_COUNT_STAR;
    case AggregationID::kSum: return QSX_AGG_SUM;
    case AggregationID::kAvg: return QSX_AGG_AVG;
    case AggregationID::kMin: return QSX_AGG_MIN;
    default: return QSX_AGG_MAX;
  }
}
// Result type of an aggregate (AggregationHandle{Count,Sum,Avg,Min,Max}::getResultType).
Type AggResultType(AggregationID id, const Type &argument) {
  switch (id) {
    case AggregationID::kCount: return Type::Long();
    case AggregationID::kSum: return (argument.id == kInt || argument.id == kLong) ? Type::Long() : Type::Double();
    case AggregationID::kAvg: return Type::Double();
    default: return argument;
  }
}
}  // namespace

TypeID ScalarResultType(const ScalarPtr &scalar, const CatalogRelation &relation) {
  if (scalar == nullptr) throw ExecutionError("ScalarResultType: null scalar", QSX_ERR_INVALID_ARGUMENT);
  switch (scalar->kind) {
    case Scalar::kAttribute: {
      const TypeID t = relation.getAttributeType(scalar->attribute).id;
      if (t != kInt && t != kLong && t != kFloat && t != kDouble) {
        throw ExecutionError("arithmetic over a non-numeric attribute", QSX_ERR_UNSUPPORTED);
      }
      return t == kFloat ? kDouble : t;   // (FLOAT operands are evaluated in double)
    }
    case Scalar::kLiteral:
      return scalar->literal_type;
    default: {
      const TypeID l = ScalarResultType(scalar->left, relation), r = ScalarResultType(scalar->right, relation);
      if (l == kDouble || r == kDouble) return kDouble;
      return l == kLong || r == kLong ? kLong : kInt;
    }
  }
}

qsx_operand_t ExpressionFlattener::add(const ScalarPtr &scalar) {
  if (scalar == nullptr) throw ExecutionError("ExpressionFlattener: null scalar", QSX_ERR_INVALID_ARGUMENT);
  switch (scalar->kind) {
    case Scalar::kAttribute:
      return qsx_operand_t{QSX_OPD_COLUMN, column_of_(scalar->attribute)};
    case Scalar::kLiteral: {
      for (std::size_t i = 0; i < consts_.size(); ++i) {
        if (std::memcmp(&consts_[i], &scalar->literal, sizeof(double)) == 0) return qsx_operand_t{QSX_OPD_CONST, static_cast<std::int32_t>(i)};
      }
      if (consts_.size() >= QSX_MAX_CONSTS) throw ExecutionError("expression: too many distinct literals", QSX_ERR_UNSUPPORTED);
      consts_.push_back(scalar->literal);
      return qsx_operand_t{QSX_OPD_CONST, static_cast<std::int32_t>(consts_.size() - 1)};
    }
    default: {
      const qsx_operand_t a = add(scalar->left), b = add(scalar->right);
      const std::int32_t op = static_cast<std::int32_t>(scalar->operation);   // kAdd .. kDivide = QSX_EX_ADD .. QSX_EX_DIV
      for (const qsx_expr_instr_t &in : instrs_) {   // the same node again (shared subexpression): its temp
        if (in.op == op && in.a.kind == a.kind && in.a.index == a.index && in.b.kind == b.kind && in.b.index == b.index) {
          return qsx_operand_t{QSX_OPD_TEMP, in.dst};
        }
      }
      if (instrs_.size() >= QSX_MAX_INSTRS || instrs_.size() >= QSX_MAX_TEMPS) {
        throw ExecutionError("expression: more nodes than the kernel's program holds", QSX_ERR_UNSUPPORTED);
      }
      qsx_expr_instr_t in;
      in.op = op;
      in.dst = static_cast<std::int32_t>(instrs_.size());   // one temp per node
      in.a = a;
      in.b = b;
      instrs_.push_back(in);
      return qsx_operand_t{QSX_OPD_TEMP, in.dst};
    }
  }
}

AggregationOperationState::AggregationOperationState(const AggregationStateSpec &spec) : spec_(spec) {
  // the library must have been built from the header this file was compiled against
  if (qsx_abi_version() != QSX_ABI_VERSION || qsx_abi_sizeof_agg_config() != sizeof(qsx_agg_config_t)) {
    throw ExecutionError("libqsx.so and include/qsx.h disagree on the ABI version / qsx_agg_config_t", QSX_ERR_INVALID_ARGUMENT);
  }
  std::memset(&config_, 0, sizeof(config_));
  const CatalogRelation &rel = *spec.input_relation;
  auto column_of = [&](attribute_id attr) -> int {
    for (std::size_t i = 0; i < column_attr_.size(); ++i) {
      if (column_attr_[i] == attr) return static_cast<int>(i);
    }
    if (column_attr_.size() >= QSX_MAX_COLUMNS) throw ExecutionError("AggregationOperationState: too many columns", QSX_ERR_UNSUPPORTED);
    const Type &t = rel.getAttributeType(attr);
    config_.column_type[column_attr_.size()] = t.id;
    config_.column_width[column_attr_.size()] = t.width;
    config_.column_nullable[column_attr_.size()] = t.nullable ? 1 : 0;
    column_attr_.push_back(attr);
    return static_cast<int>(column_attr_.size() - 1);
  };
  config_.strategy = spec.group_by.empty() ? QSX_AGG_SINGLE_STATE : spec.strategy;
  config_.num_keys = static_cast<int>(spec.group_by.size());
  for (std::size_t k = 0; k < spec.group_by.size(); ++k) config_.key_column[k] = column_of(spec.group_by[k]);
  int num_main = 0;
  ExpressionFlattener flattener(column_of);
  for (std::size_t a = 0; a < spec_.aggregates.size(); ++a) {
    if (spec_.aggregates[a].argument_expression != nullptr && spec_.aggregates[a].argument_expression->kind == Scalar::kAttribute) {
      spec_.aggregates[a].argument = spec_.aggregates[a].argument_expression->attribute;   // ScalarAttribute: the plain form
      spec_.aggregates[a].argument_expression = nullptr;
    }
    const AggregateSpec &ag = spec_.aggregates[a];
    if (ag.argument_expression != nullptr && ag.is_distinct) {
      // DISTINCT over an arithmetic expression (Distinct.test:58-72 COUNT(DISTINCT x % y) is such a query): the distinctify
      // key is (group-by..., value of the expression); the value column is computed per block (qsx_eval_expression)
      if (spec.group_by.size() + 1 > QSX_MAX_KEYS) throw ExecutionError("DISTINCT aggregate: too many group-by attributes", QSX_ERR_UNSUPPORTED);
      std::unique_ptr<Distinctify> d(new Distinctify);
      d->agg_index = a;
      d->attrs = spec.group_by;
      d->attrs.push_back(kInvalidAttributeID);
      for (attribute_id attr : spec.group_by) d->types.push_back(rel.getAttributeType(attr));
      d->types.push_back(Type::Double());
      d->expression = ag.argument_expression;
      distinctify_.push_back(std::move(d));
      main_agg_.push_back(-1);
      continue;
    }
    if (ag.argument_expression != nullptr) {
      // an arithmetic expression as the aggregate's argument: part of the state's expression program
      if (ag.function == AggregationID::kCount) throw ExecutionError("COUNT over an expression: pass COUNT(*)", QSX_ERR_UNSUPPORTED);
      const qsx_operand_t value = flattener.add(ag.argument_expression);
      if (value.kind == QSX_OPD_CONST) throw ExecutionError("aggregate over a literal", QSX_ERR_UNSUPPORTED);
      config_.aggs[num_main].fn = AggFn(ag.function);
      config_.aggs[num_main].arg = value;
      main_agg_.push_back(num_main++);
      continue;
    }
    if (ag.is_distinct) {
      // "Initialize the corresponding distinctify hash table if this is a DISTINCT aggregation" (:172-207):
      // key types = group-by types + argument types
      if (ag.argument == kInvalidAttributeID) throw ExecutionError("DISTINCT aggregate without an argument", QSX_ERR_INVALID_ARGUMENT);
      if (spec.group_by.size() + 1 > QSX_MAX_KEYS) throw ExecutionError("DISTINCT aggregate: too many group-by attributes", QSX_ERR_UNSUPPORTED);
      std::unique_ptr<Distinctify> d(new Distinctify);
      d->agg_index = a;
      d->attrs = spec.group_by;
      d->attrs.push_back(ag.argument);
      for (attribute_id attr : d->attrs) d->types.push_back(rel.getAttributeType(attr));
      distinctify_.push_back(std::move(d));
      main_agg_.push_back(-1);
      continue;
    }
    config_.aggs[num_main].fn = AggFn(ag.function);
    if (ag.function != AggregationID::kCount) {
      config_.aggs[num_main].arg.kind = QSX_OPD_COLUMN;
      config_.aggs[num_main].arg.index = column_of(ag.argument);
    } else if (ag.argument != kInvalidAttributeID && rel.getAttributeType(ag.argument).nullable) {
      // COUNT(x) over a nullable x counts the non-NULL values (AggregationHandleCount<false, true>); over a
      // non-nullable x it is COUNT(*)
      config_.aggs[num_main].fn = QSX_AGG_COUNT;
      config_.aggs[num_main].arg.kind = QSX_OPD_COLUMN;
      config_.aggs[num_main].arg.index = column_of(ag.argument);
    }
    main_agg_.push_back(num_main++);
  }
  config_.num_aggs = num_main;
  config_.num_instrs = static_cast<int>(flattener.instrs().size());
  for (std::size_t k = 0; k < flattener.instrs().size(); ++k) config_.instrs[k] = flattener.instrs()[k];
  for (std::size_t k = 0; k < flattener.consts().size(); ++k) config_.consts[k] = flattener.consts()[k];
  if (spec.predicate != nullptr) {
    int in_state = 0;
    for (const ComparisonPredicate &term : spec.predicate->conjuncts) {
      const Type &t = rel.getAttributeType(term.attribute);
      if (t.id == kChar || term.rhs_attribute != kInvalidAttributeID || in_state == QSX_MAX_PRED_TERMS) {
        external_predicate_.conjuncts.push_back(term);   // string comparisons, attribute-vs-attribute, overflow
        continue;
      }
      config_.pred[in_state].column = column_of(term.attribute);
      config_.pred[in_state].op = static_cast<int>(term.comparison);
      std::memcpy(&config_.pred[in_state].literal, &term.literal.v, sizeof(term.literal.v));
      ++in_state;
    }
    config_.num_pred_terms = in_state;
  }
  config_.num_columns = static_cast<int>(column_attr_.size());
  config_.est_groups = spec.estimated_num_groups;
  config_.num_entries = spec.collision_free_num_entries;
  // all_distinct_ (:126-127, 620-628): no upsert into the final table per block, it is filled from the distinctify tables
  if (num_main > 0 || distinctify_.empty()) CheckStatus(qsx_agg_state_create(&config_, &state_), "qsx_agg_state_create");
}

AggregationOperationState::~AggregationOperationState() {
  if (state_ != nullptr) qsx_agg_state_destroy(state_);
  if (coded_state_ != nullptr) qsx_agg_state_destroy(coded_state_);
}

void AggregationOperationState::initialize(std::size_t state_partition_id) {
  if (config_.strategy != QSX_AGG_COLLISION_FREE) {
    throw ExecutionError("AggregationOperationState::initialize() is not supported by this aggregation", QSX_ERR_UNSUPPORTED);
  }
  if (state_partition_id != 0) return;   // (the fill of slice 0 covers the whole allocation)
  if (state_ != nullptr) CheckStatus(qsx_agg_state_clear(state_, CurrentStream()), "qsx_agg_state_clear");
  if (coded_state_ != nullptr) CheckStatus(qsx_agg_state_clear(coded_state_, CurrentStream()), "qsx_agg_state_clear");
  CheckStatus(qsx_stream_synchronize(CurrentStream()), "qsx_stream_synchronize");   // the aggregation runs on other streams
}

void AggregationOperationState::aggregateBlock(const StorageBlock &block, const std::uint64_t *lip_filter) {
  const std::int64_t n = block.numTuples();
  if (!distinctify_.empty() && n > 0) {
    // insertValueAccessorIntoDistinctifyHashTable per DISTINCT aggregate (:522-528, 600-628), on the tuples that pass
    // the state's predicate and the LIP filters: the block's distinct (group-by..., argument) tuples are appended
    std::int64_t matches = n;
    void *selected = nullptr;
    if (spec_.predicate != nullptr) selected = spec_.predicate->getMatchesForBlock(block, &matches, lip_filter);
    const std::uint64_t *filter = selected != nullptr ? static_cast<const std::uint64_t *>(selected) : lip_filter;
    const std::size_t ws_bytes = qsx_sort_workspace_bytes(n);
    DeviceBuffer ws(ws_bytes), tids(static_cast<std::size_t>(n) * 4 + 16), count(8);
    for (auto &d : distinctify_) {
      const void *cols[QSX_MAX_KEYS];
      std::int32_t types[QSX_MAX_KEYS];
      std::unique_ptr<DeviceBuffer> expression_values;
      std::vector<attribute_id> nullable_sources;   // attributes whose NULLs keep a tuple out of the table
      for (std::size_t c = 0; c < d->attrs.size(); ++c) {
        types[c] = d->types[c].id;
        if (d->attrs[c] != kInvalidAttributeID) {
          cols[c] = block.stripe(d->attrs[c]);
          nullable_sources.push_back(d->attrs[c]);
          continue;
        }
        std::vector<attribute_id> attrs;
        ExpressionFlattener flattener([&](attribute_id a) {
          for (std::size_t k = 0; k < attrs.size(); ++k) if (attrs[k] == a) return static_cast<int>(k);
          attrs.push_back(a);
          return static_cast<int>(attrs.size() - 1);
        });
        const qsx_operand_t result = flattener.add(d->expression);
        const void *in_cols[QSX_MAX_COLUMNS];
        std::int32_t in_types[QSX_MAX_COLUMNS];
        if (attrs.size() > QSX_MAX_COLUMNS) throw ExecutionError("DISTINCT: expression over too many attributes", QSX_ERR_UNSUPPORTED);
        for (std::size_t k = 0; k < attrs.size(); ++k) {
          in_cols[k] = block.stripe(attrs[k]);
          in_types[k] = block.getRelation().getAttributeType(attrs[k]).id;
          nullable_sources.push_back(attrs[k]);   // a NULL operand makes the value NULL
        }
        double consts[QSX_MAX_CONSTS] = {};
        for (std::size_t k = 0; k < flattener.consts().size(); ++k) consts[k] = flattener.consts()[k];
        expression_values.reset(new DeviceBuffer(static_cast<std::size_t>(n) * 8 + 8));
        CheckStatus(qsx_eval_expression(static_cast<int>(attrs.size()), in_cols, in_types, static_cast<int>(flattener.instrs().size()),
                                        flattener.instrs().data(), consts, result, n, static_cast<double *>(expression_values->ptr),
                                        CurrentStream()), "qsx_eval_expression");
        cols[c] = expression_values->ptr;
      }
      // the distinctify table is keyed by (group-by..., argument): a tuple with a NULL in any of them is not inserted
      // (PackedPayloadHashTable.hpp:861-867)
      const std::unique_ptr<DeviceBuffer> not_null = NotNullFilter(block, nullable_sources, filter);
      const std::uint64_t *distinct_filter = not_null != nullptr ? static_cast<const std::uint64_t *>(not_null->ptr) : filter;
      CheckStatus(qsx_distinct_rows(static_cast<int>(d->attrs.size()), cols, types, n, distinct_filter, static_cast<std::int32_t *>(tids.ptr),
                                    static_cast<std::int64_t *>(count.ptr), ws.ptr, ws_bytes, CurrentStream()), "qsx_distinct_rows");
      Distinctify::Chunk chunk;
      chunk.rows = ReadCount(count.ptr);
      if (chunk.rows == 0) continue;
      for (std::size_t c = 0; c < d->attrs.size(); ++c) {
        chunk.cols.emplace_back(new DeviceBuffer(static_cast<std::size_t>(chunk.rows) * d->types[c].width + 16));
        CheckStatus(qsx_gather(d->types[c].width, cols[c], static_cast<const std::int32_t *>(tids.ptr), chunk.rows,
                               chunk.cols.back()->ptr, CurrentStream()), "qsx_gather");
      }
      CheckStatus(qsx_stream_synchronize(CurrentStream()), "qsx_stream_synchronize");
      std::lock_guard<std::mutex> lock(d->mutex);
      d->chunks.push_back(std::move(chunk));
    }
    qsx_device_free(selected);
  }
  if (state_ == nullptr) return;
  // the conjuncts the kernel does not evaluate: a TupleIdSequence like the one a SelectOperator computes, used as the filter
  struct OwnedBitmap {
    void *ptr = nullptr;
    ~OwnedBitmap() { if (ptr != nullptr) qsx_device_free(ptr); }
  } external_matches;
  if (!external_predicate_.conjuncts.empty() && n > 0) {
    std::int64_t matches = 0;
    external_matches.ptr = external_predicate_.getMatchesForBlock(block, &matches, lip_filter);
    lip_filter = static_cast<const std::uint64_t *>(external_matches.ptr);
  }
  // A block with compressed operand attributes whose values have not been materialised: aggregate on the codes.
  // (Key and predicate columns of the state take the value path here: stripe() decodes them once per block.)
  // null bitmaps of the nullable operand attributes (a block may hold none: loadBlock without bitmaps)
  const std::uint64_t *nulls[QSX_MAX_COLUMNS] = {};
  bool any_nulls = false;
  for (std::size_t i = 0; i < column_attr_.size(); ++i) {
    nulls[i] = config_.column_nullable[i] != 0 ? block.nullBitmap(column_attr_[i]) : nullptr;
    any_nulls = any_nulls || nulls[i] != nullptr;
  }
  int code_width[QSX_MAX_COLUMNS] = {};
  bool any_coded = false;
  for (std::size_t i = 0; i < column_attr_.size() && !any_nulls; ++i) {
    const CompressedAttribute *ca = block.compressedAttribute(column_attr_[i]);
    const int type = config_.column_type[i];
    if (ca != nullptr && type != kChar && !block.valuesMaterialized(column_attr_[i])) {
      code_width[i] = ca->code_width;
      any_coded = true;
    }
  }
  if (any_coded && n > 0) {
    bool use_coded = false;
    {
      std::lock_guard<std::mutex> lock(coded_mutex_);
      if (coded_state_ == nullptr && !coded_merged_) {
        coded_config_ = config_;
        for (std::size_t i = 0; i < column_attr_.size(); ++i) coded_config_.column_code_width[i] = code_width[i];
        externalizeCodedPredicate();
        CheckStatus(qsx_agg_state_create(&coded_config_, &coded_state_), "qsx_agg_state_create");
      }
      if (coded_state_ != nullptr && !coded_merged_) {
        use_coded = true;
        for (std::size_t i = 0; i < column_attr_.size(); ++i) use_coded = use_coded && coded_config_.column_code_width[i] == code_width[i];
      }
    }
    if (use_coded) {
      const void *cols[QSX_MAX_COLUMNS];
      const void *dicts[QSX_MAX_COLUMNS];
      std::int32_t dict_entries[QSX_MAX_COLUMNS] = {};
      for (std::size_t i = 0; i < column_attr_.size(); ++i) {
        const CompressedAttribute *ca = code_width[i] != 0 ? block.compressedAttribute(column_attr_[i]) : nullptr;
        cols[i] = ca != nullptr ? ca->codes : block.stripe(column_attr_[i]);
        dicts[i] = ca != nullptr && ca->kind == CompressedAttribute::kDictionary ? ca->dictionary : nullptr;
        dict_entries[i] = dicts[i] != nullptr ? static_cast<std::int32_t>(ca->num_codes) : 0;
      }
      OwnedBitmap predicate_matches;
      const std::uint64_t *coded_filter = lip_filter;
      if (coded_predicate_external_) {   // the state over code stripes leaves its predicate to the scans on codes (externalizeCodedPredicate)
        std::int64_t matches = 0;
        predicate_matches.ptr = spec_.predicate->getMatchesForBlock(block, &matches, lip_filter);
        coded_filter = static_cast<const std::uint64_t *>(predicate_matches.ptr);
      }
      CheckStatus(qsx_agg_update_coded_sized(coded_state_, cols, dicts, dict_entries, n, coded_filter, CurrentStream()),
                  "qsx_agg_update_coded_sized");
      ++coded_blocks_;
      CheckStatus(qsx_stream_synchronize(CurrentStream()), "qsx_stream_synchronize");
      return;
    }
  }
  const void *cols[QSX_MAX_COLUMNS];
  for (std::size_t i = 0; i < column_attr_.size(); ++i) cols[i] = block.stripe(column_attr_[i]);
  if (any_nulls) {
    CheckStatus(qsx_agg_update_nullable(state_, cols, nulls, n, lip_filter, CurrentStream()), "qsx_agg_update_nullable");
  } else {
    CheckStatus(qsx_agg_update(state_, cols, n, lip_filter, CurrentStream()), "qsx_agg_update");
  }
  CheckStatus(qsx_stream_synchronize(CurrentStream()), "qsx_stream_synchronize");
}

// The state over code stripes leaves a predicate that the kernel would have evaluated on DECODED values to the scans on codes
// instead (Predicate::getMatchesForBlock / RunPredicateMatches: the comparison rewritten on every block's own dictionary,
// storage/CompressedStoreUtil.cpp:51-140 — a code stripe is an eighth of the column) and takes the resulting TupleIdSequence as
// its filter.  The state then has no predicate of its own, which also lets its aggregates factor through the dictionary codes
// (csrc/agg_factored.hpp): TPC-H Q1's l_shipdate <= DATE over lineitem's compressed blocks.  Called with coded_mutex_ held.
void AggregationOperationState::externalizeCodedPredicate() {
  if (coded_config_.num_pred_terms > 0 && spec_.predicate != nullptr && external_predicate_.conjuncts.empty()) {
    coded_config_.num_pred_terms = 0;
    coded_predicate_external_ = true;
  }
}

void AggregationOperationState::aggregateBlocks(const std::vector<BlockReference> &blocks,
                                                const std::vector<const std::uint64_t *> &lip_filters) {
  std::vector<std::int64_t> rows;
  std::vector<const void *> cols;
  std::vector<const std::uint64_t *> filters;
  bool any_filter = false;
  std::vector<std::int64_t> coded_rows;
  std::vector<const void *> coded_cols, coded_dicts;
  std::vector<std::int32_t> coded_entries;   // every block's dictionary sizes (the reference builds a dictionary per block)
  std::vector<const std::uint64_t *> coded_filters;
  std::vector<BlockReference> coded_refs;
  bool any_coded_filter = false;
  const bool state_allows = state_ != nullptr && distinctify_.empty() && external_predicate_.conjuncts.empty();
  for (std::size_t i = 0; i < blocks.size(); ++i) {
    const StorageBlock &block = *blocks[i];
    const std::uint64_t *filter = i < lip_filters.size() ? lip_filters[i] : nullptr;
    bool in_run = state_allows && block.numTuples() > 0;
    int code_width[QSX_MAX_COLUMNS] = {};
    bool any_coded = false;
    for (std::size_t c = 0; c < column_attr_.size() && in_run; ++c) {
      // null bitmaps travel with single-block calls (qsx_agg_update_nullable)
      if (block.nullBitmap(column_attr_[c]) != nullptr) in_run = false;
      const CompressedAttribute *ca = block.compressedAttribute(column_attr_[c]);
      if (ca != nullptr && config_.column_type[c] != kChar && !block.valuesMaterialized(column_attr_[c])) {
        code_width[c] = ca->code_width;          // aggregated on its codes, like aggregateBlock does
        any_coded = true;
      }
    }
    if (in_run && any_coded) {
      // compressed blocks: one run through the state over code stripes (created by the first such block, which fixes the code
      // widths; blocks that compressed differently go with the run of plain stripes)
      bool use_coded = false;
      {
        std::lock_guard<std::mutex> lock(coded_mutex_);
        if (coded_state_ == nullptr && !coded_merged_) {
          coded_config_ = config_;
          for (std::size_t c = 0; c < column_attr_.size(); ++c) coded_config_.column_code_width[c] = code_width[c];
          externalizeCodedPredicate();
          CheckStatus(qsx_agg_state_create(&coded_config_, &coded_state_), "qsx_agg_state_create");
        }
        if (coded_state_ != nullptr && !coded_merged_) {
          use_coded = true;
          for (std::size_t c = 0; c < column_attr_.size(); ++c) use_coded = use_coded && coded_config_.column_code_width[c] == code_width[c];
        }
      }
      if (!use_coded) {
        // a block that compressed an operand differently than the block the coded state was created for (another code width, or
        // not at all): its values — stripe() decodes them once — join the run of plain stripes instead of a call of their own
        rows.push_back(block.numTuples());
        for (std::size_t c = 0; c < column_attr_.size(); ++c) cols.push_back(block.stripe(column_attr_[c]));
        filters.push_back(filter);
        any_filter = any_filter || filter != nullptr;
        continue;
      }
      coded_rows.push_back(block.numTuples());
      coded_refs.push_back(blocks[i]);
      for (std::size_t c = 0; c < column_attr_.size(); ++c) {
        const CompressedAttribute *ca = code_width[c] != 0 ? block.compressedAttribute(column_attr_[c]) : nullptr;
        coded_cols.push_back(ca != nullptr ? ca->codes : block.stripe(column_attr_[c]));
        coded_dicts.push_back(ca != nullptr && ca->kind == CompressedAttribute::kDictionary ? ca->dictionary : nullptr);
        coded_entries.push_back(coded_dicts.back() != nullptr ? static_cast<std::int32_t>(ca->num_codes) : 0);
      }
      coded_filters.push_back(filter);
      any_coded_filter = any_coded_filter || filter != nullptr;
      continue;
    }
    if (!in_run) {
      aggregateBlock(block, filter);
      continue;
    }
    rows.push_back(block.numTuples());
    for (std::size_t c = 0; c < column_attr_.size(); ++c) cols.push_back(block.stripe(column_attr_[c]));
    filters.push_back(filter);
    any_filter = any_filter || filter != nullptr;
  }
  RunMatches coded_matches;
  if (!coded_rows.empty() && coded_predicate_external_) {
    // the predicate over the run's code stripes: every term one launch, rewritten on every block's own codes
    // (RunPredicateMatches); a run the run forms do not cover goes block by block
    if (!RunPredicateCovers(*spec_.predicate, coded_refs)) {
      for (std::size_t b = 0; b < coded_refs.size(); ++b) aggregateBlock(*coded_refs[b], coded_filters[b]);
      coded_rows.clear();
    } else {
      RunPredicateMatches(*spec_.predicate, coded_refs, coded_rows, any_coded_filter ? coded_filters.data() : nullptr, &coded_matches);
      for (std::size_t b = 0; b < coded_refs.size(); ++b) coded_filters[b] = coded_matches.bitmaps[b];
      any_coded_filter = true;
    }
  }
  if (!coded_rows.empty()) {
    CheckStatus(qsx_agg_update_coded_blocks_sized(coded_state_, static_cast<int>(coded_rows.size()), coded_rows.data(), coded_cols.data(),
                                                  coded_dicts.data(), coded_entries.data(), any_coded_filter ? coded_filters.data() : nullptr,
                                                  CurrentStream()),
                "qsx_agg_update_coded_blocks_sized");
    coded_blocks_ += static_cast<int>(coded_rows.size());
    CheckStatus(qsx_stream_synchronize(CurrentStream()), "qsx_stream_synchronize");
  }
  if (rows.empty()) return;
  CheckStatus(qsx_agg_update_blocks(state_, static_cast<int>(rows.size()), rows.data(), cols.data(), any_filter ? filters.data() : nullptr,
                                    CurrentStream()), "qsx_agg_update_blocks");
  CheckStatus(qsx_stream_synchronize(CurrentStream()), "qsx_stream_synchronize");
}

// finalizeAggregate with DISTINCT aggregates: every distinctify table is reduced to its distinct tuples, which are
// aggregated once each into a table keyed like the final one (aggregateOnDistinctifyHashTableFor{Single,GroupBy},
// AggregationOperationState.cpp:652-670, 720-760); the per-aggregate results are then lined up on the group key.
void AggregationOperationState::finalizeWithDistinct(InsertDestination *dest) {
  const CatalogRelation &rel = *spec_.input_relation;
  const int nk = config_.num_keys;
  struct ResultSet {
    std::vector<std::unique_ptr<DeviceBuffer>> keys, vals;
    std::int64_t rows = 0;
    std::unique_ptr<DeviceBuffer> order;      // row numbers in ascending key order
  };
  auto finalize_into = [&](qsx_agg_state_t *state, const qsx_agg_config_t &cfg, const std::vector<Type> &val_types, ResultSet *out) {
    std::int64_t groups = 0;
    CheckStatus(qsx_agg_num_groups(state, &groups, CurrentStream()), "qsx_agg_num_groups");
    const std::int64_t cap = groups > 0 ? groups : 1;
    void *key_cols[QSX_MAX_KEYS];
    void *val_cols[QSX_MAX_AGGS];
    for (int k = 0; k < cfg.num_keys; ++k) {
      out->keys.emplace_back(new DeviceBuffer(static_cast<std::size_t>(cap) * cfg.column_width[cfg.key_column[k]] + 16));
      key_cols[k] = out->keys.back()->ptr;
    }
    for (int a = 0; a < cfg.num_aggs; ++a) {
      out->vals.emplace_back(new DeviceBuffer(static_cast<std::size_t>(cap) * val_types[a].width + 16));
      val_cols[a] = out->vals.back()->ptr;
    }
    DeviceBuffer rows(8);
    CheckStatus(qsx_agg_finalize(state, 0, 1, key_cols, val_cols, nullptr, cap, static_cast<std::int64_t *>(rows.ptr), CurrentStream()),
                "qsx_agg_finalize");
    out->rows = ReadCount(rows.ptr);
    if (out->rows == QSX_GROUPS_HASH_COLLISION) throw ExecutionError("qsx_agg_finalize: wide group-by key", QSX_ERR_HASH_COLLISION);
    if (cfg.num_keys > 0 && out->rows > 0) {   // ascending key order: the common order of all result sets
      const void *cols[QSX_MAX_KEYS];
      std::int32_t types[QSX_MAX_KEYS];
      for (int k = 0; k < cfg.num_keys; ++k) {
        cols[k] = out->keys[k]->ptr;
        types[k] = cfg.column_type[cfg.key_column[k]];
        if (types[k] == kChar && cfg.column_width[cfg.key_column[k]] != 1) {
          throw ExecutionError("DISTINCT aggregate: CHAR group-by keys wider than one byte", QSX_ERR_UNSUPPORTED);
        }
      }
      const std::size_t ws_bytes = qsx_sort_workspace_bytes(out->rows);
      DeviceBuffer ws(ws_bytes);
      out->order.reset(new DeviceBuffer(static_cast<std::size_t>(out->rows) * 4 + 16));
      CheckStatus(qsx_sort_permutation(cfg.num_keys, cols, types, nullptr, out->rows, static_cast<std::int32_t *>(out->order->ptr), ws.ptr,
                                       ws_bytes, CurrentStream()), "qsx_sort_permutation");
    }
  };

  // the non-DISTINCT aggregates
  ResultSet main;
  std::vector<Type> main_types;
  if (state_ != nullptr) {
    for (std::size_t a = 0; a < spec_.aggregates.size(); ++a) {
      if (main_agg_[a] < 0) continue;
      const AggregateSpec &ag = spec_.aggregates[a];
      main_types.push_back(AggResultType(ag.function, ag.argument_expression != nullptr ? Type::Double()   // expressions evaluate in DOUBLE
                                                      : ag.argument == kInvalidAttributeID ? Type::Long() : rel.getAttributeType(ag.argument)));
    }
    finalize_into(state_, config_, main_types, &main);
  }
  // one result set per DISTINCT aggregate
  std::vector<ResultSet> distinct(distinctify_.size());
  for (std::size_t i = 0; i < distinctify_.size(); ++i) {
    Distinctify &d = *distinctify_[i];
    const AggregateSpec &ag = spec_.aggregates[d.agg_index];
    const std::size_t ncols = d.attrs.size();
    std::int64_t total = 0;
    for (const auto &chunk : d.chunks) total += chunk.rows;
    // all block-level tuples side by side, then distinct over the whole input
    std::vector<std::unique_ptr<DeviceBuffer>> all, tuples;
    for (std::size_t c = 0; c < ncols; ++c) {
      all.emplace_back(new DeviceBuffer(static_cast<std::size_t>(total) * d.types[c].width + 16));
      char *at = static_cast<char *>(all.back()->ptr);
      for (const auto &chunk : d.chunks) {
        const std::size_t bytes = static_cast<std::size_t>(chunk.rows) * d.types[c].width;
        CheckStatus(qsx_copy_on_device(at, chunk.cols[c]->ptr, bytes, CurrentStream()), "qsx_copy_on_device");
        at += bytes;
      }
    }
    std::int64_t rows = 0;
    if (total > 0) {
      const void *cols[QSX_MAX_KEYS];
      std::int32_t types[QSX_MAX_KEYS];
      for (std::size_t c = 0; c < ncols; ++c) { cols[c] = all[c]->ptr; types[c] = d.types[c].id; }
      const std::size_t ws_bytes = qsx_sort_workspace_bytes(total);
      DeviceBuffer ws(ws_bytes), tids(static_cast<std::size_t>(total) * 4 + 16), count(8);
      CheckStatus(qsx_distinct_rows(static_cast<int>(ncols), cols, types, total, nullptr, static_cast<std::int32_t *>(tids.ptr),
                                    static_cast<std::int64_t *>(count.ptr), ws.ptr, ws_bytes, CurrentStream()), "qsx_distinct_rows");
      rows = ReadCount(count.ptr);
      for (std::size_t c = 0; c < ncols; ++c) {
        tuples.emplace_back(new DeviceBuffer(static_cast<std::size_t>(rows) * d.types[c].width + 16));
        CheckStatus(qsx_gather(d.types[c].width, cols[c], static_cast<const std::int32_t *>(tids.ptr), rows, tuples.back()->ptr,
                               CurrentStream()), "qsx_gather");
      }
    }
    // the aggregate over the distinct tuples, keyed like the final table
    qsx_agg_config_t cfg;
    std::memset(&cfg, 0, sizeof(cfg));
    cfg.strategy = config_.strategy;
    cfg.num_columns = static_cast<int>(ncols);
    for (std::size_t c = 0; c < ncols; ++c) {
      cfg.column_type[c] = d.types[c].id;
      cfg.column_width[c] = d.types[c].width;
    }
    cfg.num_keys = nk;
    for (int k = 0; k < nk; ++k) cfg.key_column[k] = k;
    cfg.num_aggs = 1;
    cfg.aggs[0].fn = AggFn(ag.function);             // COUNT(DISTINCT x) = COUNT(*) over the distinct tuples (no NULLs)
    if (ag.function != AggregationID::kCount) {
      cfg.aggs[0].arg.kind = QSX_OPD_COLUMN;
      cfg.aggs[0].arg.index = nk;
    }
    cfg.est_groups = config_.est_groups;
    cfg.num_entries = config_.num_entries;
    qsx_agg_state_t *state = nullptr;
    CheckStatus(qsx_agg_state_create(&cfg, &state), "qsx_agg_state_create");
    try {
      if (rows > 0) {
        const void *cols[QSX_MAX_KEYS];
        for (std::size_t c = 0; c < ncols; ++c) cols[c] = tuples[c]->ptr;
        CheckStatus(qsx_agg_update(state, cols, rows, nullptr, CurrentStream()), "qsx_agg_update");
      }
      finalize_into(state, cfg, {AggResultType(ag.function, d.types.back())}, &distinct[i]);
    } catch (...) {
      qsx_agg_state_destroy(state);
      throw;
    }
    CheckStatus(qsx_stream_synchronize(CurrentStream()), "qsx_stream_synchronize");
    qsx_agg_state_destroy(state);
  }
  // every result set holds the same groups (each group has at least one tuple in every table)
  const ResultSet &first = state_ != nullptr ? main : distinct.front();
  for (const ResultSet &r : distinct) {
    if (r.rows != first.rows) throw ExecutionError("DISTINCT aggregate: group sets differ", QSX_ERR_INVALID_ARGUMENT);
  }
  block_id id;
  BlockReference out = dest->getBlockForInsertion(first.rows > 0 ? first.rows : 1, &id);
  auto emit = [&](const ResultSet &r, const void *src, int width, attribute_id out_attr) {
    if (r.rows == 0) return;
    if (r.order != nullptr) {
      CheckStatus(qsx_gather(width, src, static_cast<const std::int32_t *>(r.order->ptr), r.rows, out->stripe(out_attr), CurrentStream()),
                  "qsx_gather");
    } else {
      CheckStatus(qsx_copy_on_device(out->stripe(out_attr), src, static_cast<std::size_t>(r.rows) * width, CurrentStream()),
                  "qsx_copy_on_device");
    }
  };
  for (int k = 0; k < nk; ++k) emit(first, first.keys[k]->ptr, config_.column_width[config_.key_column[k]], k);
  std::size_t next_distinct = 0;
  for (std::size_t a = 0; a < spec_.aggregates.size(); ++a) {
    const attribute_id out_attr = static_cast<attribute_id>(nk + a);
    if (main_agg_[a] >= 0) {
      emit(main, main.vals[main_agg_[a]]->ptr, main_types[main_agg_[a]].width, out_attr);
    } else {
      const ResultSet &r = distinct[next_distinct];
      emit(r, r.vals[0]->ptr, AggResultType(spec_.aggregates[a].function, distinctify_[next_distinct]->types.back()).width, out_attr);
      ++next_distinct;
    }
  }
  CheckStatus(qsx_stream_synchronize(CurrentStream()), "qsx_stream_synchronize");
  dest->returnBlock(id, first.rows);
}

void AggregationOperationState::buildExistenceMap(const StorageBlock &block, attribute_id build_attribute, const Type &type) {
  if (type.id != kInt && type.id != kLong) {   // LOG(FATAL) "Build attribute type not supported" (:203-206)
    throw ExecutionError("BuildAggregationExistenceMapOperator: build attribute must be INT or LONG", QSX_ERR_UNSUPPORTED);
  }
  CheckStatus(qsx_agg_mark_existence(state_, type.id, block.stripe(build_attribute), block.numTuples(), nullptr, CurrentStream()),
              "qsx_agg_mark_existence");
  CheckStatus(qsx_stream_synchronize(CurrentStream()), "qsx_stream_synchronize");
}

void AggregationOperationState::finalizeAggregate(std::size_t partition, std::size_t num_partitions, InsertDestination *dest) {
  {   // the state fed by compressed blocks joins the other one (same image layout: mergeFrom semantics)
    std::lock_guard<std::mutex> lock(coded_mutex_);
    if (coded_state_ != nullptr && !coded_merged_) {
      CheckStatus(qsx_agg_merge(state_, coded_state_, CurrentStream()), "qsx_agg_merge");
      CheckStatus(qsx_stream_synchronize(CurrentStream()), "qsx_stream_synchronize");
    }
    coded_merged_ = true;
  }
  if (!distinctify_.empty()) {
    // the distinctify tables are drained by one work order; the others of a partitioned finalize have nothing to emit
    if (partition == 0) finalizeWithDistinct(dest);
    return;
  }
  std::int64_t groups = 0;
  CheckStatus(qsx_agg_num_groups(state_, &groups, CurrentStream()), "qsx_agg_num_groups");
  block_id id;
  BlockReference out = dest->getBlockForInsertion(groups > 0 ? groups : 1, &id);
  void *key_cols[QSX_MAX_KEYS];
  void *val_cols[QSX_MAX_AGGS];
  std::uint8_t *null_cols[QSX_MAX_AGGS] = {};
  std::vector<std::unique_ptr<DeviceBuffer>> null_flags;
  for (int k = 0; k < config_.num_keys; ++k) key_cols[k] = out->stripe(k);
  for (int a = 0; a < config_.num_aggs; ++a) {
    val_cols[a] = out->stripe(config_.num_keys + a);
    // result types of SUM / AVG / MIN / MAX are nullable (AggregationHandleSum::getResultType ...->getNullableVersion()):
    // an output attribute declared nullable receives the NULL flags as its null bitmap
    if (out->nullBitmap(static_cast<attribute_id>(config_.num_keys + a)) != nullptr) {
      null_flags.emplace_back(new DeviceBuffer(static_cast<std::size_t>(out->capacity()) + 16));
      null_cols[a] = static_cast<std::uint8_t *>(null_flags.back()->ptr);
    }
  }
  DeviceBuffer rows(8);
  CheckStatus(qsx_agg_finalize(state_, static_cast<int>(partition), static_cast<int>(num_partitions), key_cols, val_cols,
                               null_flags.empty() ? nullptr : null_cols, out->capacity(), static_cast<std::int64_t *>(rows.ptr),
                               CurrentStream()),
              "qsx_agg_finalize");
  const std::int64_t written = ReadCount(rows.ptr);
  // a key wider than 8 bytes is grouped by its 64-bit hash and verified: two keys under one hash void the result
  if (written == QSX_GROUPS_HASH_COLLISION) throw ExecutionError("qsx_agg_finalize: wide group-by key", QSX_ERR_HASH_COLLISION);
  if (written > out->capacity()) throw ExecutionError("qsx_agg_finalize: more groups than the output block holds", QSX_ERR_CAPACITY);
  for (int a = 0; a < config_.num_aggs && written > 0; ++a) {
    if (null_cols[a] == nullptr) continue;
    // byte flags -> TupleIdSequence-ordered bitmap: a scan of 1-byte "codes" for flag >= 1
    CheckStatus(qsx_select_codes(1, null_cols[a], written, QSX_CODE_GE, 1, 0, nullptr,
                                 out->nullBitmap(static_cast<attribute_id>(config_.num_keys + a)), static_cast<std::int64_t *>(rows.ptr),
                                 CurrentStream()), "qsx_select_codes(null flags)");
  }
  CheckStatus(qsx_stream_synchronize(CurrentStream()), "qsx_stream_synchronize");
  dest->returnBlock(id, written);
}

// ---------------------------------------------------------------------------
// QueryContext
// ---------------------------------------------------------------------------
QueryContext::~QueryContext() {
  for (auto &parts : join_tables_) {
    for (qsx_join_table_t *t : parts) qsx_join_table_destroy(t);
  }
  for (qsx_lip_filter_t *f : lip_filters_) qsx_lip_filter_destroy(f);
}
QueryContext::lip_filter_id QueryContext::addLIPFilter(qsx_lip_kind_t kind, std::int64_t cardinality, std::int64_t min_value,
                                                       bool is_anti) {
  qsx_lip_filter_t *f = nullptr;
  CheckStatus(qsx_lip_filter_create(kind, cardinality, min_value, is_anti ? 1 : 0, &f), "qsx_lip_filter_create");
  lip_filters_.push_back(f);
  return static_cast<lip_filter_id>(lip_filters_.size() - 1);
}
void QueryContext::destroyLIPFilter(lip_filter_id id) {
  qsx_lip_filter_destroy(lip_filters_.at(id));
  lip_filters_.at(id) = nullptr;
}
QueryContext::lip_deployment_id QueryContext::addLIPDeployment(LIPFilterDeployment deployment) {
  lip_deployments_.push_back(std::move(deployment));
  return static_cast<lip_deployment_id>(lip_deployments_.size() - 1);
}

// ---------------------------------------------------------------------------
// LIP filter builder / prober
// ---------------------------------------------------------------------------
LIPFilterBuilder::LIPFilterBuilder(const QueryContext::LIPFilterDeployment &deployment, const QueryContext &query_context) {
  for (const auto &e : deployment.build_entries) entries_.emplace_back(query_context.getLIPFilterMutable(e.lip_filter), e.attribute);
}
void LIPFilterBuilder::insertValueAccessor(const StorageBlock &block, const std::uint64_t *filter) const {
  for (const auto &e : entries_) {
    // a NULL is never inserted (SingleIdentityHashFilter.hpp:115-126, BitVectorExactFilter.hpp:115-128)
    std::unique_ptr<DeviceBuffer> not_null = NotNullFilter(block, {e.second}, filter);
    CheckStatus(qsx_lip_build(e.first, block.getRelation().getAttributeType(e.second).id, block.stripe(e.second),
                              block.numTuples(), not_null != nullptr ? static_cast<const std::uint64_t *>(not_null->ptr) : filter,
                              CurrentStream()), "qsx_lip_build");
    if (not_null != nullptr) CheckStatus(qsx_stream_synchronize(CurrentStream()), "qsx_stream_synchronize");
  }
}
bool LIPFilterBuilder::coversBlocks(const std::vector<BlockReference> &blocks) const {
  for (const auto &e : entries_) {
    for (const BlockReference &b : blocks) {
      if (b->nullBitmap(e.second) != nullptr) return false;   // (a compressed attribute is read as it lies: RunJoinKeys)
    }
  }
  return true;
}
bool LIPFilterBuilder::insertBlocks(const std::vector<BlockReference> &blocks, qsx_join_table_t *built_table, attribute_id table_key) const {
  if (!coversBlocks(blocks)) return false;
  std::vector<std::int64_t> rows;
  std::int64_t total_rows = 0;
  for (const BlockReference &b : blocks) {
    rows.push_back(b->numTuples());
    total_rows += b->numTuples();
  }
  for (const auto &e : entries_) {
    if (built_table != nullptr && e.second == table_key) {
      // the table holds these keys already: its head words are the filter's bits (QSX_ERR_UNSUPPORTED: not that kind of table
      // or filter, or the run is too short for a pass over the key range to pay — the keys one by one then)
      const int rc = qsx_lip_build_from_join_table(e.first, built_table, total_rows, CurrentStream());
      if (rc == QSX_OK) continue;
      if (rc != QSX_ERR_UNSUPPORTED) CheckStatus(rc, "qsx_lip_build_from_join_table");
    }
    // (INT / LONG attributes — the key types of a LIP filter — that a block holds compressed are read as they lie)
    const RunJoinKeys keys(blocks, {e.second}, rows);
    CheckStatus(qsx_lip_build_blocks_coded(e.first, blocks.front()->getRelation().getAttributeType(e.second).id,
                                           static_cast<std::int64_t>(blocks.size()), rows.data(), keys.ptr.data(), keys.coding(), nullptr,
                                           CurrentStream()),
                "qsx_lip_build_blocks");
  }
  return true;
}
LIPFilterAdaptiveProber::LIPFilterAdaptiveProber(const QueryContext::LIPFilterDeployment &deployment,
                                                 const QueryContext &query_context) {
  for (const auto &e : deployment.probe_entries) entries_.emplace_back(query_context.getLIPFilterMutable(e.lip_filter), e.attribute);
}
void *LIPFilterAdaptiveProber::filterValueAccessor(const StorageBlock &block, const std::uint64_t *filter,
                                                   std::int64_t *num_hits) const {
  const std::int64_t n = block.numTuples();
  const std::size_t bytes = static_cast<std::size_t>((n + 63) / 64) * 8 + 8;
  void *current = nullptr, *next = nullptr;
  CheckStatus(qsx_device_alloc(bytes, &current), "qsx_device_alloc(bitmap)");
  CheckStatus(qsx_device_alloc(bytes, &next), "qsx_device_alloc(bitmap)");
  DeviceBuffer count(8);
  const std::uint64_t *in = filter;
  for (const auto &e : entries_) {
    // a NULL never passes a filter (SingleIdentityHashFilter.hpp:133-152)
    std::unique_ptr<DeviceBuffer> not_null = NotNullFilter(block, {e.second}, in);
    if (not_null != nullptr) in = static_cast<const std::uint64_t *>(not_null->ptr);
    CheckStatus(qsx_lip_probe(e.first, block.getRelation().getAttributeType(e.second).id, block.stripe(e.second), n, in,
                              static_cast<std::uint64_t *>(next), static_cast<std::int64_t *>(count.ptr), CurrentStream()),
                "qsx_lip_probe");
    if (not_null != nullptr) CheckStatus(qsx_stream_synchronize(CurrentStream()), "qsx_stream_synchronize");
    std::swap(current, next);
    in = static_cast<const std::uint64_t *>(current);
  }
  if (num_hits != nullptr) *num_hits = ReadCount(count.ptr);
  qsx_device_free(next);
  return current;
}
bool LIPFilterAdaptiveProber::filterBlocks(const std::vector<BlockReference> &blocks, void **storage,
                                           std::vector<const std::uint64_t *> *bitmaps, std::int64_t *num_hits,
                                           const std::uint64_t *const *in_bitmaps) const {
  *storage = nullptr;
  for (const auto &e : entries_) {
    for (const BlockReference &b : blocks) {
      if (b->nullBitmap(e.second) != nullptr) return false;   // (a compressed attribute is read as it lies: RunJoinKeys)
    }
  }
  const std::size_t nb = blocks.size();
  std::vector<std::int64_t> rows;
  std::size_t words = 0;
  for (const BlockReference &b : blocks) {
    rows.push_back(b->numTuples());
    words += static_cast<std::size_t>((b->numTuples() + 63) / 64) + 1;
  }
  // two sets of per-block bitmaps in one allocation; the filters ping-pong between them and the result ends in the first
  CheckStatus(qsx_device_alloc(2 * words * 8 + 8, storage), "qsx_device_alloc(bitmaps)");
  std::vector<std::uint64_t *> cur(nb), nxt(nb);
  std::size_t at = 0;
  for (std::size_t b = 0; b < nb; ++b) {
    cur[b] = static_cast<std::uint64_t *>(*storage) + at;
    nxt[b] = static_cast<std::uint64_t *>(*storage) + words + at;
    at += static_cast<std::size_t>((rows[b] + 63) / 64) + 1;
  }
  DeviceBuffer count(8);
  bool first = true;
  for (const auto &e : entries_) {
    const RunJoinKeys keys(blocks, {e.second}, rows);   // (a compressed INT / LONG attribute: its code stripes, as they lie)
    CheckStatus(qsx_lip_probe_blocks_coded(e.first, blocks.front()->getRelation().getAttributeType(e.second).id, static_cast<std::int64_t>(nb),
                                           rows.data(), keys.ptr.data(), keys.coding(),
                                           first ? in_bitmaps : reinterpret_cast<const std::uint64_t *const *>(cur.data()), nxt.data(),
                                           static_cast<std::int64_t *>(count.ptr), CurrentStream()), "qsx_lip_probe_blocks");
    std::swap(cur, nxt);
    first = false;
  }
  if (num_hits != nullptr) {
    *num_hits = 0;
    if (first) {
      for (std::int64_t r : rows) *num_hits += r;
    } else {
      *num_hits = ReadCount(count.ptr);
    }
  }
  if (first && in_bitmaps != nullptr) {   // no filter attached: what came in
    for (std::size_t b = 0; b < nb; ++b) {
      if (rows[b] > 0) CheckStatus(qsx_bitmap_combine(0, in_bitmaps[b], in_bitmaps[b], rows[b], cur[b], CurrentStream()), "qsx_bitmap_combine");
    }
  } else if (first) {   // no filter attached: every tuple
    for (std::size_t b = 0; b < nb; ++b) {
      CheckStatus(qsx_memset_device(nxt[b], 0xFF, static_cast<std::size_t>((rows[b] + 63) / 64) * 8, CurrentStream()), "qsx_memset_device");
      if (rows[b] > 0) {
        CheckStatus(qsx_bitmap_combine(0, nxt[b], nxt[b], rows[b], cur[b], CurrentStream()), "qsx_bitmap_combine");   // clears the tail bits
      }
    }
  }
  bitmaps->assign(cur.begin(), cur.end());
  return true;
}
LIPFilterBuilder *CreateLIPFilterBuilderHelper(QueryContext::lip_deployment_id id, const QueryContext *query_context) {
  const QueryContext::LIPFilterDeployment *d = query_context->getLIPDeployment(id);
  return d == nullptr || d->build_entries.empty() ? nullptr : new LIPFilterBuilder(*d, *query_context);
}
LIPFilterAdaptiveProber *CreateLIPFilterAdaptiveProberHelper(QueryContext::lip_deployment_id id, const QueryContext *query_context) {
  const QueryContext::LIPFilterDeployment *d = query_context->getLIPDeployment(id);
  return d == nullptr || d->probe_entries.empty() ? nullptr : new LIPFilterAdaptiveProber(*d, *query_context);
}
QueryContext::predicate_id QueryContext::addPredicate(Predicate p) {
  predicates_.push_back(std::move(p));
  return static_cast<predicate_id>(predicates_.size() - 1);
}
QueryContext::scalar_group_id QueryContext::addScalarGroup(std::vector<attribute_id> attrs) {
  scalar_groups_.push_back(std::move(attrs));
  return static_cast<scalar_group_id>(scalar_groups_.size() - 1);
}
QueryContext::join_hash_table_id QueryContext::addJoinHashTable(TypeID key_type, std::int64_t estimated_entries,
                                                                std::size_t num_partitions,
                                                                const ExactKeyRange *exact_key_range) {
  std::vector<qsx_join_table_t *> parts(num_partitions, nullptr);
  if (key_type == kChar) key_type = kLong;   // a CHAR(n <= 8) key travels as the LONG qsx_join_key_pack_char makes of it
  for (std::size_t p = 0; p < num_partitions; ++p) {
    if (exact_key_range != nullptr) {
      // every partition addresses the whole range: the single-node partition function is not a stride of the key
      CheckStatus(qsx_join_table_create_dense(key_type, exact_key_range->min_value, exact_key_range->max_value, 1,
                                              estimated_entries, &parts[p]),
                  "qsx_join_table_create_dense");
    } else {
      CheckStatus(qsx_join_table_create(key_type, estimated_entries, &parts[p]), "qsx_join_table_create");
    }
  }
  join_tables_.push_back(std::move(parts));
  return static_cast<join_hash_table_id>(join_tables_.size() - 1);
}
void QueryContext::destroyJoinHashTable(join_hash_table_id id, partition_id part) {
  // (DestroyHashOperator runs behind the last HashJoin work order, and every work order waits for its stream before it returns:
  // nothing queued uses the table — no wait for the other operators' kernels, qsx_join_table_release)
  qsx_join_table_release(join_tables_.at(id).at(part));
  join_tables_.at(id).at(part) = nullptr;
}
QueryContext::aggregation_state_id QueryContext::addAggregationState(const AggregationStateSpec &spec,
                                                                     std::size_t num_partitions) {
  std::vector<std::unique_ptr<AggregationOperationState>> parts;
  for (std::size_t p = 0; p < num_partitions; ++p) parts.emplace_back(new AggregationOperationState(spec));
  agg_states_.push_back(std::move(parts));
  return static_cast<aggregation_state_id>(agg_states_.size() - 1);
}
QueryContext::insert_destination_id QueryContext::addInsertDestination(CatalogRelation *relation,
                                                                       StorageManager *storage_manager) {
  destinations_.emplace_back(new InsertDestination(relation, storage_manager));
  return static_cast<insert_destination_id>(destinations_.size() - 1);
}
QueryContext::insert_destination_id QueryContext::addPartitionAwareInsertDestination(CatalogRelation *relation,
                                                                                     StorageManager *storage_manager) {
  if (!relation->hasPartitionScheme()) {
    throw ExecutionError("addPartitionAwareInsertDestination: the output relation has no partition scheme", QSX_ERR_INVALID_ARGUMENT);
  }
  destinations_.emplace_back(new InsertDestination(relation, storage_manager, relation->getNumPartitions(), relation->getPartitionAttribute()));
  return static_cast<insert_destination_id>(destinations_.size() - 1);
}

// has_repartition of an operator (RelationalOperator.hpp:311-320) and the kind of its InsertDestination must agree: a
// repartitioning operator whose destination would drop the partition scheme is a plan error, never silently accepted.
void CheckRepartition(const char *op, bool has_repartition, const InsertDestination *dest) {
  if (dest == nullptr || has_repartition == dest->isPartitionAware()) return;
  throw ExecutionError(std::string(op) + (has_repartition ? ": has_repartition needs a PartitionAwareInsertDestination (QueryContext::addPartitionAwareInsertDestination)"
                                                          : ": a PartitionAwareInsertDestination needs has_repartition = true"),
                       QSX_ERR_INVALID_ARGUMENT);
}

// ---------------------------------------------------------------------------
// WorkOrdersContainer
// ---------------------------------------------------------------------------
void WorkOrdersContainer::addNormalWorkOrder(WorkOrder *workorder, std::size_t operator_index) {
  std::lock_guard<std::mutex> lock(mutex_);
  queues_.at(operator_index).emplace_back(workorder);
}
bool WorkOrdersContainer::hasNormalWorkOrder(std::size_t operator_index) const {
  std::lock_guard<std::mutex> lock(mutex_);
  return !queues_.at(operator_index).empty();
}
WorkOrder *WorkOrdersContainer::getNormalWorkOrder(std::size_t operator_index) {
  std::lock_guard<std::mutex> lock(mutex_);
  auto &q = queues_.at(operator_index);
  if (q.empty()) return nullptr;
  WorkOrder *wo = q.front().release();
  q.pop_front();
  return wo;
}
std::size_t WorkOrdersContainer::getNumNormalWorkOrders(std::size_t operator_index) const {
  std::lock_guard<std::mutex> lock(mutex_);
  return queues_.at(operator_index).size();
}


}  // namespace quickstep
