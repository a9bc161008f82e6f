// hash_join_operator.cpp — HashJoinOperator and its inner / semi / anti / outer work orders, DestroyHashOperator (see quickstep_gpu.hpp; what the files share: quickstep_gpu_internal.hpp)
#include "quickstep_gpu_internal.hpp"

namespace quickstep {

// ---------------------------------------------------------------------------
// HashJoin
// ---------------------------------------------------------------------------
HashJoinOperator::HashJoinOperator(std::size_t query_id, const CatalogRelation &build_relation,
                                   const CatalogRelation &probe_relation, bool probe_relation_is_stored,
                                   const std::vector<attribute_id> &join_key_attributes, bool, std::size_t num_partitions,
                                   bool has_repartition, const CatalogRelation &output_relation,
                                   QueryContext::insert_destination_id output_destination_index,
                                   QueryContext::join_hash_table_id hash_table_index,
                                   QueryContext::predicate_id residual_predicate_index,
                                   QueryContext::scalar_group_id selection_index,
                                   const std::vector<bool> *is_selection_on_build, JoinType join_type)
    : RelationalOperator(query_id, num_partitions, has_repartition), build_relation_(build_relation),
      probe_relation_(probe_relation), probe_relation_is_stored_(probe_relation_is_stored),
      join_key_attributes_(join_key_attributes), output_relation_(output_relation),
      output_destination_index_(output_destination_index), hash_table_index_(hash_table_index),
      residual_predicate_index_(residual_predicate_index), selection_index_(selection_index), join_type_(join_type),
      probe_(num_partitions) {
  if (join_key_attributes.empty() || join_key_attributes.size() > QSX_MAX_KEYS) {
    throw ExecutionError("HashJoinOperator: 1 to 4 INT/LONG join key attributes are on the GPU path", QSX_ERR_UNSUPPORTED);
  }
  if (join_type == JoinType::kLeftOuterJoin && residual_predicate_index != QueryContext::kInvalidPredicateId) {
    // as in the reference: HashOuterJoinWorkOrder takes no residual predicate (HashJoinOperator.hpp:571-640)
    throw ExecutionError("HashJoinOperator: outer joins take no residual predicate", QSX_ERR_UNSUPPORTED);
  }
  if (is_selection_on_build != nullptr) is_selection_on_build_ = *is_selection_on_build;
  if (probe_relation_is_stored) {
    if (num_partitions > 1 && probe_relation.getNumPartitions() != num_partitions) {
      throw ExecutionError("HashJoinOperator: num_partitions differs from the probe relation's partition scheme", QSX_ERR_INVALID_ARGUMENT);
    }
    for (partition_id part = 0; part < num_partitions; ++part) {
      probe_.ids[part] = num_partitions > 1 ? probe_relation.getBlocksInPartition(part) : probe_relation.getBlocksSnapshot();
    }
  }
}

bool HashJoinOperator::getAllWorkOrders(WorkOrdersContainer *container, QueryContext *query_context,
                                        StorageManager *storage_manager, const tmb::client_id, tmb::MessageBus *) {
  const std::vector<attribute_id> &selection = query_context->getScalarGroup(selection_index_);
  if (is_selection_on_build_.empty()) is_selection_on_build_.assign(selection.size(), false);
  InsertDestination *dest = query_context->getInsertDestination(output_destination_index_);
  CheckRepartition("HashJoinOperator", has_repartition_, dest);
  std::lock_guard<std::mutex> lock(mutex_);
  if (!started_) {
    // the build operator is a blocking dependency: it has published its key attributes by now
    build_key_attributes_ = query_context->getJoinHashTableBuildKeyAttributes(hash_table_index_);
    if (build_key_attributes_.size() != join_key_attributes_.size()) {
      throw ExecutionError("HashJoinOperator: build and probe sides have different numbers of key attributes", QSX_ERR_INVALID_ARGUMENT);
    }
    started_ = true;
  }
  for (partition_id part = 0; part < num_partitions_; ++part) {   // HashJoinOperator.cpp:220-250
    qsx_join_table_t *table = query_context->getJoinHashTable(hash_table_index_, part);
    while (probe_.generated[part] < probe_.ids[part].size()) {
      // every probe block that has arrived, in runs of blocks_per_work_order_ (1: one work order per block)
      const std::size_t take = std::min(blocks_per_work_order_, probe_.ids[part].size() - probe_.generated[part]);
      HashInnerJoinWorkOrder *order =
          new HashInnerJoinWorkOrder(query_id_, build_relation_, probe_relation_, join_key_attributes_, build_key_attributes_,
                                     probe_.ids[part][probe_.generated[part]],
                                     query_context->getPredicate(residual_predicate_index_), selection, is_selection_on_build_,
                                     join_type_, table, dest, storage_manager, part,
                                     CreateLIPFilterAdaptiveProberHelper(lip_deployment_index_, query_context));
      // (an operator told to work on runs gives a lone block the run form too — the probe that writes the output relation
      // itself, the coded key stripes — e.g. the one block a PartitionExchangeOperator delivers per round)
      if (take > 1 || blocks_per_work_order_ > 1) {
        order->setRun(std::vector<block_id>(probe_.ids[part].begin() + static_cast<std::ptrdiff_t>(probe_.generated[part]),
                                            probe_.ids[part].begin() + static_cast<std::ptrdiff_t>(probe_.generated[part] + take)));
      }
      container->addNormalWorkOrder(order, op_index_);
      probe_.generated[part] += take;
    }
  }
  return probe_relation_is_stored_ || done_feeding_input_relation_;
}

namespace {
// The build relation as gather segments (one per build block; the reference loops over build
// blocks instead, HashJoinOperator.cpp:494-540).
struct BuildSegments {
  std::vector<BlockReference> refs;
  std::vector<std::int64_t> first_rows;
  BuildSegments(const CatalogRelation &build_relation, StorageManager *storage_manager) {
    for (block_id b : build_relation.getBlocksSnapshot()) refs.push_back(storage_manager->getBlock(b));
    std::sort(refs.begin(), refs.end(),
              [](const BlockReference &a, const BlockReference &b) { return a->firstRow() < b->firstRow(); });
    for (const BlockReference &b : refs) first_rows.push_back(b->firstRow());
  }
  void gather(attribute_id attr, int width, const void *build_tids, std::int64_t n, void *dst) const {
    std::vector<const void *> segs;
    for (const BlockReference &b : refs) segs.push_back(b->stripe(attr));
    CheckStatus(qsx_gather_segmented(width, static_cast<int>(segs.size()), segs.data(), first_rows.data(),
                                     static_cast<const std::int32_t *>(build_tids), n, dst, CurrentStream()),
                "qsx_gather_segmented");
  }
  // null bits of a build attribute for the joined pairs (negative tid = outer-join padding = NULL)
  void gatherNulls(attribute_id attr, const void *build_tids, std::int64_t n, std::uint64_t *dst) const {
    std::vector<const std::uint64_t *> segs;
    for (const BlockReference &b : refs) segs.push_back(b->nullBitmap(attr));
    CheckStatus(qsx_bitmap_gather_segmented(static_cast<int>(segs.size()), segs.data(), first_rows.data(),
                                            static_cast<const std::int32_t *>(build_tids), n, dst, CurrentStream()),
                "qsx_bitmap_gather_segmented");
  }
};

// Joined pairs of one probe block, on device.
struct JoinedPairs {
  std::unique_ptr<DeviceBuffer> probe_tids, build_tids;
  std::int64_t count = 0;
};

// Keep the pairs set in `bitmap` (order preserving).
void CompactPairs(JoinedPairs *pairs, const void *bitmap) {
  const std::int64_t n = pairs->count;
  std::unique_ptr<DeviceBuffer> p(new DeviceBuffer(static_cast<std::size_t>(n) * 4 + 8)), b(new DeviceBuffer(static_cast<std::size_t>(n) * 4 + 8));
  const void *src[2] = {pairs->probe_tids->ptr, pairs->build_tids->ptr};
  void *dst[2] = {p->ptr, b->ptr};
  const std::int32_t widths[2] = {4, 4};
  const std::size_t ws_bytes = qsx_compact_workspace_bytes(n);
  DeviceBuffer ws(ws_bytes), count(8);
  CheckStatus(qsx_compact_gather(2, src, widths, static_cast<const std::uint64_t *>(bitmap), n, dst,
                                 static_cast<std::int64_t *>(count.ptr), ws.ptr, ws_bytes, CurrentStream()),
              "qsx_compact_gather(pairs)");
  pairs->count = ReadCount(count.ptr);
  pairs->probe_tids = std::move(p);
  pairs->build_tids = std::move(b);
}
}  // namespace

bool HashInnerJoinWorkOrder::prefersExclusiveDevice() const {
  constexpr std::int64_t kRows = 4ll << 20;   // (below that a probe is over before the other streams have drained)
  std::int64_t rows = 0;
  for (block_id id : run_block_ids_) {
    rows += storage_manager_->getBlock(id)->numTuples();
    if (rows >= kRows) return true;
  }
  return false;
}

void HashInnerJoinWorkOrder::execute() {
  if (run_block_ids_.empty()) {
    executeBlock(block_id_);
    return;
  }
  if (executeRun()) return;
  for (block_id id : run_block_ids_) executeBlock(id);
}

// A run of probe blocks as one unit: one counting and one pair-emitting launch over all blocks (probe tuple ids are
// run-global row numbers), then every output attribute is one segmented gather — the probe side from the run's own
// stripes, the build side from the build relation's blocks — into ONE output block.
bool HashInnerJoinWorkOrder::executeRun() {
  using JoinType = HashJoinOperator::JoinType;
  const bool existence = join_type_ == JoinType::kLeftSemiJoin || join_type_ == JoinType::kLeftAntiJoin;
  const bool outer = join_type_ == JoinType::kLeftOuterJoin;
  if (join_type_ != JoinType::kInnerJoin && !existence && !outer) return false;
  if (outer && residual_predicate_ != nullptr) return false;   // (HashOuterJoinWorkOrder takes none either)
  // semi / anti with a residual predicate: the pairs of the run, the residual on them, then the probe tuples that kept (semi)
  // or never had (anti) a pair — HashSemiJoinWorkOrder / HashAntiJoinWorkOrder::executeWithResidualPredicate (:680-793, :880-1000)
  int key_bits = 0;
  for (attribute_id a : join_key_attributes_) {   // (a CHAR(n <= 8) component travels as a LONG)
    key_bits += probe_relation_.getAttributeType(a).id == kChar ? 64 : probe_relation_.getAttributeType(a).width * 8;
  }
  const bool hashed_key = join_key_attributes_.size() > 1 && key_bits > 64;   // the fold is a hash: pairs need their components compared
  // ... and so does a semi / anti join over a hashed composite key: an existence probe would count tuples whose key merely
  // shares the fold with a build key (compositeKeyCollisionCheck, SeparateChainingHashTable.hpp:1046)
  const bool existence_by_pairs = existence && (residual_predicate_ != nullptr || hashed_key);
  std::vector<BlockReference> blocks;
  std::vector<std::int64_t> rows, first_rows;
  std::int64_t total_rows = 0;
  bool any_null_key = false;   // some block holds NULLs in a key attribute: those tuples are not looked up
  for (block_id id : run_block_ids_) {
    blocks.push_back(storage_manager_->getBlock(id));
    const StorageBlock &b = *blocks.back();
    for (attribute_id a : join_key_attributes_) any_null_key = any_null_key || b.nullBitmap(a) != nullptr;   // (compressed key: stripe() decodes once)
    rows.push_back(b.numTuples());
    first_rows.push_back(total_rows);
    total_rows += b.numTuples();
    // (the pairs of a semi / anti join end up as ONE bitmap over the run's tuple ids: every block starts at a word boundary,
    // so that its part of the bitmap is the bitmap of the block)
    if (existence_by_pairs || outer) total_rows = (total_rows + 63) / 64 * 64;
  }
  if (total_rows > INT32_MAX || blocks.size() > 16384) return false;
  const std::int64_t nb = static_cast<std::int64_t>(blocks.size());
  // The terms evaluated on the pair list (component equalities of a hashed composite key, residual conjuncts): the run form
  // gathers their operands with widths of 1 / 2 / 4 / 8 bytes and compares numeric types — a nullable or CHAR(n) operand
  // sends the whole work order to the block-by-block form BEFORE any device work is done for the run.
  {
    auto operand_ok = [&](attribute_id attr, bool on_build, bool against_literal) {
      const Type &t = (on_build ? build_relation_ : probe_relation_).getAttributeType(attr);
      // (a CHAR(n) attribute against a literal is compared by qsx_select_cmp_char; attribute against attribute is numeric)
      return !t.nullable && t.id != kVarChar && (t.id != kChar || against_literal);
    };
    if (hashed_key) {
      for (std::size_t k = 0; k < join_key_attributes_.size(); ++k) {
        if (!operand_ok(join_key_attributes_[k], false, false) || !operand_ok(build_key_attributes_[k], true, false)) return false;
      }
    }
    if (residual_predicate_ != nullptr) {
      for (const ComparisonPredicate &term : residual_predicate_->conjuncts) {
        const bool literal = term.rhs_attribute == kInvalidAttributeID;
        if (!operand_ok(term.attribute, term.on_build_side, literal)) return false;
        if (!literal && !operand_ok(term.rhs_attribute, term.rhs_on_build_side, false)) return false;
      }
    }
  }
  const RunJoinKeys run_keys(blocks, join_key_attributes_, rows);   // composite key: one launch packs the run's keys
  const std::vector<const void *> &keys = run_keys.ptr;
  // existence_map of the LIPFilterAdaptiveProber (:462-470): the probe tuples this work order looks up
  struct OwnedStorage {
    void *ptr = nullptr;
    ~OwnedStorage() { qsx_device_free(ptr); }
  } lip_storage;
  std::vector<const std::uint64_t *> lip_bitmaps;
  if (lip_filter_adaptive_prober_ != nullptr && !lip_filter_adaptive_prober_->filterBlocks(blocks, &lip_storage.ptr, &lip_bitmaps)) return false;
  // `about` = the tuples this work order is about (the LIP filter's survivors, or all); `lookup` = those of them that are
  // looked up: a probe tuple with a NULL key component matches nothing (check_for_null_keys, HashTable.hpp:2158-2160,
  // 1855-1865) — out of an inner / semi join, NULL-padded by an outer join, KEPT by an anti join.  Blocks without NULLs in
  // their key attributes (no bitmap) cost nothing here.
  const std::uint64_t *const *about = lip_bitmaps.empty() ? nullptr : lip_bitmaps.data();
  std::vector<std::unique_ptr<DeviceBuffer>> not_null_storage;
  std::vector<const std::uint64_t *> lookup_bitmaps;
  if (any_null_key) {
    for (std::size_t b = 0; b < blocks.size(); ++b) {
      const std::uint64_t *in = about != nullptr ? about[b] : nullptr;
      std::unique_ptr<DeviceBuffer> not_null = NotNullFilter(*blocks[b], join_key_attributes_, in);
      lookup_bitmaps.push_back(not_null != nullptr ? static_cast<const std::uint64_t *>(not_null->ptr) : in);
      not_null_storage.push_back(std::move(not_null));
    }
  }
  const std::uint64_t *const *lookup = any_null_key ? lookup_bitmaps.data() : about;
  // nullable probe-side output attributes: their null bits follow the output tuples (qsx_bitmap_gather_segmented over the
  // blocks' null bitmaps, by the same tuple ids that gather the values)
  auto gather_probe_nulls = [&](std::size_t i, const void *tids, std::int64_t n, BlockReference &out, std::int64_t out_first_bit) {
    const Type &t = probe_relation_.getAttributeType(selection_[i]);
    if (!t.nullable || n == 0) return;
    bool any = false;
    std::vector<const std::uint64_t *> segs(blocks.size());
    for (std::size_t b = 0; b < blocks.size(); ++b) {
      segs[b] = blocks[b]->nullBitmap(selection_[i]);
      any = any || segs[b] != nullptr;
    }
    std::uint64_t *nulls = out->nullBitmap(static_cast<attribute_id>(i));
    if (nulls == nullptr) throw ExecutionError("join output of a nullable attribute must be nullable", QSX_ERR_INVALID_ARGUMENT);
    if (!any) return;   // (the output block's bitmaps start zeroed)
    if ((out_first_bit & 63) != 0) throw ExecutionError("join output null bits: unaligned tail", QSX_ERR_UNSUPPORTED);
    // empty blocks share their first row with the next one: the gather wants strictly increasing segment starts
    std::vector<const std::uint64_t *> used_segs;
    std::vector<std::int64_t> used_first;
    for (std::size_t b = 0; b < blocks.size(); ++b) {
      if (rows[b] == 0) continue;
      used_segs.push_back(segs[b]);
      used_first.push_back(first_rows[b]);
    }
    CheckStatus(qsx_bitmap_gather_segmented(static_cast<int>(used_segs.size()), used_segs.data(), used_first.data(), static_cast<const std::int32_t *>(tids), n,
                                            nulls + out_first_bit / 64, CurrentStream()), "qsx_bitmap_gather_segmented");
  };
  bool nullable_probe_output = false;
  for (std::size_t i = 0; i < selection_.size(); ++i) {
    nullable_probe_output = nullable_probe_output || (!is_selection_on_build_[i] && probe_relation_.getAttributeType(selection_[i]).nullable);
  }
  DeviceBuffer count(8);
  if (existence && !existence_by_pairs) {
    // HashSemiJoinWorkOrder / HashAntiJoinWorkOrder without residual (:795-816, :860-877): the probe tuples with / without a
    // match, projected on the probe attributes — one existence probe and one compaction over the run
    std::size_t words = 0;
    for (std::int64_t r : rows) words += static_cast<std::size_t>((r + 63) / 64) + 1;
    DeviceBuffer bitmap_storage(words * 8 + 8);
    std::vector<std::uint64_t *> bitmaps(blocks.size());
    std::size_t at = 0;
    for (std::size_t b = 0; b < blocks.size(); ++b) {
      bitmaps[b] = static_cast<std::uint64_t *>(bitmap_storage.ptr) + at;
      at += static_cast<std::size_t>((rows[b] + 63) / 64) + 1;
    }
    const bool anti = join_type_ == JoinType::kLeftAntiJoin;
    // (an anti join over NULL keys: the tuples FOUND among those looked up, then "about AND NOT found" block by block — a
    // tuple that was not looked up was not found and stays)
    const bool anti_by_complement = anti && any_null_key;
    CheckStatus(qsx_join_probe_exists_blocks_coded(hash_table_, nb, rows.data(), keys.data(), run_keys.coding(), lookup,
                                                   anti && !anti_by_complement ? 1 : 0, bitmaps.data(),
                                                   static_cast<std::int64_t *>(count.ptr), CurrentStream()),
                "qsx_join_probe_exists_blocks");
    std::int64_t selected = ReadCount(count.ptr);
    if (anti_by_complement) {
      selected = 0;
      for (std::size_t b = 0; b < blocks.size(); ++b) {
        selected += rows[b];
        if (rows[b] == 0) continue;
        if (about != nullptr && about[b] != nullptr) {
          CheckStatus(qsx_bitmap_combine(2, about[b], bitmaps[b], rows[b], bitmaps[b], CurrentStream()), "qsx_bitmap_combine");
        } else {
          CheckStatus(qsx_bitmap_combine(3, bitmaps[b], nullptr, rows[b], bitmaps[b], CurrentStream()), "qsx_bitmap_combine");
        }
      }
    }
    block_id out_id;
    BlockReference out = output_destination_->getBlockForInsertion(selected > 0 ? selected : 1, &out_id);
    std::vector<const void *> src(blocks.size() * selection_.size());
    std::vector<void *> dst;
    std::vector<std::int32_t> widths;
    for (std::size_t i = 0; i < selection_.size(); ++i) {
      dst.push_back(out->stripe(static_cast<attribute_id>(i)));
      widths.push_back(probe_relation_.getAttributeType(selection_[i]).width);
      for (std::size_t b = 0; b < blocks.size(); ++b) src[b * selection_.size() + i] = blocks[b]->stripe(selection_[i]);
    }
    const std::size_t ws_bytes = qsx_compact_blocks_workspace_bytes(nb, rows.data());
    DeviceBuffer ws(ws_bytes + 8);
    std::unique_ptr<DeviceBuffer> out_tids;   // run-global row number of every output tuple (nullable outputs: their null bits)
    if (nullable_probe_output) out_tids.reset(new DeviceBuffer(static_cast<std::size_t>(selected > 0 ? selected : 1) * 4 + 8));
    CheckStatus(qsx_compact_gather_blocks(static_cast<int>(selection_.size()), widths.data(), nb, rows.data(), src.data(),
                                          reinterpret_cast<const std::uint64_t *const *>(bitmaps.data()), nullptr, dst.data(),
                                          out_tids != nullptr ? static_cast<std::int32_t *>(out_tids->ptr) : nullptr,
                                          static_cast<std::int64_t *>(count.ptr), ws.ptr, ws_bytes, CurrentStream()),
                "qsx_compact_gather_blocks");
    const std::int64_t written = ReadCount(count.ptr);
    for (std::size_t i = 0; i < selection_.size() && out_tids != nullptr; ++i) gather_probe_nulls(i, out_tids->ptr, written, out, 0);
    if (out_tids != nullptr) CheckStatus(qsx_stream_synchronize(CurrentStream()), "qsx_stream_synchronize");
    output_destination_->returnBlock(out_id, written, getPartitionId());
    return true;
  }
  BuildSegments build(build_relation_, storage_manager_);
  // Nothing is evaluated on the pairs (an exact key, no residual predicate) and every output attribute is a plain value of
  // 1 / 2 / 4 / 8 bytes: the probe writes the output tuples itself (qsx_join_probe_project_blocks) into a block with room for
  // one match per probe tuple.  More matches than that (duplicate build keys) and the work order takes the pair list below.
  bool projectable = !outer && run_keys.exact && residual_predicate_ == nullptr && !selection_.empty() &&
                     selection_.size() <= QSX_MAX_PROJECTED && total_rows > 0;
  for (std::size_t i = 0; i < selection_.size() && projectable; ++i) {
    const Type &t = (is_selection_on_build_[i] ? build_relation_ : probe_relation_).getAttributeType(selection_[i]);
    projectable = !t.nullable && (t.width == 1 || t.width == 2 || t.width == 4 || t.width == 8);
  }
  if (projectable) {
    const std::size_t nc = selection_.size(), nseg = build.refs.size();
    qsx_join_projection_t proj{};
    proj.num_columns = static_cast<std::int32_t>(nc);
    std::vector<const void *> probe_stripes(blocks.size() * nc, nullptr), build_stripes(nseg * nc, nullptr);
    std::vector<void *> out_columns(nc);
    bool coded_key_projected = false;
    block_id out_id;
    BlockReference out = output_destination_->getBlockForInsertion(total_rows, &out_id);
    for (std::size_t i = 0; i < nc; ++i) {
      const bool on_build = is_selection_on_build_[i];
      proj.width[i] = (on_build ? build_relation_ : probe_relation_).getAttributeType(selection_[i]).width;
      proj.on_build[i] = on_build ? 1 : 0;
      out_columns[i] = out->stripe(static_cast<attribute_id>(i));
      if (on_build) {
        for (std::size_t sg = 0; sg < nseg; ++sg) build_stripes[sg * nc + i] = build.refs[sg]->stripe(selection_[i]);
      } else {
        // (the join attribute itself over blocks that hold it compressed: given as the key stripes, the probe emits its value —
        // qsx_join_probe_project_blocks_coded — and the attribute is not decoded for the sake of this column)
        const bool is_coded_key = run_keys.coded && join_key_attributes_.size() == 1 && selection_[i] == join_key_attributes_.front();
        coded_key_projected = coded_key_projected || is_coded_key;
        for (std::size_t b = 0; b < blocks.size(); ++b) {
          probe_stripes[b * nc + i] = is_coded_key ? keys[b] : blocks[b]->stripe(selection_[i]);
        }
      }
    }
    proj.probe_stripes = probe_stripes.data();
    proj.num_build_segments = static_cast<std::int32_t>(nseg);
    proj.build_first_tids = build.first_rows.data();
    proj.build_stripes = build_stripes.data();
    proj.out_columns = out_columns.data();
    int project_status = qsx_join_probe_project_blocks_coded(hash_table_, nb, rows.data(), keys.data(), run_keys.coding(), lookup, &proj,
                                                             total_rows, static_cast<std::int64_t *>(count.ptr), CurrentStream());
    if (project_status == QSX_ERR_UNSUPPORTED && coded_key_projected) {
      // a table without a directly addressed form gathers the output from the stripes: the key column as decoded values then
      for (std::size_t i = 0; i < nc; ++i) {
        if (is_selection_on_build_[i] || selection_[i] != join_key_attributes_.front()) continue;
        for (std::size_t b = 0; b < blocks.size(); ++b) probe_stripes[b * nc + i] = blocks[b]->stripe(selection_[i]);
      }
      project_status = qsx_join_probe_project_blocks_coded(hash_table_, nb, rows.data(), keys.data(), run_keys.coding(), lookup, &proj,
                                                           total_rows, static_cast<std::int64_t *>(count.ptr), CurrentStream());
    }
    CheckStatus(project_status, "qsx_join_probe_project_blocks");
    const std::int64_t matches = ReadCount(count.ptr);   // (synchronises the stream: the block's tuples are written)
    if (matches <= total_rows) {
      output_destination_->returnBlock(out_id, matches, getPartitionId());
      return true;
    }
    out = BlockReference();
    storage_manager_->deleteBlockOrBlobFile(out_id);   // never returned to the destination: nobody else knows the block
  }
  // No counting pass: the pair lists get room for one match per probe tuple — what a foreign-key probe of a primary-key
  // build side produces at most (the reference sizes from the same uniqueness fact, impliesUniqueAttributes).  The probe
  // counts every match it finds, also those that did not fit: a build side with duplicate keys makes this work order probe
  // once more with the exact capacity.
  JoinedPairs pairs;
  std::int64_t room = total_rows > 0 ? total_rows : 1;
  std::vector<std::int32_t> base_tids;   // semi / anti by pairs: the word-aligned first tuple id of every block (first_rows)
  if (existence_by_pairs || outer) base_tids.assign(first_rows.begin(), first_rows.end());
  for (int attempt = 0; attempt < 2; ++attempt) {
    pairs.probe_tids.reset(new DeviceBuffer(static_cast<std::size_t>(room) * 4 + 8));
    pairs.build_tids.reset(new DeviceBuffer(static_cast<std::size_t>(room) * 4 + 8));
    CheckStatus(qsx_join_probe_blocks_coded(hash_table_, nb, rows.data(), keys.data(), run_keys.coding(),
                                            base_tids.empty() ? nullptr : base_tids.data(), lookup,
                                            static_cast<std::int32_t *>(pairs.probe_tids->ptr),
                                      static_cast<std::int32_t *>(pairs.build_tids->ptr), room, static_cast<std::int64_t *>(count.ptr),
                                      CurrentStream()), "qsx_join_probe_blocks");
    pairs.count = ReadCount(count.ptr);
    if (pairs.count <= room) break;
    if (attempt == 1) throw ExecutionError("HashJoinOperator: the match count changed between two probes of one run", QSX_ERR_CAPACITY);
    room = pairs.count;
  }
  std::vector<const void *> segments(blocks.size());
  std::vector<ComparisonPredicate> terms;
  if (!run_keys.exact) {   // compositeKeyCollisionCheck (SeparateChainingHashTable.hpp:1046): equal folds, equal components?
    for (std::size_t k = 0; k < join_key_attributes_.size(); ++k) {
      terms.push_back(ComparisonPredicate::Attributes(join_key_attributes_[k], false, ComparisonID::kEqual, build_key_attributes_[k], true));
    }
  }
  if (residual_predicate_ != nullptr) terms.insert(terms.end(), residual_predicate_->conjuncts.begin(), residual_predicate_->conjuncts.end());
  if (!terms.empty() && pairs.count > 0) {
    // matchesForJoinedTuples (:510-524) on the pairs of the run: each term's operands gathered by the pair lists (the probe
    // side through the run's own stripes), compared, chained through the filter bitmap like a conjunction
    const std::int64_t m = pairs.count;
    const std::size_t pair_bitmap_bytes = static_cast<std::size_t>((m + 63) / 64) * 8 + 8;
    std::size_t widest = 8;
    for (const ComparisonPredicate &term : terms) {
      widest = std::max<std::size_t>(widest, (term.on_build_side ? build_relation_ : probe_relation_).getAttributeType(term.attribute).width);
    }
    DeviceBuffer current(pair_bitmap_bytes), next(pair_bitmap_bytes), lhs(static_cast<std::size_t>(m) * widest + 8), rhs(static_cast<std::size_t>(m) * 8 + 8);
    void *cur = current.ptr, *nxt = next.ptr;
    bool first = true;
    auto gather_side = [&](attribute_id attr, bool on_build, void *dst) -> Type {
      const Type t = (on_build ? build_relation_ : probe_relation_).getAttributeType(attr);
      if (on_build) {
        build.gather(attr, t.width, pairs.build_tids->ptr, m, dst);
      } else {
        for (std::size_t b = 0; b < blocks.size(); ++b) segments[b] = blocks[b]->stripe(attr);
        CheckStatus(qsx_gather_segmented(t.width, static_cast<int>(segments.size()), segments.data(), first_rows.data(),
                                         static_cast<const std::int32_t *>(pairs.probe_tids->ptr), m, dst, CurrentStream()),
                    "qsx_gather_segmented");
      }
      return t;
    };
    for (const ComparisonPredicate &term : terms) {
      const Type t = gather_side(term.attribute, term.on_build_side, lhs.ptr);   // (operand types were vetted above)
      if (term.rhs_attribute != kInvalidAttributeID) {
        const Type rt = gather_side(term.rhs_attribute, term.rhs_on_build_side, rhs.ptr);
        if (rt.id != t.id) throw ExecutionError("join predicate compares attributes of different types", QSX_ERR_UNSUPPORTED);
        CheckStatus(qsx_select_cmp_columns(t.id, lhs.ptr, rhs.ptr, m, static_cast<int>(term.comparison),
                                           first ? nullptr : static_cast<const std::uint64_t *>(cur), static_cast<std::uint64_t *>(nxt), nullptr,
                                           CurrentStream()), "qsx_select_cmp_columns");
      } else if (t.id == kChar) {
        CheckStatus(qsx_select_cmp_char(lhs.ptr, t.width, m, static_cast<int>(term.comparison), term.literal.text.data(),
                                        static_cast<int>(term.literal.text.size()), first ? nullptr : static_cast<const std::uint64_t *>(cur),
                                        static_cast<std::uint64_t *>(nxt), nullptr, CurrentStream()), "qsx_select_cmp_char");
      } else {
        CheckStatus(qsx_select_cmp(t.id, lhs.ptr, m, static_cast<int>(term.comparison), &term.literal.v,
                                   first ? nullptr : static_cast<const std::uint64_t *>(cur), static_cast<std::uint64_t *>(nxt), nullptr,
                                   CurrentStream()), "qsx_select_cmp");
      }
      std::swap(cur, nxt);
      first = false;
    }
    if (!first) CompactPairs(&pairs, cur);
  }
  if (existence_by_pairs) {
    // the probe tuples that kept a pair, as one bitmap over the run's (word-aligned) tuple ids; anti: its complement, block by
    // block (the complement of a block's part ends at the block's last tuple); then one compaction over the run
    const std::size_t words = static_cast<std::size_t>(total_rows / 64) + 1;
    DeviceBuffer bitmap(words * 8 + 8);
    CheckStatus(qsx_tids_to_bitmap(static_cast<const std::int32_t *>(pairs.probe_tids->ptr), pairs.count, 0, total_rows,
                                   static_cast<std::uint64_t *>(bitmap.ptr), CurrentStream()), "qsx_tids_to_bitmap");
    std::vector<std::uint64_t *> bitmaps(blocks.size());
    std::int64_t upper = 0;   // tuples the compaction can select at most
    for (std::size_t b = 0; b < blocks.size(); ++b) {
      bitmaps[b] = static_cast<std::uint64_t *>(bitmap.ptr) + first_rows[b] / 64;
      if (join_type_ == JoinType::kLeftAntiJoin && rows[b] > 0) {
        // the tuples WITHOUT a surviving pair — among those this work order is about: what a LIP filter rejected must not
        // come back through the complement
        if (about != nullptr && about[b] != nullptr) {
          CheckStatus(qsx_bitmap_combine(2, about[b], bitmaps[b], rows[b], bitmaps[b], CurrentStream()), "qsx_bitmap_combine");
        } else {
          CheckStatus(qsx_bitmap_combine(3, bitmaps[b], nullptr, rows[b], bitmaps[b], CurrentStream()), "qsx_bitmap_combine");
        }
      }
      upper += rows[b];
    }
    if (join_type_ == JoinType::kLeftSemiJoin) upper = std::min<std::int64_t>(upper, pairs.count);
    block_id out_id;
    BlockReference out = output_destination_->getBlockForInsertion(upper > 0 ? upper : 1, &out_id);
    std::vector<const void *> src(blocks.size() * selection_.size());
    std::vector<void *> dst;
    std::vector<std::int32_t> widths;
    for (std::size_t i = 0; i < selection_.size(); ++i) {
      dst.push_back(out->stripe(static_cast<attribute_id>(i)));
      widths.push_back(probe_relation_.getAttributeType(selection_[i]).width);
      for (std::size_t b = 0; b < blocks.size(); ++b) src[b * selection_.size() + i] = blocks[b]->stripe(selection_[i]);
    }
    const std::size_t ws_bytes = qsx_compact_blocks_workspace_bytes(nb, rows.data());
    DeviceBuffer ws(ws_bytes + 8);
    std::unique_ptr<DeviceBuffer> out_tids;   // the (word-aligned, like the bitmap) tuple id of every output tuple
    if (nullable_probe_output) out_tids.reset(new DeviceBuffer(static_cast<std::size_t>(upper > 0 ? upper : 1) * 4 + 8));
    CheckStatus(qsx_compact_gather_blocks(static_cast<int>(selection_.size()), widths.data(), nb, rows.data(), src.data(),
                                          reinterpret_cast<const std::uint64_t *const *>(bitmaps.data()), out_tids != nullptr ? base_tids.data() : nullptr,
                                          dst.data(), out_tids != nullptr ? static_cast<std::int32_t *>(out_tids->ptr) : nullptr,
                                          static_cast<std::int64_t *>(count.ptr), ws.ptr, ws_bytes, CurrentStream()),
                "qsx_compact_gather_blocks");
    const std::int64_t written = ReadCount(count.ptr);
    for (std::size_t i = 0; i < selection_.size() && out_tids != nullptr; ++i) gather_probe_nulls(i, out_tids->ptr, written, out, 0);
    if (out_tids != nullptr) CheckStatus(qsx_stream_synchronize(CurrentStream()), "qsx_stream_synchronize");
    output_destination_->returnBlock(out_id, written, getPartitionId());
    return true;
  }
  if (outer) {
    // HashOuterJoinWorkOrder (:1026-1099) over the run: the matched pairs, then the probe tuples without one (the complement of
    // the matched tuples' bitmap, block by block so that it ends at each block's last tuple, AND the LIP filter's survivors)
    // with NULL build-side attributes
    const std::int64_t matches = pairs.count;
    const std::size_t words = static_cast<std::size_t>(total_rows / 64) + 1;
    DeviceBuffer bitmap(words * 8 + 8);
    CheckStatus(qsx_tids_to_bitmap(static_cast<const std::int32_t *>(pairs.probe_tids->ptr), matches, 0, total_rows,
                                   static_cast<std::uint64_t *>(bitmap.ptr), CurrentStream()), "qsx_tids_to_bitmap");
    for (std::size_t b = 0; b < blocks.size(); ++b) {
      if (rows[b] == 0) continue;
      std::uint64_t *part = static_cast<std::uint64_t *>(bitmap.ptr) + first_rows[b] / 64;
      CheckStatus(qsx_bitmap_combine(3, part, nullptr, rows[b], part, CurrentStream()), "qsx_bitmap_combine");
      if (about != nullptr && about[b] != nullptr) {   // (NOT `lookup`: a tuple with a NULL key is NULL-padded, not dropped)
        CheckStatus(qsx_bitmap_combine(0, part, about[b], rows[b], part, CurrentStream()), "qsx_bitmap_combine");
      }
    }
    DeviceBuffer unmatched_tids(static_cast<std::size_t>(total_rows) * 4 + 8);
    const std::size_t ws_bytes = qsx_compact_workspace_bytes(total_rows);
    DeviceBuffer ws(ws_bytes + 8);
    CheckStatus(qsx_bitmap_to_tids(static_cast<const std::uint64_t *>(bitmap.ptr), total_rows, 0, static_cast<std::int32_t *>(unmatched_tids.ptr),
                                   static_cast<std::int64_t *>(count.ptr), ws.ptr, ws_bytes, CurrentStream()), "qsx_bitmap_to_tids");
    const std::int64_t unmatched = ReadCount(count.ptr);
    const std::int64_t total = matches + unmatched;
    block_id out_id;
    BlockReference out = output_destination_->getBlockForInsertion(total > 0 ? total : 1, &out_id);
    std::unique_ptr<DeviceBuffer> padded_build_tids;   // the pairs' build tuple ids, then -1 for every unmatched probe tuple
    std::unique_ptr<DeviceBuffer> all_probe_tids;      // the pairs' probe tuple ids, then the unmatched ones: the output's order
    if (nullable_probe_output && total > 0) {
      all_probe_tids.reset(new DeviceBuffer(static_cast<std::size_t>(total) * 4 + 8));
      if (matches > 0) {
        CheckStatus(qsx_copy_on_device(all_probe_tids->ptr, pairs.probe_tids->ptr, static_cast<std::size_t>(matches) * 4, CurrentStream()), "qsx_copy_on_device");
      }
      if (unmatched > 0) {
        CheckStatus(qsx_copy_on_device(static_cast<char *>(all_probe_tids->ptr) + static_cast<std::size_t>(matches) * 4, unmatched_tids.ptr,
                                       static_cast<std::size_t>(unmatched) * 4, CurrentStream()), "qsx_copy_on_device");
      }
    }
    for (std::size_t i = 0; i < selection_.size(); ++i) {
      char *dst = static_cast<char *>(out->stripe(static_cast<attribute_id>(i)));
      const bool on_build = is_selection_on_build_[i];
      const Type &t = (on_build ? build_relation_ : probe_relation_).getAttributeType(selection_[i]);
      char *tail = dst + static_cast<std::size_t>(matches) * t.width;
      if (on_build) {
        std::uint64_t *nulls = out->nullBitmap(static_cast<attribute_id>(i));
        if (nulls == nullptr) throw ExecutionError("outer join output attribute taken from the build side must be nullable", QSX_ERR_INVALID_ARGUMENT);
        if (matches > 0) build.gather(selection_[i], t.width, pairs.build_tids->ptr, matches, dst);
        if (unmatched > 0) CheckStatus(qsx_memset_device(tail, 0, static_cast<std::size_t>(unmatched) * t.width, CurrentStream()), "qsx_memset_device");
        if (total > 0) {
          if (padded_build_tids == nullptr) {
            padded_build_tids.reset(new DeviceBuffer(static_cast<std::size_t>(total) * 4 + 8));
            if (matches > 0) {
              CheckStatus(qsx_copy_on_device(padded_build_tids->ptr, pairs.build_tids->ptr, static_cast<std::size_t>(matches) * 4, CurrentStream()),
                          "qsx_copy_on_device");
            }
            if (unmatched > 0) {
              CheckStatus(qsx_memset_device(static_cast<char *>(padded_build_tids->ptr) + static_cast<std::size_t>(matches) * 4, 0xFF,
                                            static_cast<std::size_t>(unmatched) * 4, CurrentStream()), "qsx_memset_device");
            }
          }
          build.gatherNulls(selection_[i], padded_build_tids->ptr, total, nulls);
        }
      } else {
        for (std::size_t b = 0; b < blocks.size(); ++b) segments[b] = blocks[b]->stripe(selection_[i]);
        if (matches > 0) {
          CheckStatus(qsx_gather_segmented(t.width, static_cast<int>(segments.size()), segments.data(), first_rows.data(),
                                           static_cast<const std::int32_t *>(pairs.probe_tids->ptr), matches, dst, CurrentStream()),
                      "qsx_gather_segmented");
        }
        if (unmatched > 0) {
          CheckStatus(qsx_gather_segmented(t.width, static_cast<int>(segments.size()), segments.data(), first_rows.data(),
                                           static_cast<const std::int32_t *>(unmatched_tids.ptr), unmatched, tail, CurrentStream()),
                      "qsx_gather_segmented");
        }
        if (all_probe_tids != nullptr) gather_probe_nulls(i, all_probe_tids->ptr, total, out, 0);
      }
    }
    CheckStatus(qsx_stream_synchronize(CurrentStream()), "qsx_stream_synchronize");
    output_destination_->returnBlock(out_id, total, getPartitionId());
    return true;
  }
  const std::int64_t matches = pairs.count;
  DeviceBuffer &probe_tids = *pairs.probe_tids, &build_tids = *pairs.build_tids;
  block_id out_id;
  BlockReference out = output_destination_->getBlockForInsertion(matches > 0 ? matches : 1, &out_id);
  for (std::size_t i = 0; i < selection_.size(); ++i) {
    void *dst = out->stripe(static_cast<attribute_id>(i));
    const bool on_build = is_selection_on_build_[i];
    const Type &t = (on_build ? build_relation_ : probe_relation_).getAttributeType(selection_[i]);
    if (on_build) {
      build.gather(selection_[i], t.width, build_tids.ptr, matches, dst);
      if (t.nullable && matches > 0) {
        std::uint64_t *nulls = out->nullBitmap(static_cast<attribute_id>(i));
        if (nulls == nullptr) throw ExecutionError("join output of a nullable attribute must be nullable", QSX_ERR_INVALID_ARGUMENT);
        build.gatherNulls(selection_[i], build_tids.ptr, matches, nulls);
      }
    } else {
      for (std::size_t b = 0; b < blocks.size(); ++b) segments[b] = blocks[b]->stripe(selection_[i]);
      CheckStatus(qsx_gather_segmented(t.width, static_cast<int>(segments.size()), segments.data(), first_rows.data(),
                                       static_cast<const std::int32_t *>(probe_tids.ptr), matches, dst, CurrentStream()),
                  "qsx_gather_segmented");
      gather_probe_nulls(i, probe_tids.ptr, matches, out, 0);
    }
  }
  CheckStatus(qsx_stream_synchronize(CurrentStream()), "qsx_stream_synchronize");
  output_destination_->returnBlock(out_id, matches, getPartitionId());
  return true;
}

void HashInnerJoinWorkOrder::executeBlock(block_id probe_block_id) {
  using JoinType = HashJoinOperator::JoinType;
  BlockReference probe = storage_manager_->getBlock(probe_block_id);
  const std::int64_t n = probe->numTuples();
  JoinKeys keys(*probe, join_key_attributes_);
  DeviceBuffer count(8);
  const std::size_t bitmap_bytes = static_cast<std::size_t>((n + 63) / 64) * 8 + 8;
  // existence_map of the LIPFilterAdaptiveProber (:462-470): the probe tuples this work order looks at
  struct LipBitmap {
    void *ptr = nullptr;
    ~LipBitmap() { qsx_device_free(ptr); }
  } lip_holder;
  if (lip_filter_adaptive_prober_ != nullptr) lip_holder.ptr = lip_filter_adaptive_prober_->filterValueAccessor(*probe, nullptr, nullptr);
  const std::uint64_t *lip = static_cast<const std::uint64_t *>(lip_holder.ptr);
  // check_for_null_keys: a probe tuple with a NULL key component is not looked up (HashTable.hpp:2158-2160, 1855-1865):
  // it matches nothing — out of an inner / semi join, NULL-padded in an outer join, kept by an anti join.
  // `lookup` = the tuples that are looked up, `lip` stays the set of tuples this work order is about.
  const std::unique_ptr<DeviceBuffer> not_null_keys = NotNullFilter(*probe, join_key_attributes_, lip);
  const std::uint64_t *lookup = not_null_keys != nullptr ? static_cast<const std::uint64_t *>(not_null_keys->ptr) : lip;
  const bool pairs_needed = join_type_ == JoinType::kInnerJoin || join_type_ == JoinType::kLeftOuterJoin ||
                            residual_predicate_ != nullptr || !keys.exact;

  JoinedPairs pairs;
  std::unique_ptr<BuildSegments> build;
  if (pairs_needed) {
    // hash_table_.getAllFromValueAccessor[CompositeKey](accessor, key(s), nullable, &collector) (:480-485)
    CheckStatus(qsx_join_probe_count(hash_table_, keys.ptr, n, lookup, static_cast<std::int64_t *>(count.ptr), CurrentStream()),
                "qsx_join_probe_count");
    pairs.count = ReadCount(count.ptr);
    pairs.probe_tids.reset(new DeviceBuffer(static_cast<std::size_t>(pairs.count) * 4 + 8));
    pairs.build_tids.reset(new DeviceBuffer(static_cast<std::size_t>(pairs.count) * 4 + 8));
    CheckStatus(qsx_join_probe(hash_table_, keys.ptr, n, /*probe_base_tid=*/0, lookup,
                               static_cast<std::int32_t *>(pairs.probe_tids->ptr), static_cast<std::int32_t *>(pairs.build_tids->ptr),
                               pairs.count, static_cast<std::int64_t *>(count.ptr), CurrentStream()), "qsx_join_probe");
    build.reset(new BuildSegments(build_relation_, storage_manager_));

    // Terms evaluated on the pairs: the component equalities of a hashed composite key
    // (compositeKeyCollisionCheck, SeparateChainingHashTable.hpp:1046) and the residual predicate
    // (matchesForJoinedTuples, :510-524), chained through the filter bitmap like a conjunction.
    std::vector<ComparisonPredicate> terms;
    if (!keys.exact) {
      for (std::size_t k = 0; k < join_key_attributes_.size(); ++k) {
        terms.push_back(ComparisonPredicate::Attributes(join_key_attributes_[k], false, ComparisonID::kEqual,
                                                        build_key_attributes_[k], true));
      }
    }
    if (residual_predicate_ != nullptr) {
      terms.insert(terms.end(), residual_predicate_->conjuncts.begin(), residual_predicate_->conjuncts.end());
    }
    if (!terms.empty() && pairs.count > 0) {
      const std::int64_t m = pairs.count;
      const std::size_t pair_bitmap_bytes = static_cast<std::size_t>((m + 63) / 64) * 8 + 8;
      std::size_t widest = 8;
      for (const ComparisonPredicate &term : terms) {
        widest = std::max<std::size_t>(widest, (term.on_build_side ? build_relation_ : probe_relation_).getAttributeType(term.attribute).width);
      }
      DeviceBuffer current(pair_bitmap_bytes), next(pair_bitmap_bytes), lhs(static_cast<std::size_t>(m) * widest + 8),
          rhs(static_cast<std::size_t>(m) * 8 + 8);
      void *cur = current.ptr, *nxt = next.ptr;
      bool first = true;
      auto gather_side = [&](attribute_id attr, bool on_build, void *dst) -> Type {
        const Type t = (on_build ? build_relation_ : probe_relation_).getAttributeType(attr);
        if (on_build) {
          build->gather(attr, t.width, pairs.build_tids->ptr, m, dst);
        } else {
          CheckStatus(qsx_gather(t.width, probe->stripe(attr), static_cast<const std::int32_t *>(pairs.probe_tids->ptr), m, dst,
                                 CurrentStream()), "qsx_gather");
        }
        return t;
      };
      for (const ComparisonPredicate &term : terms) {
        const Type t = gather_side(term.attribute, term.on_build_side, lhs.ptr);
        if (term.rhs_attribute != kInvalidAttributeID) {
          if (t.id == kChar || t.id == kVarChar) {
            throw ExecutionError("join predicate compares two string attributes (only string = literal is supported)", QSX_ERR_UNSUPPORTED);
          }
          const Type rt = gather_side(term.rhs_attribute, term.rhs_on_build_side, rhs.ptr);
          if (rt.id != t.id) throw ExecutionError("join predicate compares attributes of different types", QSX_ERR_UNSUPPORTED);
          CheckStatus(qsx_select_cmp_columns(t.id, lhs.ptr, rhs.ptr, m, static_cast<int>(term.comparison),
                                             first ? nullptr : static_cast<const std::uint64_t *>(cur),
                                             static_cast<std::uint64_t *>(nxt), nullptr, CurrentStream()),
                      "qsx_select_cmp_columns");
        } else if (t.id == kChar) {
          CheckStatus(qsx_select_cmp_char(lhs.ptr, t.width, m, static_cast<int>(term.comparison), term.literal.text.data(),
                                          static_cast<int>(term.literal.text.size()), first ? nullptr : static_cast<const std::uint64_t *>(cur),
                                          static_cast<std::uint64_t *>(nxt), nullptr, CurrentStream()), "qsx_select_cmp_char");
        } else {
          CheckStatus(qsx_select_cmp(t.id, lhs.ptr, m, static_cast<int>(term.comparison), &term.literal.v,
                                     first ? nullptr : static_cast<const std::uint64_t *>(cur),
                                     static_cast<std::uint64_t *>(nxt), nullptr, CurrentStream()), "qsx_select_cmp");
        }
        std::swap(cur, nxt);
        first = false;
      }
      CompactPairs(&pairs, cur);
    }
  }

  if (join_type_ == JoinType::kLeftSemiJoin || join_type_ == JoinType::kLeftAntiJoin) {
    // HashSemiJoinWorkOrder / HashAntiJoinWorkOrder (:680-877, :880-1000): the probe tuples with
    // (semi) / without (anti) a surviving match, projected on the probe attributes.
    const bool anti = join_type_ == JoinType::kLeftAntiJoin;
    DeviceBuffer bitmap(bitmap_bytes);
    if (pairs_needed) {
      CheckStatus(qsx_tids_to_bitmap(static_cast<const std::int32_t *>(pairs.probe_tids->ptr), pairs.count, 0, n,
                                     static_cast<std::uint64_t *>(bitmap.ptr), CurrentStream()), "qsx_tids_to_bitmap");
      if (anti) {
        CheckStatus(qsx_bitmap_combine(3, static_cast<const std::uint64_t *>(bitmap.ptr), nullptr, n,
                                       static_cast<std::uint64_t *>(bitmap.ptr), CurrentStream()), "qsx_bitmap_combine");
        if (lip != nullptr) {
          CheckStatus(qsx_bitmap_combine(0, static_cast<const std::uint64_t *>(bitmap.ptr), lip, n,
                                         static_cast<std::uint64_t *>(bitmap.ptr), CurrentStream()), "qsx_bitmap_combine");
        }
      }
      CheckStatus(qsx_bitmap_count(static_cast<const std::uint64_t *>(bitmap.ptr), n, static_cast<std::int64_t *>(count.ptr),
                                   CurrentStream()), "qsx_bitmap_count");
    } else if (anti && lookup != lip) {
      // NULL keys are not looked up and therefore survive the anti join: tuples \ (looked-up tuples with a match)
      CheckStatus(qsx_join_probe_exists(hash_table_, keys.ptr, n, lookup, 0, static_cast<std::uint64_t *>(bitmap.ptr),
                                        static_cast<std::int64_t *>(count.ptr), CurrentStream()), "qsx_join_probe_exists");
      CheckStatus(qsx_bitmap_combine(3, static_cast<const std::uint64_t *>(bitmap.ptr), nullptr, n,
                                     static_cast<std::uint64_t *>(bitmap.ptr), CurrentStream()), "qsx_bitmap_combine");
      if (lip != nullptr) {
        CheckStatus(qsx_bitmap_combine(0, static_cast<const std::uint64_t *>(bitmap.ptr), lip, n,
                                       static_cast<std::uint64_t *>(bitmap.ptr), CurrentStream()), "qsx_bitmap_combine");
      }
      CheckStatus(qsx_bitmap_count(static_cast<const std::uint64_t *>(bitmap.ptr), n, static_cast<std::int64_t *>(count.ptr),
                                   CurrentStream()), "qsx_bitmap_count");
    } else {
      CheckStatus(qsx_join_probe_exists(hash_table_, keys.ptr, n, lookup, anti ? 1 : 0,
                                        static_cast<std::uint64_t *>(bitmap.ptr), static_cast<std::int64_t *>(count.ptr),
                                        CurrentStream()), "qsx_join_probe_exists");
    }
    const std::int64_t matches = ReadCount(count.ptr);
    block_id out_id;
    BlockReference out = output_destination_->getBlockForInsertion(matches > 0 ? matches : 1, &out_id);
    std::vector<const void *> src;
    std::vector<void *> dst;
    std::vector<std::int32_t> widths;
    for (std::size_t i = 0; i < selection_.size(); ++i) {
      src.push_back(probe->stripe(selection_[i]));
      dst.push_back(out->stripe(static_cast<attribute_id>(i)));
      widths.push_back(probe_relation_.getAttributeType(selection_[i]).width);
    }
    const std::size_t ws_bytes = qsx_compact_workspace_bytes(n);
    DeviceBuffer ws(ws_bytes);
    CheckStatus(qsx_compact_gather(static_cast<int>(src.size()), src.data(), widths.data(),
                                   static_cast<const std::uint64_t *>(bitmap.ptr), n, dst.data(),
                                   static_cast<std::int64_t *>(count.ptr), ws.ptr, ws_bytes, CurrentStream()),
                "qsx_compact_gather");
    const std::int64_t written = ReadCount(count.ptr);
    ProjectNullBitmaps(*probe, selection_, bitmap.ptr, written, out.get());
    output_destination_->returnBlock(out_id, written, getPartitionId());
    return;
  }

  // Inner / left outer: matched pairs first, then (outer) the probe tuples without a match with
  // NULL build-side attributes (HashOuterJoinWorkOrder, :1026-1099).
  const std::int64_t matches = pairs.count;
  std::int64_t unmatched = 0;
  std::unique_ptr<DeviceBuffer> unmatched_tids;
  if (join_type_ == JoinType::kLeftOuterJoin) {
    DeviceBuffer bitmap(bitmap_bytes);
    CheckStatus(qsx_tids_to_bitmap(static_cast<const std::int32_t *>(pairs.probe_tids->ptr), matches, 0, n,
                                   static_cast<std::uint64_t *>(bitmap.ptr), CurrentStream()), "qsx_tids_to_bitmap");
    CheckStatus(qsx_bitmap_combine(3, static_cast<const std::uint64_t *>(bitmap.ptr), nullptr, n,
                                   static_cast<std::uint64_t *>(bitmap.ptr), CurrentStream()), "qsx_bitmap_combine");
    if (lip != nullptr) {
      CheckStatus(qsx_bitmap_combine(0, static_cast<const std::uint64_t *>(bitmap.ptr), lip, n,
                                     static_cast<std::uint64_t *>(bitmap.ptr), CurrentStream()), "qsx_bitmap_combine");
    }
    unmatched_tids.reset(new DeviceBuffer(static_cast<std::size_t>(n) * 4 + 8));
    const std::size_t ws_bytes = qsx_compact_workspace_bytes(n);
    DeviceBuffer ws(ws_bytes);
    CheckStatus(qsx_bitmap_to_tids(static_cast<const std::uint64_t *>(bitmap.ptr), n, 0, static_cast<std::int32_t *>(unmatched_tids->ptr),
                                   static_cast<std::int64_t *>(count.ptr), ws.ptr, ws_bytes, CurrentStream()), "qsx_bitmap_to_tids");
    unmatched = ReadCount(count.ptr);
  }
  const std::int64_t total = matches + unmatched;
  block_id out_id;
  BlockReference out = output_destination_->getBlockForInsertion(total > 0 ? total : 1, &out_id);
  // Row numbers of all `total` output rows per side, for the null bits: pairs first, then (outer) the unmatched probe
  // tuples next to build tid -1 = NULL padding.  Only materialised when some output attribute can be NULL.
  std::unique_ptr<DeviceBuffer> all_probe_tids, all_build_tids;
  auto all_tids = [&](bool on_build) -> const void * {
    std::unique_ptr<DeviceBuffer> &buf = on_build ? all_build_tids : all_probe_tids;
    if (unmatched == 0) return on_build ? pairs.build_tids->ptr : pairs.probe_tids->ptr;
    if (buf == nullptr) {
      buf.reset(new DeviceBuffer(static_cast<std::size_t>(total) * 4 + 8));
      char *tail = static_cast<char *>(buf->ptr) + static_cast<std::size_t>(matches) * 4;
      CheckStatus(qsx_copy_on_device(buf->ptr, on_build ? pairs.build_tids->ptr : pairs.probe_tids->ptr,
                                     static_cast<std::size_t>(matches) * 4, CurrentStream()), "qsx_copy_on_device");
      if (on_build) {
        CheckStatus(qsx_memset_device(tail, 0xFF, static_cast<std::size_t>(unmatched) * 4, CurrentStream()), "qsx_memset_device");
      } else {
        CheckStatus(qsx_copy_on_device(tail, unmatched_tids->ptr, static_cast<std::size_t>(unmatched) * 4, CurrentStream()),
                    "qsx_copy_on_device");
      }
    }
    return buf->ptr;
  };
  for (std::size_t i = 0; i < selection_.size(); ++i) {
    // Scalar::getAllValuesForJoin (:529-536)
    char *dst = static_cast<char *>(out->stripe(static_cast<attribute_id>(i)));
    const bool on_build = is_selection_on_build_[i];
    const int width = (on_build ? build_relation_ : probe_relation_).getAttributeType(selection_[i]).width;
    if (on_build) {
      build->gather(selection_[i], width, pairs.build_tids->ptr, matches, dst);
    } else {
      CheckStatus(qsx_gather(width, probe->stripe(selection_[i]), static_cast<const std::int32_t *>(pairs.probe_tids->ptr), matches,
                             dst, CurrentStream()), "qsx_gather");
    }
    if (unmatched > 0) {
      char *tail = dst + static_cast<std::size_t>(matches) * width;
      if (on_build) {
        // result->fillWithNulls() (:1077-1080): zero bytes here, the null bits of rows [matches, total) below
        CheckStatus(qsx_memset_device(tail, 0, static_cast<std::size_t>(unmatched) * width, CurrentStream()), "qsx_memset_device");
        if (out->nullBitmap(static_cast<attribute_id>(i)) == nullptr) {
          throw ExecutionError("outer join output attribute taken from the build side must be nullable", QSX_ERR_INVALID_ARGUMENT);
        }
      } else {
        CheckStatus(qsx_gather(width, probe->stripe(selection_[i]), static_cast<const std::int32_t *>(unmatched_tids->ptr), unmatched,
                               tail, CurrentStream()), "qsx_gather");
      }
    }
    // null bits of the output attribute: the source attribute's bits at the joined rows, 1 under the outer join's padding
    const bool source_nullable = (on_build ? build_relation_ : probe_relation_).getAttributeType(selection_[i]).nullable;
    if (total > 0 && (source_nullable || (on_build && unmatched > 0))) {
      std::uint64_t *nulls = out->nullBitmap(static_cast<attribute_id>(i));
      if (nulls == nullptr) throw ExecutionError("join output of a nullable attribute must be nullable", QSX_ERR_INVALID_ARGUMENT);
      if (on_build) {
        build->gatherNulls(selection_[i], all_tids(true), total, nulls);
      } else {
        GatherBlockNulls(*probe, selection_[i], all_tids(false), total, nulls);
      }
    }
  }
  CheckStatus(qsx_stream_synchronize(CurrentStream()), "qsx_stream_synchronize");
  output_destination_->returnBlock(out_id, total, getPartitionId());  // output_destination_->bulkInsertTuples(&temp_result) (:539)
}

namespace {
class DestroyHashWorkOrder : public WorkOrder {
 public:
  DestroyHashWorkOrder(std::size_t query_id, QueryContext::join_hash_table_id id, QueryContext *ctx, partition_id part)
      : WorkOrder(query_id, part), id_(id), ctx_(ctx) {}
  void execute() override { ctx_->destroyJoinHashTable(id_, partition_id_); }  // DestroyHashOperator.cpp:70-72
 private:
  QueryContext::join_hash_table_id id_;
  QueryContext *ctx_;
};
}  // namespace

bool DestroyHashOperator::getAllWorkOrders(WorkOrdersContainer *container, QueryContext *query_context, StorageManager *,
                                           const tmb::client_id, tmb::MessageBus *) {
  if (!work_generated_) {
    work_generated_ = true;
    for (partition_id part = 0; part < num_partitions_; ++part) {   // DestroyHashOperator.cpp:40-50
      container->addNormalWorkOrder(new DestroyHashWorkOrder(query_id_, hash_table_index_, query_context, part), op_index_);
    }
  }
  return true;
}


}  // namespace quickstep
