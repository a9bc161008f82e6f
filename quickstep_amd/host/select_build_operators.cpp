// select_build_operators.cpp — SelectOperator and BuildHashOperator with their work orders (see quickstep_gpu.hpp; what the files share: quickstep_gpu_internal.hpp)
#include "quickstep_gpu_internal.hpp"

namespace quickstep {

// ---------------------------------------------------------------------------
// Select
// ---------------------------------------------------------------------------
SelectOperator::SelectOperator(std::size_t query_id, const CatalogRelation &input_relation, bool has_repartition,
                               const CatalogRelation &output_relation,
                               QueryContext::insert_destination_id output_destination_index,
                               QueryContext::predicate_id predicate_index, std::vector<attribute_id> &&selection,
                               bool input_relation_is_stored, bool on_gpu)
    : RelationalOperator(query_id, 1, has_repartition), input_relation_(input_relation),
      output_relation_(output_relation), output_destination_index_(output_destination_index),
      predicate_index_(predicate_index), simple_selection_(std::move(selection)),
      input_relation_is_stored_(input_relation_is_stored), on_gpu_(on_gpu) {
  if (input_relation_is_stored) input_relation_block_ids_ = input_relation.getBlocksSnapshot();
}

SelectOperator::SelectOperator(std::size_t query_id, const CatalogRelation &input_relation, bool has_repartition,
                               const CatalogRelation &output_relation,
                               QueryContext::insert_destination_id output_destination_index,
                               QueryContext::predicate_id predicate_index, std::vector<ScalarPtr> &&selection,
                               bool input_relation_is_stored)
    : RelationalOperator(query_id, 1, has_repartition), input_relation_(input_relation),
      output_relation_(output_relation), output_destination_index_(output_destination_index),
      predicate_index_(predicate_index), selection_(std::move(selection)),
      input_relation_is_stored_(input_relation_is_stored), on_gpu_(true) {
  if (selection_.size() != output_relation.size()) {
    throw ExecutionError("SelectOperator: one Scalar per output attribute", QSX_ERR_INVALID_ARGUMENT);
  }
  for (std::size_t i = 0; i < selection_.size(); ++i) {
    if (selection_[i] == nullptr) throw ExecutionError("SelectOperator: null Scalar", QSX_ERR_INVALID_ARGUMENT);
    // (an expression's output attribute must have its result type: checked per work order, against the block's relation)
  }
  if (input_relation_is_stored) input_relation_block_ids_ = input_relation.getBlocksSnapshot();
}

bool SelectOperator::getAllWorkOrders(WorkOrdersContainer *container, QueryContext *query_context,
                                      StorageManager *storage_manager, const tmb::client_id, tmb::MessageBus *) {
  // SelectOperator.cpp:47-107: one work order per input block; streaming inputs
  // generate incrementally and finish when done_feeding_input_relation_.
  const Predicate *predicate = query_context->getPredicate(predicate_index_);
  InsertDestination *dest = query_context->getInsertDestination(output_destination_index_);
  CheckRepartition("SelectOperator", has_repartition_, dest);
  std::lock_guard<std::mutex> lock(mutex_);
  while (num_workorders_generated_ < input_relation_block_ids_.size()) {
    // every block that has arrived, in runs of blocks_per_work_order_ (1: the reference's one work order per block)
    const std::size_t take = on_gpu_ ? std::min(blocks_per_work_order_, input_relation_block_ids_.size() - num_workorders_generated_) : 1;
    if (take > 1) {
      std::vector<block_id> run(input_relation_block_ids_.begin() + static_cast<std::ptrdiff_t>(num_workorders_generated_),
                                input_relation_block_ids_.begin() + static_cast<std::ptrdiff_t>(num_workorders_generated_ + take));
      container->addNormalWorkOrder(new SelectWorkOrder(query_id_, std::move(run), predicate, simple_selection_, dest, storage_manager,
                                                        CreateLIPFilterAdaptiveProberHelper(lip_deployment_index_, query_context),
                                                        selection_.empty() ? nullptr : &selection_),
                                    op_index_);
    } else {
      container->addNormalWorkOrder(new SelectWorkOrder(query_id_, input_relation_block_ids_[num_workorders_generated_],
                                                        predicate, simple_selection_, dest, storage_manager, on_gpu_,
                                                        CreateLIPFilterAdaptiveProberHelper(lip_deployment_index_, query_context),
                                                        selection_.empty() ? nullptr : &selection_),
                                    op_index_);
    }
    num_workorders_generated_ += take;
  }
  return input_relation_is_stored_ || done_feeding_input_relation_;
}

void SelectWorkOrder::execute() {
  if (!on_gpu_) {
    executeOnHost();
    return;
  }
  if (run_block_ids_.empty()) {
    executeBlock(input_block_id_);
    return;
  }
  if (executeRun()) return;
  for (block_id id : run_block_ids_) executeBlock(id);
}

// A run of blocks as one unit: every predicate term is one launch over all blocks (per-block bitmaps chained through the
// terms like a conjunction), then the selected tuples of the run, block after block, are compacted into ONE output block
// — what consecutive SelectWorkOrders do to an InsertDestination's current block (InsertDestination.cpp:222-260).
bool SelectWorkOrder::executeRun() {
  const bool has_terms = predicate_ != nullptr && !predicate_->conjuncts.empty();
  // (no predicate and no filter: a projection of every tuple — the repartitioning Select in front of a partitioned join,
  // ExecutionGenerator's build / probe side "needs repartition" — takes the run form too: one compaction under all-ones
  // TupleIdSequences instead of a block allocation, a copy and a wait per 4 MB block)
  const bool plain_copy = !has_terms && lip_filter_adaptive_prober_ == nullptr;
  static const Predicate no_terms;
  const Predicate &predicate = has_terms ? *predicate_ : no_terms;
  std::vector<attribute_id> selection;
  if (selection_ != nullptr && !selection_->empty()) {
    for (const ScalarPtr &scalar : *selection_) {
      if (scalar->kind != Scalar::kAttribute) return false;
      selection.push_back(scalar->attribute);
    }
  } else {
    selection = simple_selection_;
  }
  if (selection.size() > QSX_MAX_COLUMNS) return false;
  std::vector<BlockReference> blocks;
  std::vector<std::int64_t> rows;
  std::int64_t total_rows = 0;
  std::size_t bitmap_words = 0;
  for (block_id id : run_block_ids_) {
    blocks.push_back(storage_manager_->getBlock(id));
    const StorageBlock &b = *blocks.back();
    for (attribute_id a : selection) {
      if (b.nullBitmap(a) != nullptr) return false;   // (projected values of a compressed attribute: stripe() decodes once)
    }
    rows.push_back(b.numTuples());
    total_rows += b.numTuples();
    bitmap_words += static_cast<std::size_t>((b.numTuples() + 63) / 64) + 1;
  }
  if (!RunPredicateCovers(predicate, blocks)) return false;
  const std::size_t nb = blocks.size();
  // SelectOperator.cpp:161-195: predicate matches, then the LIP filters on what is left (a filter looks a tuple up at a random
  // place of its bit vector; a tuple the predicate dropped costs it nothing: Q3's lineitem Select loses 46 % of its lookups)
  struct OwnedStorage {
    void *ptr = nullptr;
    ~OwnedStorage() { qsx_device_free(ptr); }
  } lip_storage;
  std::vector<const std::uint64_t *> lip_bitmaps;
  std::int64_t lip_hits = 0;
  RunMatches run_matches;
  if (has_terms) RunPredicateMatches(predicate, blocks, rows, nullptr, &run_matches);
  const bool lip_after_predicate = has_terms && lip_filter_adaptive_prober_ != nullptr;
  if (lip_filter_adaptive_prober_ != nullptr &&
      !lip_filter_adaptive_prober_->filterBlocks(blocks, &lip_storage.ptr, &lip_bitmaps, &lip_hits,
                                                 has_terms ? reinterpret_cast<const std::uint64_t *const *>(run_matches.bitmaps.data()) : nullptr)) {
    return false;
  }
  std::int64_t matches = lip_hits;
  const std::uint64_t *const *selected = lip_bitmaps.empty() ? nullptr : lip_bitmaps.data();   // only LIP filters: their bitmaps
  // every tuple: one all-ones TupleIdSequence per distinct block size of the run (a relation's blocks hold the same number of
  // tuples but the last), trailing bits zero
  if (plain_copy && total_rows > 0) {
    // every tuple, projected on plain attributes: the selected stripes of the run laid end to end in the output block — one
    // qsx_copy_segments launch for all attributes (no TupleIdSequences, no compaction, no count to read back)
    // (into a partition-aware destination: K9 reads the run where it lies — no copy in front of the scatter)
    if (output_destination_->insertRunRepartitioned(blocks, selection)) return true;
    block_id copy_id;
    BlockReference copy = output_destination_->getBlockForInsertion(total_rows, &copy_id);
    std::vector<const void *> from;
    std::vector<void *> to;
    std::vector<std::int64_t> bytes;
    for (std::size_t i = 0; i < selection.size(); ++i) {
      const std::int64_t width = blocks.front()->getRelation().getAttributeType(selection[i]).width;
      std::int64_t at = 0;
      for (std::size_t b = 0; b < nb; ++b) {
        if (rows[b] == 0) continue;
        from.push_back(blocks[b]->stripe(selection[i]));
        to.push_back(static_cast<char *>(copy->stripe(static_cast<attribute_id>(i))) + at * width);
        bytes.push_back(rows[b] * width);
        at += rows[b];
      }
    }
    CheckStatus(qsx_copy_segments(static_cast<std::int64_t>(from.size()), from.data(), to.data(), bytes.data(), CurrentStream()), "qsx_copy_segments");
    CheckStatus(qsx_stream_synchronize(CurrentStream()), "qsx_stream_synchronize");   // (the work order's wait, like the reference's execute())
    output_destination_->returnBlock(copy_id, total_rows, getPartitionId());
    return true;
  }
  std::vector<std::unique_ptr<DeviceBuffer>> ones_storage;
  std::vector<const std::uint64_t *> ones_of_block;
  if (plain_copy) {
    std::vector<std::pair<std::int64_t, const std::uint64_t *>> made;
    for (std::size_t b = 0; b < nb; ++b) {
      const std::uint64_t *bits = nullptr;
      for (const auto &m : made) if (m.first == rows[b]) bits = m.second;
      if (bits == nullptr && rows[b] > 0) {
        const std::size_t words = static_cast<std::size_t>((rows[b] + 63) / 64) + 1;
        ones_storage.emplace_back(new DeviceBuffer(words * 8));
        DeviceBuffer zero(words * 8);
        CheckStatus(qsx_memset_device(zero.ptr, 0, words * 8, CurrentStream()), "qsx_memset_device");
        CheckStatus(qsx_bitmap_combine(3, static_cast<const std::uint64_t *>(zero.ptr), nullptr, rows[b], static_cast<std::uint64_t *>(ones_storage.back()->ptr),
                                       CurrentStream()), "qsx_bitmap_combine");
        CheckStatus(qsx_stream_synchronize(CurrentStream()), "qsx_stream_synchronize");   // (zero goes out of scope)
        bits = static_cast<const std::uint64_t *>(ones_storage.back()->ptr);
        made.emplace_back(rows[b], bits);
      }
      ones_of_block.push_back(bits != nullptr ? bits : static_cast<const std::uint64_t *>(nullptr));
    }
    // (an empty block of the run needs a non-null pointer: any of the others', or a word of its own)
    const std::uint64_t *any = nullptr;
    for (const std::uint64_t *p : ones_of_block) if (p != nullptr) any = p;
    if (any == nullptr) return false;
    for (const std::uint64_t *&p : ones_of_block) if (p == nullptr) p = any;
    selected = ones_of_block.data();
    matches = total_rows;
  }
  if (has_terms && !lip_after_predicate) {
    std::vector<std::int64_t> block_matches(nb);
    CheckStatus(qsx_copy_to_host(block_matches.data(), run_matches.counts->ptr, nb * 8, CurrentStream()), "qsx_copy_to_host");
    CheckStatus(qsx_stream_synchronize(CurrentStream()), "qsx_stream_synchronize");
    matches = 0;
    for (std::int64_t m : block_matches) matches += m;
    selected = reinterpret_cast<const std::uint64_t *const *>(run_matches.bitmaps.data());
  }
  block_id out_id;
  BlockReference out = output_destination_->getBlockForInsertion(matches > 0 ? matches : 1, &out_id);
  std::vector<const void *> src(nb * selection.size());
  std::vector<void *> dst;
  std::vector<std::int32_t> widths;
  for (std::size_t i = 0; i < selection.size(); ++i) {
    dst.push_back(out->stripe(static_cast<attribute_id>(i)));
    widths.push_back(blocks.front()->getRelation().getAttributeType(selection[i]).width);
    for (std::size_t b = 0; b < nb; ++b) src[b * selection.size() + i] = blocks[b]->stripe(selection[i]);
  }
  const std::size_t ws_bytes = qsx_compact_blocks_workspace_bytes(static_cast<std::int64_t>(nb), rows.data());
  DeviceBuffer ws(ws_bytes + 8), count(8);
  CheckStatus(qsx_compact_gather_blocks(static_cast<int>(selection.size()), widths.data(), static_cast<std::int64_t>(nb), rows.data(),
                                        src.data(), selected, nullptr, dst.data(),
                                        nullptr, static_cast<std::int64_t *>(count.ptr), ws.ptr, ws_bytes, CurrentStream()),
              "qsx_compact_gather_blocks");
  const std::int64_t written = ReadCount(count.ptr);   // synchronises the work order, like the reference's execute()
  output_destination_->returnBlock(out_id, written, getPartitionId());
  return true;
}

void SelectWorkOrder::executeBlock(block_id input_block_id) {
  BlockReference block = storage_manager_->getBlock(input_block_id);
  const std::int64_t n = block->numTuples();
  std::int64_t matches = 0;
  Predicate all;
  // SelectOperator.cpp:161-195: predicate matches, then the LIP filters on what is left
  void *bitmap = (predicate_ != nullptr ? predicate_ : &all)->getMatchesForBlock(*block, &matches, nullptr);  // getMatchesForPredicate
  if (lip_filter_adaptive_prober_ != nullptr) {
    void *filtered = lip_filter_adaptive_prober_->filterValueAccessor(*block, static_cast<const std::uint64_t *>(bitmap), &matches);
    qsx_device_free(bitmap);
    bitmap = filtered;
  }
  block_id out_id;
  BlockReference out = output_destination_->getBlockForInsertion(matches > 0 ? matches : 1, &out_id);
  // block->selectSimple(simple_selection_, matches, output_destination_) (StorageBlock.cpp:390-399), or block->select(
  // selection_, ...) (:363-388): every Scalar's values for the block, then the matching rows of each
  std::vector<const void *> src;
  std::vector<void *> dst;
  std::vector<std::int32_t> widths;
  std::vector<std::unique_ptr<DeviceBuffer>> expression_values;
  std::vector<attribute_id> null_sources;   // per output attribute: the input attribute whose null bitmap it inherits
  if (selection_ != nullptr && !selection_->empty()) {
    for (std::size_t i = 0; i < selection_->size(); ++i) {
      const ScalarPtr &scalar = (*selection_)[i];
      dst.push_back(out->stripe(static_cast<attribute_id>(i)));
      if (scalar->kind == Scalar::kAttribute) {
        src.push_back(block->stripe(scalar->attribute));
        widths.push_back(block->getRelation().getAttributeType(scalar->attribute).width);
        null_sources.push_back(scalar->attribute);
        continue;
      }
      // ScalarBinaryExpression / ScalarLiteral: one fused pass over the operand stripes (qsx_eval_expression)
      std::vector<attribute_id> attrs;
      ExpressionFlattener flattener([&](attribute_id a) {
        for (std::size_t c = 0; c < attrs.size(); ++c) if (attrs[c] == a) return static_cast<int>(c);
        attrs.push_back(a);
        return static_cast<int>(attrs.size() - 1);
      });
      const qsx_operand_t result = flattener.add(scalar);
      const void *cols[QSX_MAX_COLUMNS];
      std::int32_t types[QSX_MAX_COLUMNS];
      if (attrs.size() > QSX_MAX_COLUMNS) throw ExecutionError("SelectWorkOrder: expression over too many attributes", QSX_ERR_UNSUPPORTED);
      for (std::size_t c = 0; c < attrs.size(); ++c) {
        cols[c] = block->stripe(attrs[c]);
        types[c] = block->getRelation().getAttributeType(attrs[c]).id;
      }
      double consts[QSX_MAX_CONSTS] = {};
      for (std::size_t c = 0; c < flattener.consts().size(); ++c) consts[c] = flattener.consts()[c];
      expression_values.emplace_back(new DeviceBuffer(static_cast<std::size_t>(n > 0 ? n : 1) * 8));
      const TypeID value_type = ScalarResultType(scalar, block->getRelation());
      const int value_width = value_type == kInt ? 4 : 8;
      if (out->getRelation().getAttributeType(static_cast<attribute_id>(i)).width != value_width) {
        throw ExecutionError("SelectWorkOrder: the output attribute of an expression must have the expression's type "
                             "(INT op INT is an INT, with a LONG a LONG, with a FLOAT / DOUBLE a DOUBLE)", QSX_ERR_INVALID_ARGUMENT);
      }
      if (value_type == kDouble) {
        CheckStatus(qsx_eval_expression(static_cast<int>(attrs.size()), cols, types, static_cast<int>(flattener.instrs().size()),
                                        flattener.instrs().data(), consts, result, n, static_cast<double *>(expression_values.back()->ptr),
                                        CurrentStream()), "qsx_eval_expression");
      } else {
        // integer operands: integer arithmetic (ArithmeticBinaryOperators.hpp:203-340 for INT / LONG)
        std::int64_t int_consts[QSX_MAX_CONSTS] = {};
        for (std::size_t c = 0; c < flattener.consts().size(); ++c) int_consts[c] = static_cast<std::int64_t>(flattener.consts()[c]);
        CheckStatus(qsx_eval_expression_long(static_cast<int>(attrs.size()), cols, types, static_cast<int>(flattener.instrs().size()),
                                             flattener.instrs().data(), int_consts, result, n, value_width, expression_values.back()->ptr,
                                             CurrentStream()), "qsx_eval_expression_long");
      }
      src.push_back(expression_values.back()->ptr);
      widths.push_back(value_width);
      null_sources.push_back(kInvalidAttributeID);
    }
  } else {
    for (std::size_t i = 0; i < simple_selection_.size(); ++i) {
      src.push_back(block->stripe(simple_selection_[i]));
      dst.push_back(out->stripe(static_cast<attribute_id>(i)));
      widths.push_back(block->getRelation().getAttributeType(simple_selection_[i]).width);
      null_sources.push_back(simple_selection_[i]);
    }
  }
  const std::size_t ws_bytes = qsx_compact_workspace_bytes(n);
  DeviceBuffer ws(ws_bytes), count(8);
  CheckStatus(qsx_compact_gather(static_cast<int>(src.size()), src.data(), widths.data(),
                                 static_cast<const std::uint64_t *>(bitmap), n, dst.data(),
                                 static_cast<std::int64_t *>(count.ptr), ws.ptr, ws_bytes, CurrentStream()),
              "qsx_compact_gather");
  const std::int64_t written = ReadCount(count.ptr);  // synchronises the work order, like the reference's execute()
  ProjectNullBitmaps(*block, null_sources, bitmap, written, out.get());
  qsx_device_free(bitmap);
  output_destination_->returnBlock(out_id, written, getPartitionId());
}

// CPU work order of BASELINE config 1: the same plumbing with the loops on the
// host (blocks in host memory).  Not a fallback: only built by operators that
// were constructed with on_gpu = false.
void SelectWorkOrder::executeOnHost() {
  BlockReference block = storage_manager_->getBlock(input_block_id_);
  const std::int64_t n = block->numTuples();
  std::vector<bool> match(static_cast<std::size_t>(n), true);
  if (predicate_ != nullptr) {
    for (const ComparisonPredicate &term : predicate_->conjuncts) {
      const Type &t = block->getRelation().getAttributeType(term.attribute);
      const char *base = static_cast<const char *>(block->stripe(term.attribute));
      for (std::int64_t i = 0; i < n; ++i) {
        if (!match[i]) continue;
        double a, b;
        std::int64_t ia = 0, ib = 0;
        bool is_int = true;
        switch (t.id) {
          case kInt: ia = reinterpret_cast<const std::int32_t *>(base)[i]; ib = term.literal.v.i32; break;
          case kLong: ia = reinterpret_cast<const std::int64_t *>(base)[i]; ib = term.literal.v.i64; break;
          case kFloat: is_int = false; a = reinterpret_cast<const float *>(base)[i]; b = term.literal.v.f32; break;
          default: is_int = false; a = reinterpret_cast<const double *>(base)[i]; b = term.literal.v.f64; break;
        }
        bool r;
        switch (term.comparison) {
          case ComparisonID::kEqual: r = is_int ? ia == ib : a == b; break;
          case ComparisonID::kNotEqual: r = is_int ? ia != ib : a != b; break;
          case ComparisonID::kLess: r = is_int ? ia < ib : a < b; break;
          case ComparisonID::kLessOrEqual: r = is_int ? ia <= ib : a <= b; break;
          case ComparisonID::kGreater: r = is_int ? ia > ib : a > b; break;
          default: r = is_int ? ia >= ib : a >= b; break;
        }
        match[i] = r;
      }
    }
  }
  std::int64_t matches = 0;
  for (std::int64_t i = 0; i < n; ++i) matches += match[i];
  block_id out_id;
  BlockReference out = output_destination_->getBlockForInsertion(matches > 0 ? matches : 1, &out_id);
  for (std::size_t c = 0; c < simple_selection_.size(); ++c) {
    const int w = block->getRelation().getAttributeType(simple_selection_[c]).width;
    const char *src = static_cast<const char *>(block->stripe(simple_selection_[c]));
    char *dst = static_cast<char *>(out->stripe(static_cast<attribute_id>(c)));
    std::int64_t o = 0;
    for (std::int64_t i = 0; i < n; ++i) {
      if (match[i]) std::memcpy(dst + (o++) * w, src + i * w, w);
    }
  }
  output_destination_->returnBlock(out_id, matches, getPartitionId());
}

// ---------------------------------------------------------------------------
// BuildHash
// ---------------------------------------------------------------------------
BuildHashOperator::BuildHashOperator(std::size_t query_id, const CatalogRelation &input_relation,
                                     bool input_relation_is_stored, const std::vector<attribute_id> &join_key_attributes,
                                     bool, std::size_t num_partitions, QueryContext::join_hash_table_id hash_table_index,
                                     QueryContext::predicate_id build_predicate_index)
    : RelationalOperator(query_id, num_partitions), input_relation_(input_relation),
      input_relation_is_stored_(input_relation_is_stored), join_key_attributes_(join_key_attributes),
      is_broadcast_join_(num_partitions > 1u && !input_relation.hasPartitionScheme()),
      hash_table_index_(hash_table_index), build_predicate_index_(build_predicate_index), input_(num_partitions) {
  if (join_key_attributes.empty() || join_key_attributes.size() > QSX_MAX_KEYS) {
    throw ExecutionError("BuildHashOperator: 1 to 4 INT/LONG join key attributes are on the GPU path", QSX_ERR_UNSUPPORTED);
  }
  if (input_relation_is_stored) {
    // BuildHashOperator.hpp:105-119: per-partition blocks, or every block for every partition (broadcast)
    if (input_relation.hasPartitionScheme() && input_relation.getNumPartitions() != num_partitions) {
      throw ExecutionError("BuildHashOperator: num_partitions differs from the input relation's partition scheme", QSX_ERR_INVALID_ARGUMENT);
    }
    for (partition_id part = 0; part < num_partitions; ++part) {
      input_.ids[part] = is_broadcast_join_ ? input_relation.getBlocksSnapshot() : input_relation.getBlocksInPartition(part);
    }
  }
}

bool BuildHashOperator::getAllWorkOrders(WorkOrdersContainer *container, QueryContext *query_context,
                                         StorageManager *storage_manager, const tmb::client_id, tmb::MessageBus *) {
  const Predicate *predicate = query_context->getPredicate(build_predicate_index_);
  std::lock_guard<std::mutex> lock(mutex_);
  if (!started_) {
    query_context->setJoinHashTableBuildKeyAttributes(hash_table_index_, join_key_attributes_);
    started_ = true;
  }
  for (partition_id part = 0; part < num_partitions_; ++part) {   // BuildHashOperator.cpp:82-110
    qsx_join_table_t *table = query_context->getJoinHashTable(hash_table_index_, part);
    while (input_.generated[part] < input_.ids[part].size()) {
      const std::size_t take = std::min(blocks_per_work_order_, input_.ids[part].size() - input_.generated[part]);
      BuildHashWorkOrder *order = new BuildHashWorkOrder(query_id_, input_relation_, join_key_attributes_,
                                                         input_.ids[part][input_.generated[part]], predicate, table,
                                                         storage_manager, part,
                                                         CreateLIPFilterBuilderHelper(lip_deployment_index_, query_context));
      if (take > 1 || blocks_per_work_order_ > 1) {   // (run mode: a lone block takes the run form too — its key stripe as it lies)
        order->setRun(std::vector<block_id>(input_.ids[part].begin() + static_cast<std::ptrdiff_t>(input_.generated[part]),
                                            input_.ids[part].begin() + static_cast<std::ptrdiff_t>(input_.generated[part] + take)));
      }
      container->addNormalWorkOrder(order, op_index_);
      input_.generated[part] += take;
    }
  }
  return input_relation_is_stored_ || done_feeding_input_relation_;
}


void BuildHashWorkOrder::execute() {
  if (run_block_ids_.empty()) {
    executeBlock(build_block_id_);
    return;
  }
  if (executeRun()) return;
  for (block_id id : run_block_ids_) executeBlock(id);
}

bool BuildHashWorkOrder::executeRun() {
  if (predicate_ != nullptr) return false;
  std::vector<BlockReference> blocks;
  std::vector<std::int64_t> rows;
  std::vector<std::int32_t> bases;
  for (block_id id : run_block_ids_) {
    blocks.push_back(storage_manager_->getBlock(id));
    const StorageBlock &b = *blocks.back();
    for (attribute_id a : join_key_attributes_) {
      if (b.nullBitmap(a) != nullptr) return false;   // (compressed key: read as it lies, RunJoinKeys)
    }
    if (b.firstRow() + b.numTuples() > INT32_MAX) return false;
    rows.push_back(b.numTuples());
    bases.push_back(static_cast<std::int32_t>(b.firstRow()));   // the stored reference: relation-global row number
  }
  if (lip_filter_builder_ != nullptr && !lip_filter_builder_->coversBlocks(blocks)) return false;
  const RunJoinKeys run_keys(blocks, join_key_attributes_, rows);
  const std::vector<const void *> &keys = run_keys.ptr;
  CheckStatus(qsx_join_build_blocks_coded(hash_table_, static_cast<std::int64_t>(blocks.size()), rows.data(), keys.data(), run_keys.coding(),
                                          bases.data(), nullptr, CurrentStream()), "qsx_join_build_blocks");
  // BuildHashOperator.cpp:187-190 builds the LIP filters in front of the table; behind it (same stream) a filter over the join
  // key can read its bits off the table
  if (lip_filter_builder_ != nullptr) {
    lip_filter_builder_->insertBlocks(blocks, hash_table_, join_key_attributes_.size() == 1 ? join_key_attributes_.front() : kInvalidAttributeID);
  }
  CheckStatus(qsx_stream_synchronize(CurrentStream()), "qsx_stream_synchronize");
  return true;
}

void BuildHashWorkOrder::executeBlock(block_id build_block_id) {
  BlockReference block = storage_manager_->getBlock(build_block_id);
  void *bitmap = nullptr;
  if (predicate_ != nullptr) {
    std::int64_t matches = 0;
    bitmap = predicate_->getMatchesForBlock(*block, &matches);
  }
  // hash_table_->putValueAccessor[CompositeKey](accessor, key_attr(s), nullable, &TupleReferenceGenerator) (:192-203);
  // the stored reference is the relation-global row number of the tuple.
  if (lip_filter_builder_ != nullptr) {
    lip_filter_builder_->insertValueAccessor(*block, static_cast<const std::uint64_t *>(bitmap));  // :187-190
  }
  JoinKeys keys(*block, join_key_attributes_);
  // check_for_null_keys: tuples with a NULL key component are not inserted (HashTable.hpp:1409-1418, 1505-1518)
  std::unique_ptr<DeviceBuffer> not_null = NotNullFilter(*block, join_key_attributes_, static_cast<const std::uint64_t *>(bitmap));
  const std::uint64_t *build_filter = not_null != nullptr ? static_cast<const std::uint64_t *>(not_null->ptr)
                                                          : static_cast<const std::uint64_t *>(bitmap);
  CheckStatus(qsx_join_build(hash_table_, keys.ptr, block->numTuples(),
                             static_cast<std::int32_t>(block->firstRow()), build_filter,
                             CurrentStream()), "qsx_join_build");
  CheckStatus(qsx_stream_synchronize(CurrentStream()), "qsx_stream_synchronize");
  qsx_device_free(bitmap);
}


}  // namespace quickstep
