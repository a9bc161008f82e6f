// aggregation_sort_operators.cpp — Aggregation / FinalizeAggregation / sort operators and their work orders (see quickstep_gpu.hpp; what the files share: quickstep_gpu_internal.hpp)
#include "quickstep_gpu_internal.hpp"

namespace quickstep {

namespace {
class AggregationWorkOrder : public WorkOrder {
 public:
  AggregationWorkOrder(std::size_t query_id, block_id input_block_id, AggregationOperationState *state,
                       StorageManager *storage_manager, partition_id part = 0, LIPFilterAdaptiveProber *prober = nullptr)
      : WorkOrder(query_id, part), input_block_id_(input_block_id), state_(state), storage_manager_(storage_manager),
        lip_filter_adaptive_prober_(prober) {}
  // a run of blocks (AggregationOperator::setBlocksPerWorkOrder)
  AggregationWorkOrder(std::size_t query_id, std::vector<block_id> input_block_ids, AggregationOperationState *state,
                       StorageManager *storage_manager, partition_id part, LIPFilterAdaptiveProber *prober)
      : WorkOrder(query_id, part), input_block_id_(input_block_ids.front()), more_block_ids_(input_block_ids.begin() + 1, input_block_ids.end()),
        state_(state), storage_manager_(storage_manager), lip_filter_adaptive_prober_(prober) {}
  void execute() override {  // AggregationOperator.cpp:124-126
    if (!more_block_ids_.empty()) {
      std::vector<BlockReference> blocks{storage_manager_->getBlock(input_block_id_)};
      for (block_id id : more_block_ids_) blocks.push_back(storage_manager_->getBlock(id));
      std::vector<const std::uint64_t *> filters(blocks.size(), nullptr);
      std::vector<void *> owned;
      if (lip_filter_adaptive_prober_ != nullptr) {
        void *storage = nullptr;
        if (lip_filter_adaptive_prober_->filterBlocks(blocks, &storage, &filters)) {   // one launch per filter over the run
          owned.push_back(storage);
        } else {
          filters.assign(blocks.size(), nullptr);
          for (std::size_t i = 0; i < blocks.size(); ++i) {
            owned.push_back(lip_filter_adaptive_prober_->filterValueAccessor(*blocks[i], nullptr, nullptr));
            filters[i] = static_cast<const std::uint64_t *>(owned.back());
          }
        }
      }
      state_->aggregateBlocks(blocks, filters);
      for (void *p : owned) qsx_device_free(p);
      return;
    }
    BlockReference block = storage_manager_->getBlock(input_block_id_);
    void *lip = nullptr;
    if (lip_filter_adaptive_prober_ != nullptr) lip = lip_filter_adaptive_prober_->filterValueAccessor(*block, nullptr, nullptr);
    state_->aggregateBlock(*block, static_cast<const std::uint64_t *>(lip));
    qsx_device_free(lip);
  }
 private:
  block_id input_block_id_;
  std::vector<block_id> more_block_ids_;
  AggregationOperationState *state_;
  StorageManager *storage_manager_;
  std::unique_ptr<LIPFilterAdaptiveProber> lip_filter_adaptive_prober_;
};
class BuildAggregationExistenceMapWorkOrder : public WorkOrder {
 public:
  BuildAggregationExistenceMapWorkOrder(std::size_t query_id, const CatalogRelation &input_relation, partition_id part,
                                        block_id build_block_id, attribute_id build_attribute, AggregationOperationState *state,
                                        StorageManager *storage_manager)
      : WorkOrder(query_id, part), input_relation_(input_relation), build_block_id_(build_block_id),
        build_attribute_(build_attribute), state_(state), storage_manager_(storage_manager) {}
  void execute() override {   // BuildAggregationExistenceMapOperator.cpp:177-208
    BlockReference block = storage_manager_->getBlock(build_block_id_);
    state_->buildExistenceMap(*block, build_attribute_, input_relation_.getAttributeType(build_attribute_));
  }
 private:
  const CatalogRelation &input_relation_;
  block_id build_block_id_;
  attribute_id build_attribute_;
  AggregationOperationState *state_;
  StorageManager *storage_manager_;
};
class FinalizeAggregationWorkOrder : public WorkOrder {
 public:
  FinalizeAggregationWorkOrder(std::size_t query_id, std::size_t part, std::size_t num_parts,
                               AggregationOperationState *state, InsertDestination *dest)
      : WorkOrder(query_id), part_(part), num_parts_(num_parts), state_(state), dest_(dest) {}
  void execute() override { state_->finalizeAggregate(part_, num_parts_, dest_); }  // FinalizeAggregationOperator.cpp:99-101
 private:
  std::size_t part_, num_parts_;
  AggregationOperationState *state_;
  InsertDestination *dest_;
};
class InitializeAggregationWorkOrder : public WorkOrder {
 public:
  InitializeAggregationWorkOrder(std::size_t query_id, partition_id part, std::size_t state_partition_id, AggregationOperationState *state)
      : WorkOrder(query_id, part), state_partition_id_(state_partition_id), state_(state) {}
  void execute() override { state_->initialize(state_partition_id_); }   // InitializeAggregationOperator.cpp:91-93
 private:
  const std::size_t state_partition_id_;
  AggregationOperationState *state_;
};
class DestroyAggregationStateWorkOrder : public WorkOrder {
 public:
  DestroyAggregationStateWorkOrder(std::size_t query_id, QueryContext::aggregation_state_id id, QueryContext *ctx,
                                   partition_id part)
      : WorkOrder(query_id, part), id_(id), ctx_(ctx) {}
  void execute() override { ctx_->destroyAggregationState(id_, partition_id_); }
 private:
  QueryContext::aggregation_state_id id_;
  QueryContext *ctx_;
};
}  // namespace
// ---------------------------------------------------------------------------
// Aggregation
// ---------------------------------------------------------------------------
AggregationOperator::AggregationOperator(std::size_t query_id, const CatalogRelation &input_relation,
                                         bool input_relation_is_stored, QueryContext::aggregation_state_id aggr_state_index,
                                         std::size_t num_partitions)
    : RelationalOperator(query_id, num_partitions), input_relation_(input_relation),
      input_relation_is_stored_(input_relation_is_stored), aggr_state_index_(aggr_state_index), input_(num_partitions) {
  if (input_relation_is_stored) {
    if (num_partitions > 1 && input_relation.getNumPartitions() != num_partitions) {
      throw ExecutionError("AggregationOperator: num_partitions differs from the input relation's partition scheme", QSX_ERR_INVALID_ARGUMENT);
    }
    for (partition_id part = 0; part < num_partitions; ++part) {
      input_.ids[part] = num_partitions > 1 ? input_relation.getBlocksInPartition(part) : input_relation.getBlocksSnapshot();
    }
  }
}

bool AggregationOperator::getAllWorkOrders(WorkOrdersContainer *container, QueryContext *query_context,
                                           StorageManager *storage_manager, const tmb::client_id, tmb::MessageBus *) {
  std::lock_guard<std::mutex> lock(mutex_);
  for (partition_id part = 0; part < num_partitions_; ++part) {   // AggregationOperator.cpp:49-61
    AggregationOperationState *state = query_context->getAggregationState(aggr_state_index_, part);
    while (input_.generated[part] < input_.ids[part].size()) {
      // every block that has arrived, in runs of blocks_per_work_order_ (1: the reference's one work order per block)
      const std::size_t take = std::min(blocks_per_work_order_, input_.ids[part].size() - input_.generated[part]);
      if (take > 1) {
        std::vector<block_id> run(input_.ids[part].begin() + static_cast<std::ptrdiff_t>(input_.generated[part]),
                                  input_.ids[part].begin() + static_cast<std::ptrdiff_t>(input_.generated[part] + take));
        container->addNormalWorkOrder(new AggregationWorkOrder(query_id_, std::move(run), state, storage_manager, part,
                                                               CreateLIPFilterAdaptiveProberHelper(lip_deployment_index_, query_context)),
                                      op_index_);
      } else {
        container->addNormalWorkOrder(new AggregationWorkOrder(query_id_, input_.ids[part][input_.generated[part]], state,
                                                               storage_manager, part,
                                                               CreateLIPFilterAdaptiveProberHelper(lip_deployment_index_, query_context)),
                                      op_index_);
      }
      input_.generated[part] += take;
    }
  }
  return input_relation_is_stored_ || done_feeding_input_relation_;
}

BuildAggregationExistenceMapOperator::BuildAggregationExistenceMapOperator(
    std::size_t query_id, const CatalogRelation &input_relation, attribute_id build_attribute, bool input_relation_is_stored,
    QueryContext::aggregation_state_id aggr_state_index, std::size_t num_partitions)
    : RelationalOperator(query_id, num_partitions), input_relation_(input_relation), build_attribute_(build_attribute),
      input_relation_is_stored_(input_relation_is_stored), aggr_state_index_(aggr_state_index), input_(num_partitions) {
  if (input_relation_is_stored) {
    for (partition_id part = 0; part < num_partitions; ++part) {
      input_.ids[part] = num_partitions > 1 ? input_relation.getBlocksInPartition(part) : input_relation.getBlocksSnapshot();
    }
  }
}

bool BuildAggregationExistenceMapOperator::getAllWorkOrders(WorkOrdersContainer *container, QueryContext *query_context,
                                                            StorageManager *storage_manager, const tmb::client_id,
                                                            tmb::MessageBus *) {
  std::lock_guard<std::mutex> lock(mutex_);
  for (partition_id part = 0; part < num_partitions_; ++part) {   // BuildAggregationExistenceMapOperator.cpp:82-128
    AggregationOperationState *state = query_context->getAggregationState(aggr_state_index_, part);
    while (input_.generated[part] < input_.ids[part].size()) {
      container->addNormalWorkOrder(new BuildAggregationExistenceMapWorkOrder(query_id_, input_relation_, part,
                                                                              input_.ids[part][input_.generated[part]],
                                                                              build_attribute_, state, storage_manager),
                                    op_index_);
      ++input_.generated[part];
    }
  }
  return input_relation_is_stored_ || done_feeding_input_relation_;
}

bool FinalizeAggregationOperator::getAllWorkOrders(WorkOrdersContainer *container, QueryContext *query_context,
                                                   StorageManager *, const tmb::client_id, tmb::MessageBus *) {
  if (!started_) {
    started_ = true;
    InsertDestination *dest = query_context->getInsertDestination(output_destination_index_);
    CheckRepartition("FinalizeAggregationOperator", has_repartition_, dest);
    // num_partitions x aggr_state_num_partitions work orders (FinalizeAggregationOperator.cpp:48-66)
    for (partition_id part = 0; part < num_partitions_; ++part) {
      AggregationOperationState *state = query_context->getAggregationState(aggr_state_index_, part);
      for (std::size_t p = 0; p < aggr_state_num_partitions_; ++p) {
        if (rank_slice_ >= 0 && p != static_cast<std::size_t>(rank_slice_)) continue;   // the other slices are other ranks'
        container->addNormalWorkOrder(new FinalizeAggregationWorkOrder(query_id_, p, aggr_state_num_partitions_, state, dest),
                                      op_index_);
      }
    }
  }
  return true;
}

bool InitializeAggregationOperator::getAllWorkOrders(WorkOrdersContainer *container, QueryContext *query_context, StorageManager *,
                                                     const tmb::client_id, tmb::MessageBus *) {
  if (started_) return true;
  for (partition_id part = 0; part < num_partitions_; ++part) {
    AggregationOperationState *state = query_context->getAggregationState(aggr_state_index_, part);
    for (std::size_t p = 0; p < aggr_state_num_init_partitions_; ++p) {
      container->addNormalWorkOrder(new InitializeAggregationWorkOrder(query_id_, part, p, state), op_index_);
    }
  }
  started_ = true;
  return true;
}

bool DestroyAggregationStateOperator::getAllWorkOrders(WorkOrdersContainer *container, QueryContext *query_context,
                                                       StorageManager *, const tmb::client_id, tmb::MessageBus *) {
  if (!work_generated_) {
    work_generated_ = true;
    for (partition_id part = 0; part < num_partitions_; ++part) {
      container->addNormalWorkOrder(new DestroyAggregationStateWorkOrder(query_id_, aggr_state_index_, query_context, part), op_index_);
    }
  }
  return true;
}

// ---------------------------------------------------------------------------
// ORDER BY
// ---------------------------------------------------------------------------
namespace {
// Sorts the concatenation of `blocks` by `config` and writes the first `limit` tuples (0 = all) into one output block.
void SortBlocksInto(const std::vector<BlockReference> &blocks, const CatalogRelation &relation,
                    const QueryContext::SortConfiguration &config, std::size_t limit, InsertDestination *dest) {
  std::int64_t n = 0;
  for (const BlockReference &b : blocks) n += b->numTuples();
  const std::int64_t out_rows = limit != 0 && static_cast<std::int64_t>(limit) < n ? static_cast<std::int64_t>(limit) : n;
  block_id out_id;
  BlockReference out = dest->getBlockForInsertion(out_rows > 0 ? out_rows : 1, &out_id);
  if (n == 0) {
    dest->returnBlock(out_id, 0);
    return;
  }
  // one contiguous stripe per attribute (a single input block is used in place)
  std::vector<std::unique_ptr<DeviceBuffer>> owned;
  std::vector<const void *> stripes(relation.size(), nullptr);
  for (std::size_t a = 0; a < relation.size(); ++a) {
    const int width = relation.getAttributeType(static_cast<attribute_id>(a)).width;
    if (blocks.size() == 1) {
      stripes[a] = blocks.front()->stripe(static_cast<attribute_id>(a));
      continue;
    }
    owned.emplace_back(new DeviceBuffer(static_cast<std::size_t>(n) * width + 16));
    char *at = static_cast<char *>(owned.back()->ptr);
    for (const BlockReference &b : blocks) {
      const std::size_t bytes = static_cast<std::size_t>(b->numTuples()) * width;
      CheckStatus(qsx_copy_on_device(at, b->stripe(static_cast<attribute_id>(a)), bytes, CurrentStream()), "qsx_copy_on_device");
      at += bytes;
    }
    stripes[a] = owned.back()->ptr;
  }
  std::vector<const void *> key_cols;
  std::vector<std::int32_t> key_types, descending;
  for (std::size_t k = 0; k < config.order_by.size(); ++k) {
    key_cols.push_back(stripes.at(config.order_by[k]));
    key_types.push_back(relation.getAttributeType(config.order_by[k]).id);
    descending.push_back(config.ordering.at(k) ? 0 : 1);
  }
  const std::size_t ws_bytes = qsx_sort_workspace_bytes(n);
  DeviceBuffer ws(ws_bytes), tids(static_cast<std::size_t>(n) * 4 + 16);
  if (out_rows < n) {   // top_k: only the leading rows are wanted
    CheckStatus(qsx_sort_top_k(static_cast<int>(key_cols.size()), key_cols.data(), key_types.data(), descending.data(), n, out_rows,
                               static_cast<std::int32_t *>(tids.ptr), ws.ptr, ws_bytes, CurrentStream()),
                "qsx_sort_top_k");
  } else {
    CheckStatus(qsx_sort_permutation(static_cast<int>(key_cols.size()), key_cols.data(), key_types.data(), descending.data(), n,
                                     static_cast<std::int32_t *>(tids.ptr), ws.ptr, ws_bytes, CurrentStream()),
                "qsx_sort_permutation");
  }
  for (std::size_t a = 0; a < relation.size(); ++a) {
    CheckStatus(qsx_gather(relation.getAttributeType(static_cast<attribute_id>(a)).width, stripes[a],
                           static_cast<const std::int32_t *>(tids.ptr), out_rows, out->stripe(static_cast<attribute_id>(a)),
                           CurrentStream()), "qsx_gather");
  }
  CheckStatus(qsx_stream_synchronize(CurrentStream()), "qsx_stream_synchronize");
  dest->returnBlock(out_id, out_rows);
}

class SortWorkOrder : public WorkOrder {
 public:
  SortWorkOrder(std::size_t query_id, std::vector<block_id> blocks, const CatalogRelation &relation,
                const QueryContext::SortConfiguration &config, std::size_t limit, InsertDestination *dest,
                StorageManager *storage_manager)
      : WorkOrder(query_id), blocks_(std::move(blocks)), relation_(relation), config_(config), limit_(limit), dest_(dest),
        storage_manager_(storage_manager) {}
  void execute() override {   // SortRunGenerationOperator.cpp:88-105 / SortMergeRunOperator.cpp:150-200
    std::vector<BlockReference> refs;
    for (block_id b : blocks_) refs.push_back(storage_manager_->getBlock(b));
    SortBlocksInto(refs, relation_, config_, limit_, dest_);
  }
 private:
  std::vector<block_id> blocks_;
  const CatalogRelation &relation_;
  const QueryContext::SortConfiguration &config_;
  const std::size_t limit_;
  InsertDestination *dest_;
  StorageManager *storage_manager_;
};
}  // namespace

bool SortRunGenerationOperator::getAllWorkOrders(WorkOrdersContainer *container, QueryContext *query_context,
                                                 StorageManager *storage_manager, const tmb::client_id, tmb::MessageBus *) {
  const QueryContext::SortConfiguration &config = query_context->getSortConfig(sort_config_index_);
  InsertDestination *dest = query_context->getInsertDestination(output_destination_index_);
  std::lock_guard<std::mutex> lock(mutex_);
  while (num_workorders_generated_ < input_relation_block_ids_.size()) {   // one sorted run per input block
    container->addNormalWorkOrder(new SortWorkOrder(query_id_, {input_relation_block_ids_[num_workorders_generated_]}, input_relation_,
                                                    config, top_k_, dest, storage_manager), op_index_);
    ++num_workorders_generated_;
  }
  return input_relation_is_stored_ || done_feeding_input_relation_;
}

bool SortMergeRunOperator::getAllWorkOrders(WorkOrdersContainer *container, QueryContext *query_context,
                                            StorageManager *storage_manager, const tmb::client_id, tmb::MessageBus *) {
  std::lock_guard<std::mutex> lock(mutex_);
  if (!input_relation_is_stored_ && !done_feeding_input_relation_) return false;   // every run must have arrived
  if (!work_generated_) {
    work_generated_ = true;
    container->addNormalWorkOrder(new SortWorkOrder(query_id_, input_relation_block_ids_, input_relation_,
                                                    query_context->getSortConfig(sort_config_index_), top_k_,
                                                    query_context->getInsertDestination(output_destination_index_), storage_manager),
                                  op_index_);
  }
  return true;
}


}  // namespace quickstep
