// rank_exchange.cpp — one process per GPU: RankGroup, PartitionExchangeOperator, ExchangeAggregationStatesOperator and
// AggregationOperationState::mergeAcrossRanks (see quickstep_gpu.hpp "Multi-GPU"; what the files share:
// quickstep_gpu_internal.hpp).  Everything that crosses a GPU boundary goes through the C ABI's multi-GPU entry points
// (qsx_exchange_counts, qsx_alltoallv, qsx_allgather, qsx_agg_reduce_scatter, qsx_agg_allgather_merge).
#include "quickstep_gpu_internal.hpp"

namespace quickstep {

using namespace host_internal;

std::vector<unsigned char> RankGroup::MakeUniqueId() {
  std::vector<unsigned char> id(QSX_COMM_ID_BYTES);
  CheckStatus(qsx_comm_unique_id(id.data()), "qsx_comm_unique_id");
  return id;
}
RankGroup::RankGroup(int world, int rank, const void *id_bytes) : world_(world), rank_(rank) {
  CheckStatus(qsx_comm_create(world, rank, id_bytes, &comm_), "qsx_comm_create");
}
RankGroup::~RankGroup() { qsx_comm_destroy(comm_); }

void RankGroup::agreeOn(const std::function<void()> &prepare, const char *where) {
  std::exception_ptr mine;
  int status = QSX_OK;
  try {
    prepare();
  } catch (const ExecutionError &e) {
    mine = std::current_exception();
    status = e.status() != QSX_OK ? e.status() : QSX_ERR_INVALID_ARGUMENT;
  } catch (const std::bad_alloc &) {
    mine = std::current_exception();
    status = QSX_ERR_OUT_OF_MEMORY;
  } catch (...) {
    mine = std::current_exception();
    status = QSX_ERR_INVALID_ARGUMENT;
  }
  const int verdict = qsx_comm_agree(comm_, status, CurrentStream());
  if (mine != nullptr) std::rethrow_exception(mine);
  CheckStatus(verdict, where);
}
void RankGroup::synchronize() { CheckStatus(qsx_comm_synchronize(comm_, CurrentStream()), "qsx_comm_synchronize"); }
void RankGroup::abort() noexcept { (void)qsx_comm_abort(comm_); }

void AggregationOperationState::mergeAcrossRanks(qsx_comm_t *comm) {
  // The local part first, and its outcome agreed on with the peers (qsx_comm_agree): a rank that fails here must not leave
  // the others inside the merge's collectives.  (The collectives agree on their own scratch the same way, csrc/aggregate.hip.)
  int status = QSX_OK;
  std::exception_ptr mine;
  try {
    if (!distinctify_.empty()) {
      throw ExecutionError("AggregationOperationState::mergeAcrossRanks: DISTINCT aggregates are not merged across ranks", QSX_ERR_UNSUPPORTED);
    }
    // the state fed by compressed blocks joins the other one first (as finalizeAggregate does)
    std::lock_guard<std::mutex> lock(coded_mutex_);
    if (coded_state_ != nullptr && !coded_merged_) {
      CheckStatus(qsx_agg_merge(state_, coded_state_, CurrentStream()), "qsx_agg_merge");
      CheckStatus(qsx_stream_synchronize(CurrentStream()), "qsx_stream_synchronize");
    }
    coded_merged_ = true;
  } catch (const ExecutionError &e) {
    mine = std::current_exception();
    status = e.status() != QSX_OK ? e.status() : QSX_ERR_INVALID_ARGUMENT;
  } catch (...) {
    mine = std::current_exception();
    status = QSX_ERR_INVALID_ARGUMENT;
  }
  const int verdict = qsx_comm_agree(comm, status, CurrentStream());
  if (mine != nullptr) std::rethrow_exception(mine);
  CheckStatus(verdict, "AggregationOperationState::mergeAcrossRanks");
  if (config_.strategy == QSX_AGG_COLLISION_FREE) {
    CheckStatus(qsx_agg_reduce_scatter(comm, state_, CurrentStream()), "qsx_agg_reduce_scatter");
  } else {
    CheckStatus(qsx_agg_allgather_merge(comm, state_, CurrentStream()), "qsx_agg_allgather_merge");
  }
  CheckStatus(qsx_comm_synchronize(comm, CurrentStream()), "qsx_comm_synchronize");
}

namespace {
std::int64_t Sum(const std::vector<std::int64_t> &v) {
  std::int64_t s = 0;
  for (std::int64_t x : v) s += x;
  return s;
}
std::int64_t WordsOf(std::int64_t rows) { return (rows + 63) / 64; }

// Row numbers 0 .. n-1 on the device (the tuple ids of an all-ones TupleIdSequence): the tid list the segmented bitmap
// gather takes to re-pack word-aligned null bitmaps into one.
std::unique_ptr<DeviceBuffer> RowNumbers(std::int64_t n) {
  std::unique_ptr<DeviceBuffer> rows(new DeviceBuffer(static_cast<std::size_t>(n) * 4 + 8));
  if (n == 0) return rows;
  const std::size_t words = static_cast<std::size_t>(WordsOf(n)) + 1;
  DeviceBuffer zero(words * 8), ones(words * 8), count(8);
  CheckStatus(qsx_memset_device(zero.ptr, 0, words * 8, CurrentStream()), "qsx_memset_device");
  CheckStatus(qsx_bitmap_combine(3, static_cast<const std::uint64_t *>(zero.ptr), nullptr, n, static_cast<std::uint64_t *>(ones.ptr), CurrentStream()),
              "qsx_bitmap_combine");
  const std::size_t tws = qsx_compact_workspace_bytes(n);
  DeviceBuffer tw(tws + 8);
  CheckStatus(qsx_bitmap_to_tids(static_cast<const std::uint64_t *>(ones.ptr), n, 0, static_cast<std::int32_t *>(rows->ptr),
                                 static_cast<std::int64_t *>(count.ptr), tw.ptr, tws, CurrentStream()), "qsx_bitmap_to_tids");
  CheckStatus(qsx_stream_synchronize(CurrentStream()), "qsx_stream_synchronize");   // (zero / ones / tw go out of scope)
  return rows;
}

// rows[s] tuples with their null bits at bitmaps[s] (nullptr: none NULL), s in order -> one bitmap of Sum(rows) bits at dst
// (word-aligned destination, trailing bits zero).  Empty pieces are dropped: two segments must not start at the same row.
void PackNullBits(const std::vector<const std::uint64_t *> &bitmaps, const std::vector<std::int64_t> &rows, const DeviceBuffer &row_numbers,
                  std::uint64_t *dst) {
  std::vector<const std::uint64_t *> segs;
  std::vector<std::int64_t> first;
  std::int64_t at = 0;
  for (std::size_t s = 0; s < rows.size(); ++s) {
    if (rows[s] == 0) continue;
    segs.push_back(bitmaps[s]);
    first.push_back(at);
    at += rows[s];
  }
  if (at == 0) return;
  CheckStatus(qsx_bitmap_gather_segmented(static_cast<int>(segs.size()), segs.data(), first.data(), static_cast<const std::int32_t *>(row_numbers.ptr), at,
                                          dst, CurrentStream()), "qsx_bitmap_gather_segmented");
}
}  // namespace

// (not in the anonymous namespace: PartitionExchangeOperator names it as a friend)
class PartitionExchangeWorkOrder : public WorkOrder {
 public:
  // more_to_come: this rank's input is not complete yet — another work order of the operator follows (see
  // PartitionExchangeOperator::getAllWorkOrders); the flag travels with the round's first counts exchange, and the round in which
  // no rank has more to come is the operator's last on every rank.
  PartitionExchangeWorkOrder(std::size_t query_id, PartitionExchangeOperator *op, std::vector<std::vector<block_id>> &&blocks, bool more_to_come,
                             InsertDestination *dest, StorageManager *storage_manager)
      : WorkOrder(query_id), op_(op), blocks_(std::move(blocks)), more_to_come_(more_to_come), dest_(dest), storage_manager_(storage_manager) {}

  void execute() override {
    try {
      if (op_->broadcast_) {
        broadcastAll();
      } else {
        const std::size_t world = static_cast<std::size_t>(op_->ranks_->world());
        const std::size_t parts = blocks_.size();
        for (std::size_t first = 0; first < parts; first += world) exchangeRound(first);
      }
      op_->roundFinished(!any_rank_has_more_);
    } catch (...) {
      op_->roundFinished(true);   // (the query is over: the Foreman rethrows)
      // A failure every rank agreed on (RankGroup::agreeOn) is thrown by all of them between collectives: nothing is in
      // flight.  Anything else happened on this rank alone while a step's collectives were being issued — the peers are
      // inside them, or about to be: the communicator is given up so that they end with QSX_ERR_COMM instead of waiting.
      if (in_collectives_) op_->ranks_->abort();
      throw;
    }
  }

 private:
  struct Piece {   // the local tuples bound for one rank in one round
    std::vector<BlockReference> blocks;
    std::int64_t rows = 0;
  };

  // Partitions first .. first + world - 1: partition first + r goes to rank r.
  // Order of a round: everything that can fail on this rank alone — fetching the blocks, validating what is
  // about to arrive, the output block, every send buffer — happens BEFORE the first data collective and its outcome is
  // agreed on with the peers (RankGroup::agreeOn); behind the agreement only copies and collectives are issued.
  void exchangeRound(std::size_t first) {
    RankGroup *ranks = op_->ranks_;
    const std::size_t world = static_cast<std::size_t>(ranks->world()), me = static_cast<std::size_t>(ranks->rank());
    const CatalogRelation &relation = op_->output_relation_;
    std::vector<Piece> pieces(world);
    std::vector<std::int64_t> send_rows(world, 0);
    ranks->agreeOn([&]() {
      for (std::size_t r = 0; r < world; ++r) {
        if (first + r >= blocks_.size()) continue;
        for (block_id id : blocks_[first + r]) {
          BlockReference b = storage_manager_->getBlock(id);
          if (b->numTuples() == 0) continue;
          pieces[r].blocks.push_back(b);
          pieces[r].rows += b->numTuples();
        }
        send_rows[r] = pieces[r].rows;
      }
    }, "PartitionExchangeOperator: a rank could not collect its blocks");
    in_collectives_ = true;
    const std::vector<std::int64_t> recv_rows = exchangeCounts(send_rows, /*with_flag=*/first == 0);
    in_collectives_ = false;
    const std::int64_t total_send = Sum(send_rows), total_recv = Sum(recv_rows);
    block_id out_id = 0;
    BlockReference out;
    std::unique_ptr<DeviceBuffer> row_numbers;
    std::vector<std::unique_ptr<DeviceBuffer>> send(relation.size()), send_bits(relation.size()), recv_bits(relation.size());
    std::vector<std::int64_t> send_words(world), recv_words(world);
    ranks->agreeOn([&]() {
      const bool owns_one = first + me < blocks_.size();
      if (!owns_one && total_recv != 0) throw ExecutionError("PartitionExchangeOperator: tuples arrived for a partition that does not exist", QSX_ERR_INVALID_ARGUMENT);
      if (total_recv > 0) out = dest_->getBlockForInsertion(total_recv, &out_id);
      std::int64_t max_rows = total_recv;
      for (std::size_t r = 0; r < world; ++r) {
        send_words[r] = WordsOf(send_rows[r]);
        recv_words[r] = WordsOf(recv_rows[r]);
        max_rows = std::max(max_rows, send_rows[r]);
      }
      bool nullable_attribute = false;
      for (std::size_t a = 0; a < relation.size(); ++a) nullable_attribute = nullable_attribute || relation.getAttributeType(static_cast<attribute_id>(a)).nullable;
      // (the rank's own piece is copied straight into the output block unless null bits travel: nothing is staged for it)
      const std::int64_t staged_rows = nullable_attribute ? total_send : total_send - send_rows[me];
      for (std::size_t a = 0; a < relation.size(); ++a) {
        const Type &t = relation.getAttributeType(static_cast<attribute_id>(a));
        send[a].reset(new DeviceBuffer(static_cast<std::size_t>(staged_rows) * t.width + 8));
        if (t.nullable) {
          if (row_numbers == nullptr) row_numbers = RowNumbers(max_rows);
          send_bits[a].reset(new DeviceBuffer(static_cast<std::size_t>(Sum(send_words)) * 8 + 8));
          recv_bits[a].reset(new DeviceBuffer(static_cast<std::size_t>(Sum(recv_words)) * 8 + 8));
        }
      }
    }, "PartitionExchangeOperator: a rank could not prepare its side of the exchange");
    in_collectives_ = true;
    // This rank's own partition needs no transport: its tuples are copied from their blocks straight to the END of the output
    // block (behind what the peers sent, in rank order), and only the pieces bound for other ranks are staged and exchanged —
    // one copy of 1 / world of the relation instead of a staging copy and a trip through the transport.  (With a nullable
    // attribute everything takes the transport, in rank order: the null bits of a block are packed per source rank.)
    bool any_nullable = false;
    for (std::size_t a = 0; a < relation.size(); ++a) any_nullable = any_nullable || relation.getAttributeType(static_cast<attribute_id>(a)).nullable;
    const bool self_direct = !any_nullable;
    std::vector<std::int64_t> wire_send = send_rows, wire_recv = recv_rows;
    if (self_direct) wire_send[me] = wire_recv[me] = 0;
    for (std::size_t a = 0; a < relation.size(); ++a) {
      const Type &t = relation.getAttributeType(static_cast<attribute_id>(a));
      // values: the pieces back to back in rank order -> all-to-all(v) straight into the output block's stripe
      // (every block's piece in ONE launch per direction: qsx_copy_segments, not a copy call per block)
      std::vector<const void *> from;
      std::vector<void *> to_where;
      std::vector<std::int64_t> piece_bytes;
      std::int64_t at = 0;
      for (std::size_t r = 0; r < world; ++r) {
        if (self_direct && r == me) continue;
        for (const BlockReference &b : pieces[r].blocks) {
          from.push_back(b->stripe(static_cast<attribute_id>(a)));
          to_where.push_back(static_cast<char *>(send[a]->ptr) + at * t.width);
          piece_bytes.push_back(b->numTuples() * t.width);
          at += b->numTuples();
        }
      }
      CheckStatus(qsx_copy_segments(static_cast<std::int64_t>(from.size()), from.data(), to_where.data(), piece_bytes.data(), CurrentStream()),
                  "qsx_copy_segments");
      CheckStatus(qsx_alltoallv(ranks->comm(), t.width, send[a]->ptr, wire_send.data(), out != nullptr ? out->stripe(static_cast<attribute_id>(a)) : nullptr,
                                wire_recv.data(), CurrentStream()), "qsx_alltoallv");
      if (self_direct && out != nullptr) {
        from.clear(); to_where.clear(); piece_bytes.clear();
        std::int64_t to = total_recv - recv_rows[me];
        for (const BlockReference &b : pieces[me].blocks) {
          from.push_back(b->stripe(static_cast<attribute_id>(a)));
          to_where.push_back(static_cast<char *>(out->stripe(static_cast<attribute_id>(a))) + to * t.width);
          piece_bytes.push_back(b->numTuples() * t.width);
          to += b->numTuples();
        }
        CheckStatus(qsx_copy_segments(static_cast<std::int64_t>(from.size()), from.data(), to_where.data(), piece_bytes.data(), CurrentStream()),
                    "qsx_copy_segments");
      }
      op_->bytes_sent_ += static_cast<std::uint64_t>(total_send - send_rows[me]) * t.width;
      if (t.nullable) {
        // null bits: one word-aligned bitmap per (this rank, destination) and per (source, this rank)
        std::int64_t word_at = 0;
        for (std::size_t r = 0; r < world; ++r) {
          std::vector<const std::uint64_t *> bitmaps;
          std::vector<std::int64_t> rows;
          for (const BlockReference &b : pieces[r].blocks) {
            bitmaps.push_back(b->nullBitmap(static_cast<attribute_id>(a)));
            rows.push_back(b->numTuples());
          }
          PackNullBits(bitmaps, rows, *row_numbers, static_cast<std::uint64_t *>(send_bits[a]->ptr) + word_at);
          word_at += send_words[r];
        }
        CheckStatus(qsx_alltoallv(ranks->comm(), 8, send_bits[a]->ptr, send_words.data(), recv_bits[a]->ptr, recv_words.data(), CurrentStream()),
                    "qsx_alltoallv(null bits)");
        if (out != nullptr) {
          std::vector<const std::uint64_t *> bitmaps;
          word_at = 0;
          for (std::size_t r = 0; r < world; ++r) {
            bitmaps.push_back(static_cast<const std::uint64_t *>(recv_bits[a]->ptr) + word_at);
            word_at += recv_words[r];
          }
          PackNullBits(bitmaps, recv_rows, *row_numbers, out->nullBitmap(static_cast<attribute_id>(a)));
        }
      }
    }
    ranks->synchronize();         // (the watchdog's wait: a peer that never arrives ends the round with QSX_ERR_COMM)
    in_collectives_ = false;
    if (out != nullptr) dest_->returnBlock(out_id, total_recv, first + me);
  }

  // Every rank receives the tuples of all ranks (unpartitioned relation, broadcast join build side).
  void broadcastAll() {
    RankGroup *ranks = op_->ranks_;
    const std::size_t world = static_cast<std::size_t>(ranks->world()), me = static_cast<std::size_t>(ranks->rank());
    const CatalogRelation &relation = op_->output_relation_;
    Piece mine;
    ranks->agreeOn([&]() {
      for (block_id id : blocks_.at(0)) {
        BlockReference b = storage_manager_->getBlock(id);
        if (b->numTuples() == 0) continue;
        mine.blocks.push_back(b);
        mine.rows += b->numTuples();
      }
    }, "PartitionExchangeOperator: a rank could not collect its blocks");
    in_collectives_ = true;
    const std::vector<std::int64_t> rows = exchangeCounts(std::vector<std::int64_t>(world, mine.rows), /*with_flag=*/true);   // rows[r] = rank r's tuples
    in_collectives_ = false;
    const std::int64_t total = Sum(rows);
    if (total == 0) return;
    std::int64_t pad = 1;
    for (std::int64_t r : rows) pad = std::max(pad, r);
    block_id out_id = 0;
    BlockReference out;
    std::unique_ptr<DeviceBuffer> row_numbers;
    std::vector<std::unique_ptr<DeviceBuffer>> send_of(relation.size()), gathered_of(relation.size());
    ranks->agreeOn([&]() {      // the output block and every buffer before the first all-gather (see exchangeRound)
      out = dest_->getBlockForInsertion(total, &out_id);
      for (std::size_t a = 0; a < relation.size(); ++a) {
        const Type &t = relation.getAttributeType(static_cast<attribute_id>(a));
        const std::size_t piece_bytes = static_cast<std::size_t>(pad) * t.width;
        send_of[a].reset(new DeviceBuffer(piece_bytes + 8));
        gathered_of[a].reset(new DeviceBuffer(piece_bytes * world + 8));
        if (t.nullable && row_numbers == nullptr) row_numbers = RowNumbers(total);
      }
    }, "PartitionExchangeOperator: a rank could not prepare its side of the broadcast");
    in_collectives_ = true;
    for (std::size_t a = 0; a < relation.size(); ++a) {
      const Type &t = relation.getAttributeType(static_cast<attribute_id>(a));
      const std::size_t piece_bytes = static_cast<std::size_t>(pad) * t.width;
      DeviceBuffer &send = *send_of[a], &gathered = *gathered_of[a];
      std::int64_t at = 0;
      for (const BlockReference &b : mine.blocks) {
        CheckStatus(qsx_copy_on_device(static_cast<char *>(send.ptr) + at * t.width, b->stripe(static_cast<attribute_id>(a)),
                                       static_cast<std::size_t>(b->numTuples()) * t.width, CurrentStream()), "qsx_copy_on_device");
        at += b->numTuples();
      }
      CheckStatus(qsx_allgather(ranks->comm(), send.ptr, piece_bytes, gathered.ptr, CurrentStream()), "qsx_allgather");
      op_->bytes_sent_ += static_cast<std::uint64_t>(mine.rows) * t.width * (world - 1);
      at = 0;
      for (std::size_t r = 0; r < world; ++r) {
        if (rows[r] == 0) continue;
        CheckStatus(qsx_copy_on_device(static_cast<char *>(out->stripe(static_cast<attribute_id>(a))) + at * t.width,
                                       static_cast<const char *>(gathered.ptr) + r * piece_bytes, static_cast<std::size_t>(rows[r]) * t.width,
                                       CurrentStream()), "qsx_copy_on_device");
        at += rows[r];
      }
      if (t.nullable) {
        const std::int64_t pad_words = WordsOf(pad);
        DeviceBuffer send_bits(static_cast<std::size_t>(pad_words) * 8 + 8), all_bits(static_cast<std::size_t>(pad_words) * 8 * world + 8);
        CheckStatus(qsx_memset_device(send_bits.ptr, 0, static_cast<std::size_t>(pad_words) * 8, CurrentStream()), "qsx_memset_device");
        std::vector<const std::uint64_t *> bitmaps;
        std::vector<std::int64_t> piece_rows;
        for (const BlockReference &b : mine.blocks) {
          bitmaps.push_back(b->nullBitmap(static_cast<attribute_id>(a)));
          piece_rows.push_back(b->numTuples());
        }
        PackNullBits(bitmaps, piece_rows, *row_numbers, static_cast<std::uint64_t *>(send_bits.ptr));
        CheckStatus(qsx_allgather(ranks->comm(), send_bits.ptr, static_cast<std::size_t>(pad_words) * 8, all_bits.ptr, CurrentStream()), "qsx_allgather(null bits)");
        bitmaps.clear();
        for (std::size_t r = 0; r < world; ++r) bitmaps.push_back(static_cast<const std::uint64_t *>(all_bits.ptr) + static_cast<std::int64_t>(r) * pad_words);
        PackNullBits(bitmaps, rows, *row_numbers, out->nullBitmap(static_cast<attribute_id>(a)));
      }
      ranks->synchronize();
    }
    in_collectives_ = false;
    (void)me;
    dest_->returnBlock(out_id, total, 0);
  }

  // with_flag: bit 62 of every count this rank sends says "more tuples of mine will follow in a later round"; what arrives tells
  // the same of every peer (any_rank_has_more_).
  std::vector<std::int64_t> exchangeCounts(const std::vector<std::int64_t> &send, bool with_flag) {
    constexpr std::int64_t kMoreBit = std::int64_t(1) << 62;
    const std::size_t world = send.size();
    std::vector<std::int64_t> tagged = send;
    if (with_flag && more_to_come_) {
      for (std::int64_t &v : tagged) v |= kMoreBit;
    }
    DeviceBuffer send_dev(world * 8 + 8), recv_dev(world * 8 + 8);
    CheckStatus(qsx_copy_to_device(send_dev.ptr, tagged.data(), world * 8, CurrentStream()), "qsx_copy_to_device");
    CheckStatus(qsx_exchange_counts(op_->ranks_->comm(), static_cast<const std::int64_t *>(send_dev.ptr), static_cast<std::int64_t *>(recv_dev.ptr),
                                    CurrentStream()), "qsx_exchange_counts");
    std::vector<std::int64_t> recv(world);
    CheckStatus(qsx_copy_to_host(recv.data(), recv_dev.ptr, world * 8, CurrentStream()), "qsx_copy_to_host");
    op_->ranks_->synchronize();
    for (std::int64_t &v : recv) {
      if (with_flag && (v & kMoreBit) != 0) any_rank_has_more_ = true;
      v &= ~kMoreBit;
    }
    if (with_flag && more_to_come_) any_rank_has_more_ = true;
    return recv;
  }

  PartitionExchangeOperator *op_;
  bool in_collectives_ = false;                 // a failure now is this rank's alone: execute() aborts the communicator
  std::vector<std::vector<block_id>> blocks_;   // per partition (broadcast: all in [0]): what arrived since the last round
  const bool more_to_come_;
  bool any_rank_has_more_ = false;
  InsertDestination *dest_;
  StorageManager *storage_manager_;
};

PartitionExchangeOperator::PartitionExchangeOperator(std::size_t query_id, const CatalogRelation &input_relation, bool input_relation_is_stored,
                                                     const CatalogRelation &output_relation,
                                                     QueryContext::insert_destination_id output_destination_index, RankGroup *ranks, bool broadcast)
    : RelationalOperator(query_id, broadcast ? 1 : input_relation.getNumPartitions()), input_relation_(input_relation),
      input_relation_is_stored_(input_relation_is_stored), output_relation_(output_relation),
      output_destination_index_(output_destination_index), ranks_(ranks), broadcast_(broadcast),
      input_(broadcast ? 1 : input_relation.getNumPartitions()) {
  consumed_.assign(input_.ids.size(), 0);
  if (ranks == nullptr) throw ExecutionError("PartitionExchangeOperator: no RankGroup", QSX_ERR_INVALID_ARGUMENT);
  if (input_relation.size() != output_relation.size()) {
    throw ExecutionError("PartitionExchangeOperator: input and output relation differ in their attributes", QSX_ERR_INVALID_ARGUMENT);
  }
  for (std::size_t a = 0; a < input_relation.size(); ++a) {
    const Type &x = input_relation.getAttributeType(static_cast<attribute_id>(a)), &y = output_relation.getAttributeType(static_cast<attribute_id>(a));
    if (x.id != y.id || x.width != y.width || x.nullable != y.nullable) {
      throw ExecutionError("PartitionExchangeOperator: input and output relation differ in their attributes", QSX_ERR_INVALID_ARGUMENT);
    }
  }
  if (broadcast) {
    if (output_relation.hasPartitionScheme()) throw ExecutionError("PartitionExchangeOperator: a broadcast output is not partitioned", QSX_ERR_INVALID_ARGUMENT);
  } else if (!input_relation.hasPartitionScheme() || output_relation.getNumPartitions() != input_relation.getNumPartitions()) {
    // (never ignored, like has_repartition against a plain destination: an unpartitioned input has nowhere to go)
    throw ExecutionError("PartitionExchangeOperator: input and output relation must share a hash partition scheme", QSX_ERR_INVALID_ARGUMENT);
  }
  if (input_relation_is_stored) {
    if (broadcast) {
      input_.ids[0] = input_relation.getBlocksSnapshot();
      for (partition_id p = 1; p < input_relation.getNumPartitions() && input_relation.hasPartitionScheme(); ++p) {
        for (block_id b : input_relation.getBlocksInPartition(p)) input_.ids[0].push_back(b);
      }
    } else {
      for (partition_id p = 0; p < num_partitions_; ++p) input_.ids[p] = input_relation.getBlocksInPartition(p);
    }
  }
}

// Rounds.  The reference's consumers take a producer's blocks as they are filled (streaming edges, kDataPipelineMessage); an
// exchange that waited for its whole local input would be a pipeline breaker and hold the whole shuffled relation at once.  So
// the operator issues ONE work order at a time, each over the blocks that have arrived since the last one: a round of
// collectives (counts, then the attributes).  Every rank runs the same number of rounds: the counts carry "more of mine will
// follow", and the round in which no rank says so is the last everywhere; a rank that has nothing new while a peer still has
// takes part with empty pieces.  (A rank waits for new blocks — or the end of its input — before it issues its next round; its
// peers wait for it inside theirs.)
bool PartitionExchangeOperator::getAllWorkOrders(WorkOrdersContainer *container, QueryContext *query_context, StorageManager *storage_manager,
                                                 const tmb::client_id, tmb::MessageBus *) {
  std::lock_guard<std::mutex> lock(mutex_);
  if (finished_) return true;
  if (round_in_flight_) return false;
  const bool input_done = input_relation_is_stored_ || done_feeding_input_relation_;
  std::vector<std::vector<block_id>> fresh(input_.ids.size());
  std::size_t fresh_blocks = 0;
  for (std::size_t p = 0; p < input_.ids.size(); ++p) {
    fresh[p].assign(input_.ids[p].begin() + static_cast<std::ptrdiff_t>(consumed_.at(p)), input_.ids[p].end());
    fresh_blocks += fresh[p].size();
  }
  // (a round that must not end the operator needs something to carry — except when the peers are waiting for this rank's part in
  // a round of theirs: rounds_ > 0 and the last one did not end it)
  if (!input_done && fresh_blocks == 0) return false;
  InsertDestination *dest = query_context->getInsertDestination(output_destination_index_);
  if (dest->isPartitionAware()) {
    throw ExecutionError("PartitionExchangeOperator: the output destination must not repartition (the tuples arrive partitioned)", QSX_ERR_INVALID_ARGUMENT);
  }
  for (std::size_t p = 0; p < input_.ids.size(); ++p) consumed_[p] = input_.ids[p].size();
  container->addNormalWorkOrder(new PartitionExchangeWorkOrder(query_id_, this, std::move(fresh), /*more_to_come=*/!input_done, dest, storage_manager), op_index_);
  round_in_flight_ = true;
  ++rounds_;
  return false;      // (finished only when a round has ended it: roundFinished)
}

void PartitionExchangeOperator::roundFinished(bool last) {
  std::lock_guard<std::mutex> lock(mutex_);
  round_in_flight_ = false;
  if (last) finished_ = true;
}

namespace {
class ExchangeAggregationStatesWorkOrder : public WorkOrder {
 public:
  ExchangeAggregationStatesWorkOrder(std::size_t query_id, std::vector<AggregationOperationState *> &&states, RankGroup *ranks)
      : WorkOrder(query_id), states_(std::move(states)), ranks_(ranks) {}
  void execute() override {
    for (AggregationOperationState *state : states_) state->mergeAcrossRanks(ranks_->comm());
  }
 private:
  std::vector<AggregationOperationState *> states_;
  RankGroup *ranks_;
};
}  // namespace

bool ExchangeAggregationStatesOperator::getAllWorkOrders(WorkOrdersContainer *container, QueryContext *query_context, StorageManager *,
                                                         const tmb::client_id, tmb::MessageBus *) {
  if (work_generated_) return true;
  std::vector<AggregationOperationState *> states;
  for (partition_id part = 0; part < num_partitions_; ++part) states.push_back(query_context->getAggregationState(aggr_state_index_, part));
  container->addNormalWorkOrder(new ExchangeAggregationStatesWorkOrder(query_id_, std::move(states), ranks_), op_index_);
  work_generated_ = true;
  return true;
}

}  // namespace quickstep
