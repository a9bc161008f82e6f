// Reference block images in, stripes out: images laid out like the reference's buffer pool holds them
// (storage/StorageBlock.cpp:81-195 + storage/BasicColumnStoreTupleStorageSubBlock.cpp:100-183, written by
// block_image_util.hpp) are copied to device memory as they are and ADOPTED in place (StorageManager::adoptBlockImage: the
// block's stripes and null bitmaps point into the image, stripes max_tuples x width apart).  Select, HashJoin and
// Aggregation over the adopted blocks must give what they give over blocks loaded column by column (loadBlock) from the
// same values — a non-nullable and a nullable relation, 2 MB and 4 MB blocks, ragged fill, one empty block, work orders per
// block and per run of blocks.  The same for COMPRESSED column store images (the reference's TPC-H DDL stores lineitem and
// orders that way, benchmarks/tpch/create.sql:69-121): l_orderkey truncated to 2 bytes and the block's sort column,
// l_quantity dictionary-coded in one byte — with the NULL code of compression/CompressionDictionary.hpp:49-52 in the nullable
// relation — l_extendedprice as values with its own null bitmap; codes and dictionaries are used where they lie.
#include <algorithm>
#include <cstring>
#include <map>

#include "block_image_util.hpp"
#include "test_util.hpp"

using namespace quickstep;

namespace {
struct Data {
  std::vector<std::vector<std::int32_t>> key, qty;
  std::vector<std::vector<double>> price;
  std::vector<std::vector<bool>> qty_null, price_null;
};

Data makeData(int blocks, std::int64_t rows_per_block) {
  Data d;
  std::uint64_t x = 0x243F6A8885A308D3ull;
  auto rnd = [&]() { x ^= x << 13; x ^= x >> 7; x ^= x << 17; return x; };
  for (int b = 0; b < blocks; ++b) {
    const std::int64_t n = b == 2 ? 0 : rows_per_block - 1009 * (b % 5);
    std::vector<std::int32_t> k(n), q(n);
    std::vector<double> p(n);
    std::vector<bool> qn(n), pn(n);
    for (std::int64_t i = 0; i < n; ++i) {
      k[i] = static_cast<std::int32_t>(rnd() % 5000);
      q[i] = static_cast<std::int32_t>(rnd() % 50) + 1;
      p[i] = static_cast<double>(rnd() % 1000000) / 100.0;
      qn[i] = rnd() % 13 == 0;
      pn[i] = rnd() % 7 == 0;
    }
    // ascending l_orderkey inside a block: what a column store sorted on it holds (the compressed images declare it)
    std::vector<std::size_t> order(static_cast<std::size_t>(n));
    for (std::size_t i = 0; i < order.size(); ++i) order[i] = i;
    std::stable_sort(order.begin(), order.end(), [&](std::size_t x, std::size_t y) { return k[x] < k[y]; });
    std::vector<std::int32_t> k2(k.size()), q2(q.size());
    std::vector<double> p2(p.size());
    std::vector<bool> qn2(qn.size()), pn2(pn.size());
    for (std::size_t i = 0; i < order.size(); ++i) { k2[i] = k[order[i]]; q2[i] = q[order[i]]; p2[i] = p[order[i]]; qn2[i] = qn[order[i]]; pn2[i] = pn[order[i]]; }
    d.key.push_back(k2); d.qty.push_back(q2); d.price.push_back(p2); d.qty_null.push_back(qn2); d.price_null.push_back(pn2);
  }
  return d;
}

struct Loaded {
  StorageManager storage;
  CatalogRelation rel{1, "lineitem"};
  std::vector<void *> images;   // device copies of the block images (owned here)
  ~Loaded() { for (void *p : images) qsx_device_free(p); }
};

void addAttributes(CatalogRelation *rel, bool nullable) {
  rel->addAttribute("l_orderkey", Type::Int());
  rel->addAttribute("l_quantity", nullable ? Type::Int().getNullableVersion() : Type::Int());
  rel->addAttribute("l_extendedprice", nullable ? Type::Double().getNullableVersion() : Type::Double());
}

// as_images: 0 = loaded column by column, 1 = BasicColumnStore images, 2 = CompressedColumnStore images
void load(Loaded *l, const Data &d, bool nullable, int as_images, std::size_t block_bytes) {
  addAttributes(&l->rel, nullable);
  for (std::size_t b = 0; b < d.key.size(); ++b) {
    const std::int64_t n = static_cast<std::int64_t>(d.key[b].size());
    if (as_images != 0) {
      std::int64_t max_tuples = 0;
      const std::vector<const void *> columns = {d.key[b].data(), d.qty[b].data(), d.price[b].data()};
      const std::vector<std::vector<bool>> null_flags = {{}, nullable ? d.qty_null[b] : std::vector<bool>(), nullable ? d.price_null[b] : std::vector<bool>()};
      std::vector<unsigned char> image;
      if (as_images == 2) {
        block_image::Coding truncated, dictionary, values;
        truncated.kind = block_image::Coding::kTruncated;
        truncated.code_width = 2;
        dictionary.kind = block_image::Coding::kDictionary;
        dictionary.code_width = 1;
        image = block_image::BuildCompressed(l->rel, columns, null_flags, n, block_bytes, 0, {truncated, dictionary, values}, &max_tuples);
      } else {
        image = block_image::Build(l->rel, columns, null_flags, n, block_bytes, -1, &max_tuples);
      }
      EXPECT_TRUE(max_tuples >= n);
      void *dev = nullptr;
      CheckStatus(qsx_device_alloc(image.size(), &dev), "qsx_device_alloc");
      CheckStatus(qsx_copy_to_device(dev, image.data(), image.size(), nullptr), "qsx_copy_to_device");
      CheckStatus(qsx_stream_synchronize(nullptr), "qsx_stream_synchronize");
      l->images.push_back(dev);
      const block_id id = l->storage.adoptBlockImage(&l->rel, dev, image.size());
      EXPECT_EQ(l->storage.getBlock(id)->numTuples(), n);
      if (as_images == 2) {   // adopted as codes: nothing decoded yet, the key column is the block's sort column
        BlockReference blk = l->storage.getBlock(id);
        EXPECT_TRUE(blk->compressedAttribute(0) != nullptr && blk->compressedAttribute(0)->kind == CompressedAttribute::kTruncated);
        EXPECT_TRUE(blk->compressedAttribute(1) != nullptr && blk->compressedAttribute(1)->kind == CompressedAttribute::kDictionary);
        EXPECT_TRUE(blk->compressedAttribute(2) == nullptr);
        EXPECT_TRUE(!blk->valuesMaterialized(0) && !blk->valuesMaterialized(1));
        EXPECT_EQ(blk->sortColumn(), 0);
      }
    } else {
      std::vector<std::uint64_t> qn(static_cast<std::size_t>((n + 63) / 64) + 1, 0), pn(qn.size(), 0);
      for (std::int64_t i = 0; i < n && nullable; ++i) {
        if (d.qty_null[b][i]) qn[i >> 6] |= 1ull << (63 - (i & 63));
        if (d.price_null[b][i]) pn[i >> 6] |= 1ull << (63 - (i & 63));
      }
      const std::vector<const std::uint64_t *> null_bitmaps = {nullptr, nullable ? qn.data() : nullptr, nullable ? pn.data() : nullptr};
      l->storage.loadBlock(&l->rel, {d.key[b].data(), d.qty[b].data(), d.price[b].data()}, n, 0, nullptr, &null_bitmaps);
    }
  }
}

struct Results {
  std::vector<std::pair<std::int32_t, double>> selected;      // (l_orderkey, l_extendedprice) where l_quantity < 24, NULL prices as -1
  std::map<std::int32_t, std::pair<double, std::int64_t>> groups;   // l_quantity -> (SUM(price), COUNT(*)) [NULL group: key -1]
  std::int64_t joined = 0;
};

Results run(const Data &d, bool nullable, int as_images, std::size_t block_bytes, std::size_t blocks_per_order) {
  Loaded l;
  load(&l, d, nullable, as_images, block_bytes);
  Results r;
  QueryContext ctx;
  // select l_orderkey, l_extendedprice where l_quantity < 24
  CatalogRelation sel(2, "sel");
  sel.addAttribute("l_orderkey", Type::Int());
  sel.addAttribute("l_extendedprice", nullable ? Type::Double().getNullableVersion() : Type::Double());
  Predicate p;
  p.conjuncts.push_back(ComparisonPredicate(1, ComparisonID::kLess, TypedLiteral::Int(24)));
  const auto pred = ctx.addPredicate(p);
  const auto d_sel = ctx.addInsertDestination(&sel, &l.storage);
  SelectOperator select(0, l.rel, false, sel, d_sel, pred, std::vector<attribute_id>{0, 2}, true);
  select.setBlocksPerWorkOrder(blocks_per_order);
  fetchAndExecuteWorkOrders(&select, &ctx, &l.storage);
  for (block_id b : ctx.getInsertDestination(d_sel)->getTouchedBlocks()) {
    BlockReference blk = l.storage.getBlock(b);
    const std::size_t k = static_cast<std::size_t>(blk->numTuples());
    if (k == 0) continue;
    std::vector<std::int32_t> key(k);
    std::vector<double> price(k);
    std::vector<std::uint64_t> nulls((k + 63) / 64);
    blk->copyAttributeToHost(0, key.data());
    blk->copyAttributeToHost(1, price.data());
    blk->copyNullBitmapToHost(1, nulls.data());
    for (std::size_t i = 0; i < k; ++i) r.selected.emplace_back(key[i], ((nulls[i >> 6] >> (63 - (i & 63))) & 1u) ? -1.0 : price[i]);
  }
  std::sort(r.selected.begin(), r.selected.end());
  // select l_quantity, sum(l_extendedprice), count(*) group by l_quantity
  CatalogRelation agg_out(3, "agg");
  agg_out.addAttribute("l_quantity", nullable ? Type::Int().getNullableVersion() : Type::Int());
  agg_out.addAttribute("sum", Type::Double().getNullableVersion());
  agg_out.addAttribute("count", Type::Long());
  AggregationStateSpec spec;
  spec.input_relation = &l.rel;
  spec.group_by = {1};
  spec.aggregates = {AggregateSpec(AggregationID::kSum, 2), AggregateSpec(AggregationID::kCount, kInvalidAttributeID)};
  spec.strategy = QSX_AGG_GENERIC;
  spec.estimated_num_groups = 64;
  const auto state = ctx.addAggregationState(spec);
  const auto d_agg = ctx.addInsertDestination(&agg_out, &l.storage);
  AggregationOperator agg(0, l.rel, true, state);
  agg.setBlocksPerWorkOrder(blocks_per_order);
  FinalizeAggregationOperator fin(0, state, 1, false, 1, agg_out, d_agg);
  fetchAndExecuteWorkOrders(&agg, &ctx, &l.storage);
  fetchAndExecuteWorkOrders(&fin, &ctx, &l.storage);
  for (block_id b : ctx.getInsertDestination(d_agg)->getTouchedBlocks()) {
    BlockReference blk = l.storage.getBlock(b);
    const std::size_t k = static_cast<std::size_t>(blk->numTuples());
    if (k == 0) continue;
    std::vector<std::int32_t> q(k);
    std::vector<double> sum(k);
    std::vector<std::int64_t> cnt(k);
    std::vector<std::uint64_t> qnull((k + 63) / 64);
    blk->copyAttributeToHost(0, q.data());
    blk->copyAttributeToHost(1, sum.data());
    blk->copyAttributeToHost(2, cnt.data());
    blk->copyNullBitmapToHost(0, qnull.data());
    for (std::size_t i = 0; i < k; ++i) r.groups[((qnull[i >> 6] >> (63 - (i & 63))) & 1u) ? -1 : q[i]] = {sum[i], cnt[i]};
  }
  // join with a small dimension on l_orderkey (probe side = the adopted blocks)
  CatalogRelation dim(4, "dim"), joined(5, "joined");
  dim.addAttribute("k", Type::Int());
  joined.addAttribute("k", Type::Int());
  std::vector<std::int32_t> dk;
  for (std::int32_t k = 0; k < 5000; k += 3) dk.push_back(k);
  l.storage.loadBlock(&dim, {dk.data()}, static_cast<std::int64_t>(dk.size()));
  const auto table = ctx.addJoinHashTable(kInt, static_cast<std::int64_t>(dk.size()));
  const auto d_join = ctx.addInsertDestination(&joined, &l.storage);
  const auto selection = ctx.addScalarGroup({0});
  const std::vector<bool> on_build{false};
  BuildHashOperator build(0, dim, true, {0}, false, 1, table);
  HashJoinOperator join(0, dim, l.rel, true, {0}, false, 1, false, joined, d_join, table, QueryContext::kInvalidPredicateId, selection, &on_build,
                        HashJoinOperator::JoinType::kInnerJoin);
  join.setBlocksPerWorkOrder(blocks_per_order);
  fetchAndExecuteWorkOrders(&build, &ctx, &l.storage);
  fetchAndExecuteWorkOrders(&join, &ctx, &l.storage);
  for (block_id b : ctx.getInsertDestination(d_join)->getTouchedBlocks()) r.joined += l.storage.getBlock(b)->numTuples();
  return r;
}
}  // namespace

int main() {
  if (qsx_device_count() < 1) {
    std::fprintf(stderr, "block_image_test needs an MI355X: %s\n", qsx_status_string(QSX_ERR_NO_DEVICE));
    return 2;
  }
  for (const bool nullable : {false, true}) {
    const std::size_t block_bytes = nullable ? (4u << 20) : (2u << 20);
    const Data d = makeData(7, nullable ? 200000 : 120000);
    // what the query computes, straight from the values
    std::int64_t want_joined = 0, want_selected = 0;
    for (std::size_t b = 0; b < d.key.size(); ++b) {
      for (std::size_t i = 0; i < d.key[b].size(); ++i) {
        want_joined += d.key[b][i] % 3 == 0 ? 1 : 0;
        want_selected += (!(nullable && d.qty_null[b][i]) && d.qty[b][i] < 24) ? 1 : 0;
      }
    }
    const Results loaded = run(d, nullable, 0, block_bytes, 1);
    EXPECT_EQ(loaded.joined, want_joined);
    EXPECT_EQ(static_cast<std::int64_t>(loaded.selected.size()), want_selected);
    for (const int form : {1, 1, 2, 2}) {
      static int turn = 0;
      const std::size_t per_order = (turn++ % 2) == 0 ? 1 : 4;
      const Results adopted = run(d, nullable, form, block_bytes, per_order);
      EXPECT_EQ(adopted.joined, want_joined);
      EXPECT_TRUE(adopted.selected == loaded.selected);
      EXPECT_EQ(adopted.groups.size(), loaded.groups.size());
      for (const auto &kv : loaded.groups) {
        const auto it = adopted.groups.find(kv.first);
        EXPECT_TRUE(it != adopted.groups.end());
        if (it == adopted.groups.end()) continue;
        EXPECT_EQ(it->second.second, kv.second.second);                                   // COUNT(*): exact
        EXPECT_NEAR(it->second.first, kv.second.first, 1e-9 * std::fabs(kv.second.first) + 1e-9);
      }
    }
  }
  return finish("block_image_test");
}
