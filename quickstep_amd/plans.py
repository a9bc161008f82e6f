"""Physical plans of BASELINE.json's multi-GPU configurations over the ops surface of quickstep_amd.capi:

  C4  partitioned hash join  orders ⋈ lineitem on orderkey, one 8-byte payload column per side
      (PartitionedJoin: K9 scatter -> all-to-all(v) of key + payload -> local build / probe -> K5 payload gathers);
  C5  TPC-H Q3 (benchmarks/tpch/queries/03.sql): customer ⋈ orders ⋈ lineitem with LIP filters, group by l_orderkey,
      SUM(l_extendedprice * (1 - l_discount)), ORDER BY revenue DESC LIMIT 10 (DistributedQ3).

One process per GPU; with one rank every collective degenerates to a local copy, so the same code is the single-GPU plan.
The reference runs these plans as per-partition work orders over shared memory (BuildHashOperator.cpp:82-91,
HashJoinOperator.cpp:220-231, AggregationOperator.cpp:49-61, merge at AggregationOperationState.cpp:831-843); here a
partition is a GPU and the two exchange steps are collectives (quickstep_amd/distributed.py).

Synthetic inputs (no dbgen here): dense keys, every rank owns a contiguous key range of each relation in random row order,
so `key & (P - 1)` is uniform over a rank's rows and a shuffle really moves (P - 1) / P of them.
"""
import torch

from . import distributed as qd
from . import types as T

DATE_CUT = 19950315          # '1995-03-15' as the 4-byte yyyymmdd stand-in used by tools/q3_pipeline.py
SEG_BUILDING = 1


def _world_rank(group=None):
    return qd.world_size(group), qd.rank_of(group)


# ------------------------------------------------------------------------------------------------ C4
def generate_c4_inputs(dev, orders_per_rank, rank, seed=5):
    """orders: o_orderkey INT (this rank's contiguous range, shuffled), 8-byte payload; lineitem: 1-7 lines per order
    (mean 4), l_orderkey clustered like dbgen writes it, 8-byte payload.  Payloads are functions of the key so that a
    joined row can be checked without the other side: o_payload = 3 * key + 1, l_payload = 5 * key + line number."""
    g = torch.Generator(device=dev)
    g.manual_seed(seed + 1000 * rank)
    first = rank * orders_per_rank + 1
    o_orderkey = torch.randperm(orders_per_rank, device=dev, generator=g, dtype=torch.int32) + first
    o_payload = o_orderkey.long() * 3 + 1
    lines = torch.randint(1, 8, (orders_per_rank,), device=dev, generator=g)
    l_orderkey = torch.repeat_interleave(torch.arange(first, first + orders_per_rank, device=dev, dtype=torch.int32), lines)
    starts = torch.cumsum(lines, 0) - lines
    line_no = torch.arange(l_orderkey.numel(), device=dev) - torch.repeat_interleave(starts, lines)
    l_payload = l_orderkey.long() * 5 + line_no
    del starts, line_no, lines
    return {"o_orderkey": o_orderkey, "o_payload": o_payload, "l_orderkey": l_orderkey, "l_payload": l_payload}


class PartitionedJoin:
    """BASELINE config 4.  step() = shuffle both sides on the join key, build, probe, materialise
    (key, o_payload, l_payload) for every pair of this rank's partition."""

    def __init__(self, ops, orders_total, est_orders_per_rank, group=None, dense=True, fused=True, overlap=True):
        """fused: the probe writes the output relation itself (PartitionedHashJoin.probe_output) instead of a pair list that
        K5 gathers materialise.  overlap: the probe side's exchange runs on a second stream under the build kernel."""
        self.ops, self.group, self.fused, self.overlap = ops, group, fused, overlap
        self._side = None
        self.join = qd.PartitionedHashJoin(ops, T.INT, est_orders_per_rank, group=group,
                                           key_domain=(1, orders_total) if dense else None)

    def step(self, inputs, tid_base_orders=0, tid_base_lines=0):
        j = self.join
        overlapped = self.fused and self.overlap and inputs["l_orderkey"].is_cuda
        if overlapped:
            # The build kernel is one atomic per row (bound by the atomic units: HBM and the links idle), the probe side's
            # K9 scatter + exchange is HBM- / link-bound: the second runs on a side stream under the first.  The build side's
            # exchange has finished (its counts were read on the host) before the probe side's starts, so every rank still
            # issues its collectives in the same order.
            main = torch.cuda.current_stream()
            if self._side is None:
                self._side = torch.cuda.Stream()
            j.shuffle_build(inputs["o_orderkey"], tid_base_orders, payload=[inputs["o_payload"]], with_tids=False)
            self._side.wait_stream(main)              # (the inputs are ready; nothing of the build is queued yet)
            j.table.clear()
            j.table.build(j.build_keys)               # main stream, asynchronous
            with torch.cuda.stream(self._side):
                received = j.shuffle_probe(inputs["l_orderkey"], tid_base_lines, payload=[inputs["l_payload"]], with_tids=False)
            for t in received:
                t.record_stream(main)                 # allocated under the side stream, read by the probe on the main one
            main.wait_stream(self._side)
            cols = j.probe_output_received()
            return cols, j.shuffled_bytes
        j.shuffle_build(inputs["o_orderkey"], tid_base_orders, payload=[inputs["o_payload"]], with_tids=not self.fused)
        j.build_received()
        # a lineitem row has exactly one order: the rows that arrive bound the pairs; how many arrive is only known
        # after the counts exchange, so the capacity is left to probe() (rows received)
        if self.fused:
            cols = j.probe_output(inputs["l_orderkey"], tid_base_lines, payload=[inputs["l_payload"]])
        else:
            _, _, out_p, out_b, count = j.probe(inputs["l_orderkey"], tid_base_lines, capacity=0, payload=[inputs["l_payload"]])
            cols = j.materialize_payload(out_p, out_b, count)
        return cols, j.shuffled_bytes

    @staticmethod
    def check(cols):
        """Every output row satisfies the join condition (payloads are functions of their side's key) — exact, any size."""
        key, o_pay, l_pay = cols
        k = key.long()
        ok_o = bool((o_pay == 3 * k + 1).all())
        line = l_pay - 5 * k
        ok_l = bool(((line >= 0) & (line < 7)).all())
        return ok_o and ok_l


# ------------------------------------------------------------------------------------------------ C5
def generate_q3_inputs(dev, sf_per_rank, rank, world, seed=7):
    """TPC-H-shaped columns of Q3 for one rank at scale factor sf_per_rank (150 K customers, 1.5 M orders, ~6 M lineitems
    per unit): this rank's contiguous custkey / orderkey ranges in random row order, o_custkey over the customers of all
    `world` ranks, dates as 4-byte yyyymmdd integers, lineitem clustered on l_orderkey."""
    g = torch.Generator(device=dev)
    g.manual_seed(seed + 1000 * rank)
    n_c, n_o = int(150_000 * sf_per_rank), int(1_500_000 * sf_per_rank)
    c_first, o_first = rank * n_c + 1, rank * n_o + 1
    c_custkey = torch.randperm(n_c, device=dev, generator=g, dtype=torch.int32) + c_first
    c_mktsegment = torch.randint(0, 5, (n_c,), device=dev, generator=g, dtype=torch.int32)
    o_orderkey = torch.randperm(n_o, device=dev, generator=g, dtype=torch.int32) + o_first
    o_custkey = torch.randint(1, n_c * world + 1, (n_o,), device=dev, generator=g, dtype=torch.int32)
    o_orderdate = torch.randint(19920101, 19981231, (n_o,), device=dev, generator=g, dtype=torch.int32)
    # 1-7 lines per order (mean 4): 1-4 of them on the rank that holds the order, 0-3 on the previous rank, so that
    # partial sums of one group really meet in the merge; both runs clustered on l_orderkey like dbgen's output
    nxt_first = ((rank + 1) % world) * n_o + 1
    lines_own = torch.randint(1, 5, (n_o,), device=dev, generator=g)
    lines_nxt = torch.randint(0, 4, (n_o,), device=dev, generator=g)
    l_orderkey = torch.cat([
        torch.repeat_interleave(torch.arange(o_first, o_first + n_o, device=dev, dtype=torch.int32), lines_own),
        torch.repeat_interleave(torch.arange(nxt_first, nxt_first + n_o, device=dev, dtype=torch.int32), lines_nxt)])
    del lines_own, lines_nxt
    n_l = l_orderkey.numel()
    l_extendedprice = torch.rand(n_l, device=dev, generator=g, dtype=torch.float64) * 104100 + 900
    l_discount = torch.randint(0, 11, (n_l,), device=dev, generator=g).double() / 100
    l_shipdate = torch.randint(19920101, 19981231, (n_l,), device=dev, generator=g, dtype=torch.int32)
    return {"c_custkey": c_custkey, "c_mktsegment": c_mktsegment, "o_orderkey": o_orderkey, "o_custkey": o_custkey,
            "o_orderdate": o_orderdate, "l_orderkey": l_orderkey, "l_extendedprice": l_extendedprice,
            "l_discount": l_discount, "l_shipdate": l_shipdate,
            "customers_total": n_c * world, "orders_total": n_o * world}


def q3_input_bytes(inputs):
    """Algorithmic bytes of one Q3 pass over this rank's share: the referenced columns, once."""
    n_c, n_o, n_l = inputs["c_custkey"].numel(), inputs["o_orderkey"].numel(), inputs["l_orderkey"].numel()
    return n_c * (4 + 4) + n_o * (4 + 4 + 4) + n_l * (4 + 8 + 8 + 4)


class DistributedQ3:
    """BASELINE config 5: the Q3 plan with broadcast build sides and a reduce-scatter of the partial aggregates.

      customer   select c_mktsegment = 'BUILDING'; qualifying custkeys all-gathered -> every rank builds the whole
                 customer table; exact LIP bit vector on custkey, OR-ed across ranks
      orders     select o_orderdate < DATE; LIP probe on o_custkey; semi probe of the customer table; qualifying
                 (orderkey, global tid) all-gathered -> every rank builds the whole qualifying-orders table; exact LIP
                 bit vector on orderkey, OR-ed across ranks
      lineitem   select l_shipdate > DATE; LIP probe on l_orderkey; inner probe of the orders table (local rows only: no
                 shuffle of the big relation); group by l_orderkey into this rank's dense state THROUGH the pair list
      merge      reduce-scatter of the dense state images: rank r ends with the merged groups of key range r and
                 finalizes them; top 10 of every rank -> all-gather -> the global top 10.

    Reference plan: ExecutionGenerator's broadcast hash join (BuildHashOperator.hpp:99,146-152) + LIP deployment
    (query_optimizer/rules/AttachLIPFilters.cpp) + CollisionFreeVector aggregation (StarSchemaSimpleCostModel.cpp:614-776)."""

    def __init__(self, ops, customers_total, orders_total, group=None, use_lip=True, fused=True):
        self.ops, self.group, self.use_lip, self.fused = ops, group, use_lip, fused
        self.world, self.rank = _world_rank(group)
        self.n_c, self.n_o = customers_total, orders_total
        self.t_c = ops.JoinTable(T.INT, customers_total, key_range=(1, customers_total))
        self.t_o = ops.JoinTable(T.INT, orders_total, key_range=(1, orders_total))
        self.lip_c = ops.LipFilter(T.LIP_BITVECTOR_EXACT, customers_total, 1)
        self.lip_o = ops.LipFilter(T.LIP_BITVECTOR_EXACT, orders_total, 1)
        kw = dict(code_widths=[4, 4, 4]) if fused else {}
        self.cfg = T.make_agg_config(T.AGG_COLLISION_FREE, [(T.INT, None), (T.DOUBLE, None), (T.DOUBLE, None)], keys=[0],
                                     instrs=[(T.EX_SUB, 0, T.const(0), T.col(2)), (T.EX_MUL, 1, T.col(1), T.temp(0))],
                                     consts=[1.0], aggs=[(T.AGG_SUM, T.temp(1))], num_entries=orders_total + 1, **kw)
        self.state = ops.AggState(self.cfg)
        self.comm_bytes = 0

    # -- collectives (identity with one rank) -------------------------------------------------------
    def _or_filter(self, lip):
        if self.world == 1:
            return
        words = lip.export(self._dev)
        qd._allreduce_or(words, self.group)
        self.comm_bytes += words.numel() * 8 * (self.world - 1) // self.world * 2
        lip.merge_or(words)

    def _all_gather_rows(self, cols, count):
        """Variable-length all-gather of `count` leading rows of every column (qualifying build rows of every rank)."""
        if self.world == 1:
            return [c[:count] for c in cols], [count]
        dev = cols[0].device
        counts = torch.empty(self.world, dtype=torch.int64, device=dev)
        qd.xfer.all_gather_into_tensor(counts, torch.tensor([count], dtype=torch.int64, device=dev), group=self.group)
        counts = counts.cpu().tolist()
        pad = max(max(counts), 1)
        out = []
        for c in cols:
            mine = c.new_zeros(pad)
            mine[:count] = c[:count]
            gathered = torch.empty(self.world * pad, dtype=c.dtype, device=dev)
            qd.xfer.all_gather_into_tensor(gathered, mine, group=self.group)
            out.append(torch.cat([gathered[r * pad: r * pad + counts[r]] for r in range(self.world)]))
            self.comm_bytes += (self.world - 1) * pad * c.element_size()
        return out, counts

    # -- the query ----------------------------------------------------------------------------------
    def run(self, inp, tid_base_orders=0):
        ops, dev = self.ops, inp["c_custkey"].device
        self._dev = dev
        self.comm_bytes = 0
        self.t_c.clear(); self.t_o.clear(); self.state.clear()
        if self.use_lip:
            self.lip_c.clear(); self.lip_o.clear()
        ph = qd.phases.phase
        # customer
        with ph("customer: select + gather + build + LIP"):
            c_sel, c_cnt = ops.select_cmp(inp["c_mktsegment"], T.EQ, SEG_BUILDING)
            (c_keys,), _ = ops.compact_gather([inp["c_custkey"]], c_sel, inp["c_custkey"].numel())
            n_c_sel = int(c_cnt.item())
            (all_c,), _ = self._all_gather_rows([c_keys], n_c_sel)
            self.t_c.build(all_c)
            if self.use_lip:
                # (the table holds every rank's qualifying customers: read off it, the filter needs no OR across the ranks)
                if not (hasattr(self.lip_c, "build_from_table") and self.lip_c.build_from_table(self.t_c, n_c_sel)):
                    self.lip_c.build(inp["c_custkey"], filter_bitmap=c_sel)
                    self._or_filter(self.lip_c)
        # orders
        with ph("orders: select + LIP probe + semi probe + gather + build + LIP"):
            o_sel, _ = ops.select_cmp(inp["o_orderdate"], T.LT, DATE_CUT)
            if self.use_lip and self.fused and hasattr(self.t_c, "probe_exists_lip"):
                # (the semi join's own LIP prober: filter bit and table word from one pass over o_custkey)
                o_ok, o_cnt = self.t_c.probe_exists_lip(inp["o_custkey"], [self.lip_c], filter_bitmap=o_sel)
            else:
                o_lip = self.lip_c.probe(inp["o_custkey"], in_bitmap=o_sel)[0] if self.use_lip else o_sel
                o_ok, o_cnt = self.t_c.probe_exists(inp["o_custkey"], filter_bitmap=o_lip)
            n_o_local = inp["o_orderkey"].numel()
            # the qualifying orders' keys and their tuple ids: the ids come from the bitmap itself (no row-number column is
            # written and gathered: 56 M x 4 bytes out and in again per step)
            (o_keys,), _ = ops.compact_gather([inp["o_orderkey"]], o_ok, n_o_local)
            if hasattr(ops, "bitmap_to_tids"):
                o_tids, _ = ops.bitmap_to_tids(o_ok, n_o_local, tid_base_orders)
            else:
                (o_tids,), _ = ops.compact_gather([self._tids(n_o_local, tid_base_orders, dev)], o_ok, n_o_local)
            (all_ok, all_ot), counts = self._all_gather_rows([o_keys, o_tids], int(o_cnt.item()))
            # the table keeps the position in the gathered list; all_ot turns it into the global orders tuple id
            self.t_o.build(all_ok)
            self.qualifying_order_tids = all_ot
            if self.use_lip:
                # all ranks hold all qualifying keys already: no OR needed.  The table is an existence map of the orders' key range:
                # the filter's bits are read off it (qsx_lip_build_from_join_table) where that beats one atomic per key
                if not (hasattr(self.lip_o, "build_from_table") and self.lip_o.build_from_table(self.t_o, int(all_ok.numel()))):
                    self.lip_o.build(all_ok)
        # lineitem
        with ph("lineitem: select l_shipdate"):
            l_sel, l_sel_count = ops.select_cmp(inp["l_shipdate"], T.GT, DATE_CUT)
        if self.use_lip and self.fused and hasattr(self.t_o, "probe_lip"):
            # the join work order's own LIP prober (HashJoinOperator.cpp:450-470): filter bit and table word from one pass over
            # l_orderkey; the rows that pass the predicate bound the pairs (an order is unique), so nothing is counted first
            with ph("lineitem: LIP + inner probe (one pass)"):
                p, b, cnt = self.t_o.probe_lip(inp["l_orderkey"], [self.lip_o], capacity=int(l_sel_count.item()), filter_bitmap=l_sel)
                total = int(cnt.item())
        else:
            with ph("lineitem: LIP probe"):
                if self.use_lip:
                    l_lip, l_live = self.lip_o.probe(inp["l_orderkey"], in_bitmap=l_sel)
                else:
                    l_lip, l_live = l_sel, l_sel_count
            with ph("lineitem: inner probe"):
                p, b, cnt = self.t_o.probe(inp["l_orderkey"], capacity=int(l_live.item()), filter_bitmap=l_lip)
                total = int(cnt.item())
        pt = p[:total]
        with ph("group by l_orderkey (dense state, through the pair list)"):
            if self.fused:
                self.state.update_coded([pt, pt, pt], [inp["l_orderkey"], inp["l_extendedprice"], inp["l_discount"]], total)
            else:
                self.state.update([ops.gather(inp["l_orderkey"], pt), ops.gather(inp["l_extendedprice"], pt),
                                   ops.gather(inp["l_discount"], pt)], total)
        # merge + finalize this rank's key range
        with ph("merge (reduce-scatter)"):
            self._merge_state(dev)
        with ph("finalize + top 10"):
            keys, vals, _, groups = self.state.finalize(dev, partition=self.rank, num_partitions=self.world)
            g = int(groups.item())
            k = min(10, g)
            if k:
                perm = ops.sort_top_k([vals[0][:g]], k, [True])
                top_keys, top_rev = ops.gather(keys[0][:g], perm), ops.gather(vals[0][:g], perm)
            else:
                top_keys, top_rev = keys[0][:0], vals[0][:0]
            (all_keys, all_rev), _ = self._all_gather_rows([top_keys, top_rev], k)
            if all_rev.numel() > 10:
                perm = ops.sort_top_k([all_rev], 10, [True])
                all_keys, all_rev = ops.gather(all_keys, perm), ops.gather(all_rev, perm)
        return {"pairs": total, "groups": g, "top_keys": all_keys, "top_revenue": all_rev,
                "qualifying_customers": int(all_c.numel()), "qualifying_orders": int(all_ok.numel())}

    @staticmethod
    def _tids(n, base, dev):
        return torch.arange(base, base + n, dtype=torch.int32, device=dev)

    def _merge_state(self, dev):
        if self.world > 1:
            self.comm_bytes += qd.reduce_scatter_dense_state(self.state, dev, group=self.group)
