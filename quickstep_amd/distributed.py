"""Multi-GPU execution of the hot path: one process per GPU, torch.distributed
(backend "nccl" = RCCL over xGMI) for the two exchange steps the path has.

The reference has no data-plane collective: single-node Quickstep
shares memory, and its partitioned execution is purely logical — relation rows
are routed to partition ``HashPartitionSchemeHeader::getPartitionId(key)``
(catalog/PartitionSchemeHeader.hpp:200-214) by PartitionAwareInsertDestination
(storage/InsertDestination.hpp:490-660), then each partition gets its own
BuildHash / HashJoin / Aggregation work orders (BuildHashOperator.cpp:82-91,
HashJoinOperator.cpp:220-231).  Here GPU g *is* partition g of P = world size:

  * join-key shuffle  = local K9 partition scatter (qsx_partition_scatter)
                        -> counts all-to-all -> all-to-all(v) of the key and
                        payload columns -> local build / probe;
  * partial aggregate merge (group keys not co-partitioned with the shuffle)
                      = all-gather of the exported state images + local
                        qsx_agg_state_import_merge for hash-table states,
                        reduce-scatter (SUM / MIN / MAX per column, bit-OR of the
                        existence words) of the dense CollisionFreeVector image:
                        rank r keeps the merged keys of finalize partition r
                        (all-reduce variant for callers that want the whole table).

Everything numeric happens behind an ``ops`` object with the same surface as
``quickstep_amd.capi``; the product passes ``quickstep_amd.capi`` itself.
(The gloo/CPU tests pass an adapter over the CPU checker so that the rank
logic — split sizes, offsets, merge order — is exercised without a GPU.)
"""
import torch
import torch.distributed as dist

from . import types as T


def _splits(offsets_host):
    return [int(offsets_host[i + 1] - offsets_host[i]) for i in range(len(offsets_host) - 1)]


class CapiGroup:
    """The ranks of a job as the C ABI sees them: one qsx_comm_t per rank (quickstep_amd.capi.Comm — RCCL bound inside
    libqsx.so, or whatever QSX_RCCL_LIBRARY names).  Pass one wherever this module takes `group=` and every exchange goes
    through the C ABI's own entry points — qsx_exchange_counts, qsx_alltoallv, qsx_allgather, qsx_bitmap_allreduce_or,
    qsx_agg_reduce_scatter, qsx_agg_allgather_merge: the calls a compiled host makes (quickstep_amd/host) — instead of
    torch.distributed.  `group=None` / a torch ProcessGroup keeps torch.distributed (backend "nccl" = RCCL)."""

    def __init__(self, comm):
        self.comm, self.world, self.rank = comm, comm.world, comm.rank

    @classmethod
    def from_torch_group(cls, ops, device, group=None):
        """A communicator over the ranks of an initialised torch process group: rank 0 makes the id, the process group's
        own (host-side) broadcast carries it — torch.distributed is the control plane here, the data goes through the C ABI."""
        world, rank = dist.get_world_size(group), dist.get_rank(group)
        box = [ops.Comm.unique_id() if rank == 0 else None]
        dist.broadcast_object_list(box, src=dist.get_global_rank(group, 0) if group is not None else 0, group=group)
        return cls(ops.Comm(world, rank, box[0]))


def world_size(group=None):
    if isinstance(group, CapiGroup):
        return group.world
    return dist.get_world_size(group) if dist.is_available() and dist.is_initialized() else 1


def rank_of(group=None):
    if isinstance(group, CapiGroup):
        return group.rank
    return dist.get_rank(group) if dist.is_available() and dist.is_initialized() else 0


class _Transport:
    """The collectives this file issues, by where the tensors live and what the group can move.

    torch process group, backend "nccl" (= RCCL over xGMI): device tensors handed through unchanged.  A CapiGroup: the C
    ABI's entry points.  A process group that cannot take device memory (gloo) gets *host staging*: device tensor -> host
    copy -> collective -> device copy — how the rank logic runs with the real kernels when ranks cannot have a GPU each
    (rank processes sharing the one GPU of a test box, where RCCL refuses the duplicate device); it moves the same bytes
    in the same order, only slower."""

    @staticmethod
    def _staged(tensor, group):
        return tensor.is_cuda and dist.get_backend(group) != "nccl"

    @classmethod
    def all_to_all_single(cls, out, inp, output_split_sizes=None, input_split_sizes=None, group=None):
        if isinstance(group, CapiGroup):
            each_in, each_out = inp.numel() // group.world, out.numel() // group.world
            group.comm.alltoallv(inp, input_split_sizes or [each_in] * group.world, output_split_sizes or [each_out] * group.world, out=out)
            return
        if not cls._staged(inp, group):
            dist.all_to_all_single(out, inp, output_split_sizes=output_split_sizes, input_split_sizes=input_split_sizes, group=group)
            return
        host = torch.empty(out.shape, dtype=out.dtype)
        dist.all_to_all_single(host, inp.cpu(), output_split_sizes=output_split_sizes, input_split_sizes=input_split_sizes, group=group)
        out.copy_(host)

    @classmethod
    def all_gather_into_tensor(cls, out, inp, group=None):
        if isinstance(group, CapiGroup):
            group.comm.allgather(inp.contiguous(), out=out)
            return
        if not cls._staged(inp, group):
            dist.all_gather_into_tensor(out, inp, group=group)
            return
        host = torch.empty(out.shape, dtype=out.dtype)
        dist.all_gather_into_tensor(host, inp.cpu().contiguous(), group=group)
        out.copy_(host)

    @classmethod
    def reduce_scatter_tensor(cls, out, inp, op=dist.ReduceOp.SUM, group=None):
        if isinstance(group, CapiGroup):
            raise NotImplementedError("the C ABI reduces whole states (qsx_agg_reduce_scatter), not tensors")
        if not cls._staged(inp, group):
            dist.reduce_scatter_tensor(out, inp, op=op, group=group)
            return
        host = torch.empty(out.shape, dtype=out.dtype)
        dist.reduce_scatter_tensor(host, inp.cpu(), op=op, group=group)
        out.copy_(host)

    @classmethod
    def all_reduce(cls, tensor, op=dist.ReduceOp.SUM, group=None):
        if isinstance(group, CapiGroup):
            raise NotImplementedError("the C ABI has no tensor all-reduce (bit vectors: qsx_bitmap_allreduce_or)")
        if not cls._staged(tensor, group):
            dist.all_reduce(tensor, op=op, group=group)
            return
        host = tensor.cpu()
        dist.all_reduce(host, op=op, group=group)
        tensor.copy_(host)


xfer = _Transport


class _PhaseTimer:
    """Optional HIP-event brackets around the device phases of a plan (bench.py --config c4 / c5): phase(name) is a context
    manager that records an event pair on the current stream while `enabled`; read() sums the pairs per phase."""
    enabled = False
    _pairs = []

    class _Span:
        def __init__(self, name):
            self.name = name

        def __enter__(self):
            if _PhaseTimer.enabled and torch.cuda.is_available():
                self.start = torch.cuda.Event(enable_timing=True)
                self.start.record()
            return self

        def __exit__(self, *exc):
            if _PhaseTimer.enabled and torch.cuda.is_available():
                end = torch.cuda.Event(enable_timing=True)
                end.record()
                _PhaseTimer._pairs.append((self.name, self.start, end))
            return False

    @classmethod
    def phase(cls, name):
        return cls._Span(name)

    @classmethod
    def read(cls):
        torch.cuda.synchronize()
        out = {}
        for name, a, b in cls._pairs:
            out[name] = out.get(name, 0.0) + a.elapsed_time(b)
        cls._pairs = []
        return out


phases = _PhaseTimer


def exchange_counts(send_counts, group=None):
    """send_counts: int64[P] on the compute device.  Returns int64[P] recv counts."""
    if isinstance(group, CapiGroup):
        return group.comm.exchange_counts(send_counts)
    recv = torch.empty_like(send_counts)
    xfer.all_to_all_single(recv, send_counts, group=group)
    return recv


def shuffle_by_key(ops, keys, cols, group=None):
    """Repartition rows across ranks by the hash partition of ``keys``.

    cols: list of equally long 1-D tensors travelling with the rows (include
    ``keys`` itself if the receiver needs it).  Returns (received columns,
    recv_counts list).  Rows from rank r arrive before rows from rank r + 1 and
    keep their local order (the scatter is stable)."""
    world = world_size(group)
    with phases.phase("partition_scatter"):
        scattered, offsets = ops.partition_scatter(keys, world, cols)
    with phases.phase("exchange"):
        send_counts_dev = offsets[1:] - offsets[:-1]
        recv_counts_dev = exchange_counts(send_counts_dev, group)
        send_splits = send_counts_dev.cpu().tolist()          # host sync: all_to_all_single needs host split sizes
        recv_splits = recv_counts_dev.cpu().tolist()
        total = int(sum(recv_splits))
        if world == 1:
            return scattered, recv_splits                      # nothing leaves the rank: the scattered columns are the result
        received = []
        for col in scattered:
            out = torch.empty(total, dtype=col.dtype, device=col.device)
            xfer.all_to_all_single(out, col, output_split_sizes=recv_splits, input_split_sizes=send_splits, group=group)
            received.append(out)
    return received, recv_splits


def _checked_count(count, out_pairs):
    """qsx_join_probe keeps counting past the capacity it was given and only guards the writes: a count above the
    buffer means pairs were dropped (duplicate build keys, capacity sized for a foreign-key join) — an error, not a
    shorter result."""
    k = int(count.item())
    if k > out_pairs.numel():
        raise RuntimeError(f"join produced {k} pairs but the output buffer holds {out_pairs.numel()}: "
                           "probe again with capacity >= that count (or count first with probe_count)")
    return k


def rank_progression(key_domain, world, rank):
    """(first, last) of the keys k in [min, max] with k & (world-1) == rank, or None when the dense
    flavour does not apply (no statistics, world not a power of two, negative keys — their
    zero-extended hash is not the key value — or an empty partition)."""
    if key_domain is None or world & (world - 1):
        return None
    lo, hi = int(key_domain[0]), int(key_domain[1])
    if lo < 0 or hi < lo:
        return None
    first = lo + ((rank - lo) % world)
    last = hi - ((hi - rank) % world)
    return (first, last) if first <= last else None


class PartitionedHashJoin:
    """lineitem ⋈ orders style partitioned join (BASELINE config 4).

    Build rows and probe rows start block-round-robin on the ranks; both sides
    are shuffled on the join key, then every rank joins its own partition.
    Output pairs carry *global* tuple ids (rank-local tid + the rank's base)."""

    def __init__(self, ops, key_type, est_build_rows_per_rank, group=None, key_domain=None):
        """key_domain = (min, max): exact statistics of the build-side join attribute (TPC-H keys are
        dense).  With a power-of-two world the partition of this rank, key & (P-1) == rank, is the
        progression first, first + P, ... so the rank's table is the strided directly addressed
        flavour (qsx_join_table_create_dense); otherwise the hashed table."""
        self.ops = ops
        self.group = group
        world = world_size(group)
        rank = rank_of(group)
        progression = rank_progression(key_domain, world, rank)
        if progression is None:
            self.table = ops.JoinTable(key_type, est_build_rows_per_rank)
        else:
            self.table = ops.JoinTable(key_type, est_build_rows_per_rank, key_range=progression, key_stride=world)

    def build(self, keys, tid_base, payload=()):
        """payload: further columns of the build relation that travel with the rows (BASELINE config 4: one 8-byte
        column per side); they end up in self.build_payload, in the order of the rows of this rank's partition."""
        self.shuffle_build(keys, tid_base, payload)
        return self.build_received()

    def shuffle_build(self, keys, tid_base, payload=(), with_tids=True):
        """The build side's exchange step alone (K9 scatter, counts, all-to-all(v)); build_received() then inserts what
        arrived.  Split so that a plan can put the probe side's exchange on a second stream next to the build kernel —
        the build runs at the rate of the atomic units and leaves HBM and the links idle (plans.PartitionedJoin.step).
        with_tids=False: the global tuple ids do not travel (a plan whose output relation is written by the probe itself
        never looks at them: 4 of 16 bytes per row less through K9 and over the links)."""
        cols = [keys, *payload]
        if with_tids:
            cols.insert(1, torch.arange(tid_base, tid_base + keys.numel(), dtype=torch.int32, device=keys.device))
        received, _ = shuffle_by_key(self.ops, keys, cols, self.group)
        self.build_keys = received[0]
        self.build_tids = received[1] if with_tids else None   # table stores positions into this column
        self.build_payload = received[2:] if with_tids else received[1:]
        self.shuffled_bytes = sum(c.numel() * c.element_size() for c in received)

    def build_received(self):
        with phases.phase("build"):
            self.table.clear()
            self.table.build(self.build_keys)
        return self.build_keys.numel()

    def shuffle_probe(self, keys, tid_base, payload=(), with_tids=True):
        """The probe side's exchange step alone; probe_output_received() joins what arrived."""
        cols = [keys, *payload]
        if with_tids:
            cols.insert(1, torch.arange(tid_base, tid_base + keys.numel(), dtype=torch.int32, device=keys.device))
        received, _ = shuffle_by_key(self.ops, keys, cols, self.group)
        self.probe_keys = received[0]
        self.probe_tids = received[1] if with_tids else None
        self.probe_payload = received[2:] if with_tids else received[1:]
        self.shuffled_bytes += sum(c.numel() * c.element_size() for c in received)
        return received

    def probe(self, keys, tid_base, capacity=None, payload=()):
        tids = torch.arange(tid_base, tid_base + keys.numel(), dtype=torch.int32, device=keys.device)
        received, _ = shuffle_by_key(self.ops, keys, [keys, tids, *payload], self.group)
        rkeys, rtids = received[0], received[1]
        self.probe_keys = rkeys
        self.probe_payload = received[2:]
        self.shuffled_bytes += sum(c.numel() * c.element_size() for c in received)
        # how many rows arrive depends on the data: a foreign-key probe (every key matches at most one build row)
        # needs room for all of them, never less; without a bound from the caller the pairs are counted first
        # (duplicate build keys can yield more pairs than probe rows)
        if capacity is not None:
            capacity = max(int(capacity), rkeys.numel())
        else:
            capacity = max(int(self.table.probe_count(rkeys).item()), 1)
        with phases.phase("probe"):
            out_p, out_b, count = self.table.probe(rkeys, capacity=capacity)
        return rtids, self.build_tids, out_p, out_b, count

    def probe_output(self, keys, tid_base, payload=()):
        """probe() + materialize_payload() in one pass where the table offers it (qsx_join_probe_project_blocks: the probe
        writes (join key, build payload columns..., probe payload columns...) itself — no pair list, no gathers); the pair
        list and K5 gathers otherwise (the CPU checker of the gloo tests)."""
        if not hasattr(self.table, "probe_project_blocks"):
            _, _, out_p, out_b, count = self.probe(keys, tid_base, capacity=None, payload=payload)   # counts first: duplicates
            return self.materialize_payload(out_p, out_b, count)
        self.shuffle_probe(keys, tid_base, payload, with_tids=False)
        return self.probe_output_received()

    def probe_output_received(self):
        rkeys = self.probe_keys
        with phases.phase("probe"):
            room = max(rkeys.numel(), 1)
            for _ in range(2):
                outs, count = self.table.probe_project_blocks([rkeys], [[rkeys]] + [[c] for c in self.probe_payload],
                                                              [[c] for c in self.build_payload], capacity=room)
                k = int(count.item())
                if k <= room:
                    break
                room = k                    # duplicate build keys: once more with the exact capacity
        npay = len(self.probe_payload)
        cols = [outs[0]] + outs[1 + npay:] + outs[1:1 + npay]
        return [c[:k] for c in cols]

    def materialize_payload(self, out_p, out_b, count):
        """The join's output relation on this rank: (join key, build payload columns..., probe payload columns...) of
        every pair — K5 gathers on the columns that arrived with the shuffle (HashJoinOperator.cpp:526-541 builds the
        same columns through ScalarAttribute::getAllValuesForJoin)."""
        k = _checked_count(count, out_p)
        p, b = out_p[:k], out_b[:k]
        with phases.phase("materialise"):
            return ([self.ops.gather(self.probe_keys, p)] + [self.ops.gather(c, b) for c in self.build_payload]
                    + [self.ops.gather(c, p) for c in self.probe_payload])

    def materialize(self, probe_tids, build_tids, out_p, out_b, count):
        """Global (probe_tid, build_tid) pairs of this partition (K5 gathers)."""
        k = _checked_count(count, out_p)
        return self.ops.gather(probe_tids, out_p[:k]), self.ops.gather(build_tids, out_b[:k])


class BroadcastHashJoin:
    """Broadcast join: the build side is small, so every rank all-gathers ALL build rows and probes its
    own probe rows locally — no probe-side shuffle.  The reference's plan for a build relation without
    a partition scheme under a partitioned probe (BuildHashOperator.hpp:99, 146-152:
    is_broadcast_join_ inserts every build block into every partition's table).  Same result surface
    as PartitionedHashJoin: pairs carry global tuple ids."""

    def __init__(self, ops, key_type, est_build_rows_total, group=None, key_domain=None):
        self.ops = ops
        self.group = group
        self.table = ops.JoinTable(key_type, est_build_rows_total, key_range=key_domain) if key_domain is not None \
            else ops.JoinTable(key_type, est_build_rows_total)

    def build(self, keys, tid_base):
        world = world_size(self.group)
        # (row count, tid base) of every rank: one collective, one host synchronisation
        mine_meta = torch.tensor([keys.numel(), tid_base], dtype=torch.int64, device=keys.device)
        meta = torch.empty(2 * world, dtype=torch.int64, device=keys.device)
        xfer.all_gather_into_tensor(meta, mine_meta, group=self.group)
        meta = meta.cpu().tolist()
        sizes, bases = meta[0::2], meta[1::2]
        self.table.clear()
        contiguous = len(set(sizes)) == 1 and all(bases[r] == bases[0] + r * sizes[0] for r in range(world))
        if contiguous and sizes[0] > 0:
            # equal shares whose global tids follow one another (the block-round-robin layout): the gathered buffer IS
            # the build relation in tid order -> one gather into one tensor, one build
            everything = torch.empty(world * sizes[0], dtype=keys.dtype, device=keys.device)
            xfer.all_gather_into_tensor(everything, keys.contiguous(), group=self.group)
            self.table.build(everything, base_tid=bases[0])
            return sum(sizes)
        pad = max(max(sizes), 1)
        mine = torch.zeros(pad, dtype=keys.dtype, device=keys.device)
        mine[:keys.numel()] = keys
        gathered = torch.empty(world * pad, dtype=keys.dtype, device=keys.device)
        xfer.all_gather_into_tensor(gathered, mine, group=self.group)
        for r in range(world):                              # the stored reference is the GLOBAL build tid
            if sizes[r]:
                self.table.build(gathered[r * pad: r * pad + sizes[r]], base_tid=bases[r])
        return sum(sizes)

    def probe(self, keys, tid_base, capacity=None):
        out_p, out_b, count = self.table.probe(keys, capacity=capacity, probe_base_tid=tid_base)
        return None, None, out_p, out_b, count

    def materialize(self, probe_tids, build_tids, out_p, out_b, count):
        k = _checked_count(count, out_p)
        return out_p[:k], out_b[:k]                         # already global


def merge_agg_state_images(ops, state, group=None):
    """Merge the partial aggregation states of all ranks into every rank's state
    (counterpart of merging the thread-private tables at finalize,
    storage/AggregationOperationState.cpp:925-948).  Hash-table images are
    all-gathered and merged locally (Q1-sized: a few KiB per rank)."""
    if isinstance(group, CapiGroup):
        group.comm.agg_allgather_merge(state)      # sizes, images and the merges: all inside qsx_agg_allgather_merge
        return state
    world = world_size(group)
    rank = rank_of(group)
    device = state.device if hasattr(state, "device") else None
    image = state.export(device)
    # a table that outgrew its estimate has a bigger image than its peers': exchange the sizes, pad to the largest
    words = torch.tensor([image.numel()], dtype=torch.int64, device=image.device)
    all_words = torch.empty(world, dtype=torch.int64, device=image.device)
    xfer.all_gather_into_tensor(all_words, words, group=group)
    all_words = all_words.cpu().tolist()
    pad = max(all_words)
    if image.numel() < pad:
        image = torch.cat([image, image.new_zeros(pad - image.numel())])
    gathered = torch.empty(world * pad, dtype=image.dtype, device=image.device)
    xfer.all_gather_into_tensor(gathered, image, group=group)
    for r in range(world):
        if r != rank:
            state.import_merge(gathered[r * pad: r * pad + all_words[r]])
    return state


def _allreduce_or(words, group=None):
    """Bit-OR all-reduce of an int64 tensor in place.  RCCL has no bitwise reductions (SUM / PROD / MIN / MAX / AVG only:
    ReduceOp.BOR raises on the "nccl" backend), so there the words are all-gathered and OR-ed locally; the existence map
    of a CollisionFreeVector is 1 bit per key, a sixty-fourth of one state column."""
    if isinstance(group, CapiGroup):
        return group.comm.bitmap_allreduce_or(words)
    if dist.get_backend(group) != "nccl":
        xfer.all_reduce(words, op=dist.ReduceOp.BOR, group=group)
        return words
    world = dist.get_world_size(group)
    gathered = torch.empty(world * words.numel(), dtype=words.dtype, device=words.device)
    xfer.all_gather_into_tensor(gathered, words.contiguous(), group=group)
    gathered = gathered.view(world, words.numel())
    acc = gathered[0].clone()
    for r in range(1, world):
        acc |= gathered[r]
    words.copy_(acc)
    return words


def dense_partition_range(num_entries, world, rank):
    """Key range [begin, end) that finalize partition `rank` of `world` owns in a CollisionFreeVector state
    (CollisionFreeVectorTable.hpp:192-208; qsx_agg_finalize uses the same split)."""
    return _dense_split(num_entries, world, rank)[:2]


def _dense_split(num_entries, world, rank):
    """(begin, end, first_word, last_word, first_mask, last_mask) from the C ABI's own statement of the split
    (qsx_agg_dense_partition_range: host arithmetic, runs without a device) — the torch.distributed path below moves
    exactly the ranges and applies exactly the masks qsx_agg_reduce_scatter does."""
    from . import capi
    return capi.dense_partition_range(num_entries, world, rank)


def reduce_scatter_dense_agg_image(image, exist_words, num_entries, int_col_mask, num_cols, group=None, min_max_cols=None):
    """Reduce-scatter of the ranks' CollisionFreeVector images: afterwards rank r holds the MERGED state of the keys of
    finalize partition r (dense_partition_range) and nothing else — the returned image is zero / identity outside that
    range — so every rank finalizes ITS partition of the result (qsx_agg_finalize(partition=rank, num_partitions=world))
    and no rank ever holds, or receives, the whole merged table.  Per rank this moves 1/world of what the all-reduce
    moves.  Columns: reduce_scatter_tensor with SUM (int64 / f64) or MIN / MAX; existence bits: every rank sends each
    peer the words covering that peer's key range (all_to_all_single), the receiver ORs them.

    image: int64[exist_words + num_cols * num_entries] as exported by the state.  Returns a new image of the same layout
    to be imported into a CLEARED state (state.clear(); state.import_merge(result))."""
    if isinstance(group, CapiGroup):
        raise NotImplementedError("a CapiGroup reduces the state itself: reduce_scatter_dense_state -> qsx_agg_reduce_scatter")
    world = world_size(group)
    rank = rank_of(group)
    length = _dense_split(num_entries, world, 0)[1]          # every range but the last ones: ceil(entries / world) keys
    begin, end = dense_partition_range(num_entries, world, rank)
    out = torch.zeros_like(image)
    padded = world * length
    for col in range(num_cols):
        seg = image[exist_words + col * num_entries: exist_words + (col + 1) * num_entries]
        is_int = bool((int_col_mask >> col) & 1) or bool(min_max_cols and col in min_max_cols)
        op = dist.ReduceOp.SUM
        fill = 0
        if min_max_cols and col in min_max_cols:
            op = dist.ReduceOp.MIN if min_max_cols[col] == "min" else dist.ReduceOp.MAX
            fill = torch.iinfo(torch.int64).max if min_max_cols[col] == "min" else torch.iinfo(torch.int64).min
        src = seg if is_int else seg.view(torch.float64)
        send = src.new_full((padded,), fill) if is_int else src.new_zeros(padded)
        send[:num_entries] = src
        mine = torch.empty(length, dtype=send.dtype, device=send.device)
        xfer.reduce_scatter_tensor(mine, send, op=op, group=group)
        dst = out[exist_words + col * num_entries + begin: exist_words + col * num_entries + end]
        if end > begin:
            dst.copy_((mine if is_int else mine.view(torch.int64))[: end - begin])
        if min_max_cols and col in min_max_cols:
            # outside the owned range: the accumulator's identity, what a cleared state holds
            full = out[exist_words + col * num_entries: exist_words + (col + 1) * num_entries]
            full[:begin] = fill
            full[end:] = fill
    # existence words covering every peer's range (the same word may go to two neighbours: their ranges share it)
    ranges = [_dense_split(num_entries, world, r) for r in range(world)]
    splits = [r[3] - r[2] for r in ranges]
    send = torch.cat([image[r[2]: r[3]] for r in ranges]) if sum(splits) else image[:0]
    my_words = splits[rank]
    recv = torch.empty(world * my_words, dtype=image.dtype, device=image.device)
    xfer.all_to_all_single(recv, send, output_split_sizes=[my_words] * world, input_split_sizes=splits, group=group)
    if my_words:
        acc = recv[:my_words].clone()
        for r in range(1, world):
            acc |= recv[r * my_words: (r + 1) * my_words]
        _, _, first_word, last_word, first_mask, last_mask = ranges[rank]
        # bits of neighbouring partitions that share the boundary words are dropped: the range is exactly [begin, end)
        signed = lambda m: m - (1 << 64) if m >> 63 else m        # noqa: E731  (uint64 mask as the int64 the image holds)
        acc[0] &= signed(first_mask)
        acc[-1] &= signed(last_mask)
        out[first_word:last_word] = acc
    return out


def reduce_scatter_dense_state(state, device, group=None):
    """Merge the dense (CollisionFreeVector) states of all ranks so that rank r holds — and can finalize — the groups of key
    range r: export, reduce-scatter every column the way ITS accumulator combines (qsx_agg_state_image_layout: f64 +,
    int64 +, MIN, MAX), OR the existence bits of the owned range, clear, import.  Returns the bytes a ring reduce-scatter
    moves per rank.  The caller finalizes with partition = rank, num_partitions = world."""
    dense, exist_words, entries, kinds = state.image_layout()
    if not dense:
        raise ValueError("reduce_scatter_dense_state needs a COLLISION_FREE state (hash states: merge_agg_state_images)")
    world = world_size(group)
    if isinstance(group, CapiGroup):
        group.comm.agg_reduce_scatter(state)
        return (len(kinds) * entries * 8 + exist_words * 8) * (world - 1) // world
    int_mask, min_max = 0, {}
    for c, kind in enumerate(kinds):
        if kind != T.ACC_SUM_F64:
            int_mask |= 1 << c
        if kind in (T.ACC_MIN_I64, T.ACC_MAX_I64):
            min_max[c] = "min" if kind == T.ACC_MIN_I64 else "max"
    image = state.export(device)
    reduced = reduce_scatter_dense_agg_image(image, exist_words, entries, int_mask, len(kinds), group=group, min_max_cols=min_max or None)
    del image
    state.clear()
    state.import_merge(reduced)
    return (len(kinds) * entries * 8 + exist_words * 8) * (world - 1) // world


def allreduce_dense_agg_image(image, exist_words, num_entries, int_col_mask, num_cols, group=None, min_max_cols=None):
    """All-reduce a CollisionFreeVector state image in place: bit-OR for the
    existence words, integer SUM for count / integer columns, f64 SUM for the
    rest.  image: int64[exist_words + num_cols * num_entries].
    min_max_cols: {column: "min" | "max"} for MIN / MAX accumulators — those columns hold int64 words
    (the value itself or the order-preserving image of a double) and reduce with integer MIN / MAX."""
    _allreduce_or(image[:exist_words], group)
    for col in range(num_cols):
        seg = image[exist_words + col * num_entries: exist_words + (col + 1) * num_entries]
        if min_max_cols and col in min_max_cols:
            xfer.all_reduce(seg, op=dist.ReduceOp.MIN if min_max_cols[col] == "min" else dist.ReduceOp.MAX, group=group)
        elif (int_col_mask >> col) & 1:
            xfer.all_reduce(seg, op=dist.ReduceOp.SUM, group=group)
        else:
            xfer.all_reduce(seg.view(torch.float64), op=dist.ReduceOp.SUM, group=group)
    return image
