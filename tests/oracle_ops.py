"""An `ops` object with the surface of quickstep_amd.capi that quickstep_amd/distributed.py and quickstep_amd/plans.py call,
implemented on CPU tensors by the CPU checker (oracle/pyoracle.py) and numpy: the rank logic of the multi-GPU plans —
split sizes, offsets, tid bases, merge order, key-range ownership — runs under gloo without a GPU.  TEST INFRASTRUCTURE."""
import numpy as np
import torch

from oracle import pyoracle as O
from quickstep_amd import types as T


def _bm_in(t):
    return None if t is None else np.ascontiguousarray(t.numpy().view(np.uint64))


def _bm_out(words):
    return torch.from_numpy(words.view(np.int64).copy())


class OracleJoinTable:
    def __init__(self, key_type, est, key_range=None, key_stride=1):
        self.key_type, self.est = key_type, est
        self.key_range, self.key_stride = key_range, key_stride
        self.t = O.JoinTable(key_type, est)
        self.live_keys = []          # what the builds inserted (ExactLipFilter.build_from_table reads them off the table)

    def clear(self):
        self.t = O.JoinTable(self.key_type, self.est)
        self.live_keys = []

    def build(self, keys, base_tid=0, filter_bitmap=None):
        k = keys.numpy()
        if self.key_range is not None and k.size:
            # the dense flavour's precondition: every build key is a member of the table's progression
            live = k if filter_bitmap is None else k[O.bools_from_bitmap(_bm_in(filter_bitmap), k.size)]
            if live.size:
                assert live.min() >= self.key_range[0] and live.max() <= self.key_range[1]
                assert ((live.astype(np.int64) - self.key_range[0]) % self.key_stride == 0).all()
        self.t.build(k, base_tid=base_tid, filter_bitmap=_bm_in(filter_bitmap))
        self.live_keys.append(k if filter_bitmap is None else k[O.bools_from_bitmap(_bm_in(filter_bitmap), k.size)])

    def probe_count(self, keys, filter_bitmap=None):
        return torch.tensor([self.t.probe(keys.numpy(), filter_bitmap=_bm_in(filter_bitmap))[0].size], dtype=torch.int64)

    def probe(self, keys, capacity=None, probe_base_tid=0, filter_bitmap=None):
        p, b = self.t.probe(keys.numpy(), probe_base_tid=probe_base_tid, filter_bitmap=_bm_in(filter_bitmap))
        return torch.from_numpy(p), torch.from_numpy(b), torch.tensor([p.size], dtype=torch.int64)

    def probe_exists(self, keys, anti=False, filter_bitmap=None):
        bm = self.t.probe_exists(keys.numpy(), anti=anti, filter_bitmap=_bm_in(filter_bitmap))
        return _bm_out(bm), torch.tensor([O.bitmap_count(bm, keys.numel())], dtype=torch.int64)


class ExactLipFilter:
    """BitVectorExactFilter (utility/lip_filter/BitVectorExactFilter.hpp:150-176): bit = value - min, LSB-first words."""

    def __init__(self, kind, cardinality, min_value=0, is_anti=False):
        assert kind == T.LIP_BITVECTOR_EXACT and not is_anti
        self.card, self.min = cardinality, min_value
        self.bits = np.zeros((cardinality + 63) // 64, dtype=np.uint64)

    def clear(self):
        self.bits[:] = 0

    def build(self, keys, filter_bitmap=None):
        k = keys.numpy().astype(np.int64) - self.min
        if filter_bitmap is not None:
            k = k[O.bools_from_bitmap(_bm_in(filter_bitmap), k.size)]
        k = k[(k >= 0) & (k < self.card)]
        np.bitwise_or.at(self.bits, k >> 6, np.uint64(1) << (k & 63).astype(np.uint64))

    def build_from_table(self, table, num_new_keys=-1):
        """The mirror of qsx_lip_build_from_join_table: every key a directly addressed table holds; declined for a hashed one."""
        if table.key_range is None or table.key_stride != 1:
            return False
        for k in table.live_keys:
            self.build(torch.from_numpy(np.ascontiguousarray(k)))
        return True

    def probe(self, keys, in_bitmap=None):
        k = keys.numpy().astype(np.int64) - self.min
        inside = (k >= 0) & (k < self.card)
        kk = np.where(inside, k, 0)
        hit = inside & (((self.bits[kk >> 6] >> (kk & 63).astype(np.uint64)) & np.uint64(1)) != 0)
        if in_bitmap is not None:
            hit &= O.bools_from_bitmap(_bm_in(in_bitmap), k.size)
        bm = O.bitmap_from_bools(hit)
        return _bm_out(bm), torch.tensor([int(hit.sum())], dtype=torch.int64)

    def export(self, device):
        return torch.from_numpy(self.bits.view(np.int64).copy())

    def merge_or(self, words):
        self.bits |= words.numpy().view(np.uint64)


class DenseSumState:
    """CollisionFreeVector state with COUNT(*) + SUM(expression) in the image layout of qsx_agg_state_export:
    [existence words, LSB-first][row counts int64][sums f64] — update / export / import_merge / finalize in numpy."""
    device = None

    def __init__(self, cfg):
        self.cfg = cfg
        self.entries = int(cfg.num_entries)
        self.clear()

    def clear(self):
        self.count = np.zeros(self.entries, dtype=np.int64)
        self.sum = np.zeros(self.entries, dtype=np.float64)

    def _accumulate(self, cols):
        cfg = self.cfg
        instrs = [(cfg.instrs[i].op, cfg.instrs[i].dst, cfg.instrs[i].a, cfg.instrs[i].b) for i in range(cfg.num_instrs)]
        value = O.eval_expression(cols, instrs, list(cfg.consts), cfg.aggs[0].arg)
        keys = cols[cfg.key_column[0]].astype(np.int64)
        self.count += np.bincount(keys, minlength=self.entries)
        self.sum += np.bincount(keys, weights=value, minlength=self.entries)

    def update(self, cols, n=None):
        self._accumulate([c.numpy()[:n] for c in cols])

    def update_coded(self, codes, dictionaries, n=None):
        self._accumulate([d.numpy()[c.numpy()[:n]] for c, d in zip(codes, dictionaries)])

    def image_layout(self):
        return True, (self.entries + 63) // 64, self.entries, [T.ACC_SUM_I64, T.ACC_SUM_F64]

    def export(self, device):
        exist = np.zeros((self.entries + 63) // 64, dtype=np.uint64)
        k = np.nonzero(self.count)[0]
        np.bitwise_or.at(exist, k >> 6, np.uint64(1) << (k & 63).astype(np.uint64))
        return torch.from_numpy(np.concatenate([exist.view(np.int64), self.count, self.sum.view(np.int64)]))

    def import_merge(self, image):
        w = (self.entries + 63) // 64
        img = image.numpy()
        self.count += img[w: w + self.entries]
        self.sum += img[w + self.entries: w + 2 * self.entries].view(np.float64)

    def finalize(self, device, partition=0, num_partitions=1, capacity=None):
        length = (self.entries + num_partitions - 1) // num_partitions
        lo, hi = min(partition * length, self.entries), min((partition + 1) * length, self.entries)
        k = np.nonzero(self.count[lo:hi])[0] + lo
        return ([torch.from_numpy(k.astype(np.int32))], [torch.from_numpy(self.sum[k])], [torch.zeros(k.size, dtype=torch.uint8)],
                torch.tensor([k.size], dtype=torch.int64))


class OracleOps:
    JoinTable = OracleJoinTable
    LipFilter = ExactLipFilter
    AggState = DenseSumState

    @staticmethod
    def partition_scatter(keys, num_partitions, cols):
        k = keys.numpy()
        offs = O.partition_offsets(k, num_partitions)
        return [torch.from_numpy(O.partition_scatter(k, num_partitions, c.numpy())) for c in cols], torch.from_numpy(offs)

    @staticmethod
    def gather(src, tids):
        return torch.from_numpy(O.gather(src.numpy(), tids.numpy().astype(np.int32)))

    @staticmethod
    def select_cmp(col, op, literal, filter_bitmap=None):
        bm = O.select_cmp(col.numpy(), op, literal, filter_bitmap=_bm_in(filter_bitmap))
        return _bm_out(bm), torch.tensor([O.bitmap_count(bm, col.numel())], dtype=torch.int64)

    @staticmethod
    def compact_gather(cols, bitmap, n):
        bm = _bm_in(bitmap)
        outs = []
        for c in cols:
            sel = O.compact_gather(np.ascontiguousarray(c.numpy()[:n]), bm)
            full = np.zeros(n, dtype=sel.dtype)          # capi hands back n-row buffers with the selected rows in front
            full[:sel.size] = sel
            outs.append(torch.from_numpy(full))
        return outs, torch.tensor([O.bitmap_count(bm, n)], dtype=torch.int64)

    @staticmethod
    def sort_top_k(key_cols, k, descending=None):
        perm = O.sort_permutation([c.numpy() for c in key_cols], descending)
        return torch.from_numpy(perm[:k].copy())
