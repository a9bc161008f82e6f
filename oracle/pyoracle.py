"""ctypes binding of the CPU ORACLE (oracle/libqsx_oracle.so).

TEST INFRASTRUCTURE ONLY — importable from tests/, __graft_entry__.smoke() and
the cpu_baseline leg of bench.py; nothing under quickstep_amd/ imports it.
Works on numpy arrays in host memory.
"""
import ctypes as C
import os
import subprocess
import sys

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(_HERE))
from quickstep_amd import types as T  # noqa: E402  (descriptor structs only; loads no library)

LIB_PATH = os.path.join(_HERE, "libqsx_oracle.so")
REF_PIN = os.path.join(_HERE, "_ref", "ref_pin")


def build():
    subprocess.run(["make", "-C", _HERE, "-s"], check=True)


if not os.path.exists(LIB_PATH):
    build()
_lib = C.CDLL(LIB_PATH)

_vp, _i64, _i32, _int, _u64, _sz = C.c_void_p, C.c_int64, C.c_int32, C.c_int, C.c_uint64, C.c_size_t
_pp = C.POINTER(C.c_void_p)


class CompressedInfo(C.Structure):
    _fields_ = [("kind", C.c_int), ("code_width", C.c_int), ("num_codes", C.c_uint32)]


class CodePredicate(C.Structure):
    _fields_ = [("result", C.c_int), ("comp", C.c_int), ("first", C.c_uint32), ("second", C.c_uint32)]


class JoinBenchResult(C.Structure):
    _fields_ = [("build_seconds", C.c_double), ("probe_seconds", C.c_double), ("matches", C.c_int64),
                ("checksum", C.c_uint64)]


class PartitionedJoinResult(C.Structure):
    _fields_ = [("repartition_seconds", C.c_double), ("build_seconds", C.c_double), ("probe_seconds", C.c_double),
                ("output_rows", C.c_int64), ("violations", C.c_int64), ("checksum", C.c_uint64)]


class Q3Inputs(C.Structure):
    _fields_ = [("c_custkey", C.c_void_p), ("c_mktsegment", C.c_void_p), ("n_customer", C.c_int64),
                ("o_orderkey", C.c_void_p), ("o_custkey", C.c_void_p), ("o_orderdate", C.c_void_p), ("n_orders", C.c_int64),
                ("l_orderkey", C.c_void_p), ("l_shipdate", C.c_void_p), ("l_extendedprice", C.c_void_p),
                ("l_discount", C.c_void_p), ("n_lineitem", C.c_int64), ("customers_total", C.c_int64),
                ("orders_total", C.c_int64), ("segment", C.c_int32), ("date_cut", C.c_int32)]


class Q3Result(C.Structure):
    _fields_ = [("customer_seconds", C.c_double), ("orders_seconds", C.c_double), ("lineitem_seconds", C.c_double),
                ("finalize_seconds", C.c_double), ("total_seconds", C.c_double), ("qualifying_orders", C.c_int64),
                ("pairs", C.c_int64), ("groups", C.c_int64), ("top_revenue", C.c_double * 10), ("top_orderkey", C.c_int32 * 10)]


for _name, _res, _args in [
    ("qso_sizeof_agg_config", _sz, []),
    ("qso_hash_scalar", _u64, [_int, _vp]),
    ("qso_combine_hashes", _u64, [_u64, _u64]),
    ("qso_next_prime", _u64, [_u64]),
    ("qso_prev_prime", _u64, [_u64]),
    ("qso_partition_id", _u64, [_u64, _u64]),
    ("qso_select_cmp", None, [_int, _vp, _i64, _int, _vp, _vp, _vp]),
    ("qso_select_cmp_sorted", None, [_int, _vp, _i64, _int, _vp, _vp, _vp]),
    ("qso_eval_expression", None, [_int, _pp, C.POINTER(_i32), _int, C.POINTER(T.ExprInstr), C.POINTER(C.c_double), T.Operand, _i64, _vp]),
    ("qso_select_cmp_char", None, [_vp, _int, _i64, _int, C.c_char_p, _int, _vp, _vp]),
    ("qso_bitmap_count", _i64, [_vp, _i64]),
    ("qso_compact_gather", _i64, [_int, _vp, _vp, _i64, _vp]),
    ("qso_bitmap_to_tids", _i64, [_vp, _i64, _i32, _vp]),
    ("qso_gather", None, [_int, _vp, _vp, _i64, _vp]),
    ("qso_join_table_create", _vp, [_int, _i64]),
    ("qso_join_table_destroy", None, [_vp]),
    ("qso_join_table_info", None, [_vp, C.POINTER(_u64)]),
    ("qso_join_build", None, [_vp, _vp, _i64, _u64, _i32, _vp]),
    ("qso_join_probe", _i64, [_vp, _vp, _i64, _i32, _vp, _vp, _vp, _vp, _i64]),
    ("qso_join_probe_exists", None, [_vp, _vp, _i64, _vp, _int, _vp]),
    ("qso_cjoin_table_create", _vp, [_int, C.POINTER(_i32), _i64]),
    ("qso_cjoin_table_destroy", None, [_vp]),
    ("qso_cjoin_build", None, [_vp, _pp, _i64, _u64, _i32, _vp]),
    ("qso_cjoin_probe", _i64, [_vp, _pp, _i64, _i32, _vp, _vp, _vp, _i64]),
    ("qso_cjoin_hash_row", _u64, [_vp, _pp, _i64]),
    ("qso_select_cmp_columns", None, [_int, _vp, _vp, _i64, _int, _vp, _vp]),
    ("qso_tids_to_bitmap", None, [_vp, _i64, _i32, _i64, _vp]),
    ("qso_compress_column", None, [_int, _vp, _i64, C.POINTER(CompressedInfo), _vp, _vp]),
    ("qso_transform_predicate", None, [C.POINTER(CompressedInfo), _int, _vp, _int, _vp, C.POINTER(CodePredicate)]),
    ("qso_select_codes", None, [_int, _vp, _i64, _int, C.c_uint32, C.c_uint32, _vp, _vp]),
    ("qso_decode_codes", None, [_int, _vp, _i64, _vp, _int, _vp]),
    ("qso_sort_permutation", None, [_int, _pp, C.POINTER(_i32), C.POINTER(_i32), _i64, _vp]),
    ("qso_distinct_rows", _i64, [_int, _pp, C.POINTER(_i32), _i64, _vp, _vp]),
    ("qso_agg_state_create", _vp, [C.POINTER(T.AggConfig)]),
    ("qso_agg_state_destroy", None, [_vp]),
    ("qso_agg_update", None, [_vp, _pp, _i64, _vp]),
    ("qso_agg_mark_existence", None, [_vp, _int, _vp, _i64, _vp]),
    ("qso_agg_update_coded", None, [_vp, _pp, _pp, _i64, _vp]),
    ("qso_agg_update_nullable", None, [_vp, _pp, _pp, _i64, _vp]),
    ("qso_agg_merge", None, [_vp, _vp]),
    ("qso_agg_num_groups", _i64, [_vp]),
    ("qso_agg_finalize", _i64, [_vp, _int, _int, _pp, _pp, _pp, _i64]),
    ("qso_lip_filter_create", _vp, [_int, _i64, _i64, _int]),
    ("qso_lip_filter_destroy", None, [_vp]),
    ("qso_lip_build", None, [_vp, _int, _vp, _i64, _vp]),
    ("qso_lip_probe", None, [_vp, _int, _vp, _i64, _vp, _vp]),
    ("qso_partition_offsets", None, [_int, _vp, _i64, _int, _vp]),
    ("qso_partition_scatter_col", None, [_int, _vp, _i64, _int, _int, _vp, _vp]),
    ("qso_bench_join", None, [_int, _vp, _i64, _vp, _i64, _i64, _int, C.POINTER(JoinBenchResult)]),
    ("qso_bench_agg", C.c_double, [C.POINTER(T.AggConfig), _pp, _i64, _i64, _int, _pp]),
    ("qso_bench_select", C.c_double, [_int, _vp, _i64, _int, _vp, _i64, _int, _vp, C.POINTER(_i64)]),
    ("qso_bench_agg_coded", C.c_double, [C.POINTER(T.AggConfig), _pp, _pp, _i64, _i64, _int, _pp]),
    ("qso_bench_partitioned_join", None, [_vp, _vp, _i64, _vp, _vp, _i64, _int, _i64, _int, C.POINTER(PartitionedJoinResult)]),
    ("qso_bench_q3", None, [C.POINTER(Q3Inputs), _i64, _int, C.POINTER(Q3Result)]),
]:
    _f = getattr(_lib, _name)
    _f.restype = _res
    _f.argtypes = _args

assert _lib.qso_sizeof_agg_config() == C.sizeof(T.AggConfig)

_NP_TYPE = {np.dtype(np.int32): T.INT, np.dtype(np.int64): T.LONG, np.dtype(np.float32): T.FLOAT,
            np.dtype(np.float64): T.DOUBLE, np.dtype(np.uint8): T.CHAR}   # CHAR(1)
_C_SCALAR = {T.INT: C.c_int32, T.LONG: C.c_int64, T.FLOAT: C.c_float, T.DOUBLE: C.c_double}


def qtype(a):
    return _NP_TYPE[a.dtype]


def _p(a):
    if a is None:
        return None
    assert a.flags["C_CONTIGUOUS"]
    return C.c_void_p(a.ctypes.data)


def _ptr_array(arrays):
    arr = (C.c_void_p * max(len(arrays), 1))()
    for i, a in enumerate(arrays):
        arr[i] = None if a is None else a.ctypes.data
    return arr


def words(n):
    return (n + 63) // 64


# ---- scalar helpers ---------------------------------------------------------
def hash_scalar(qt, value):
    v = _C_SCALAR[qt](value)
    return _lib.qso_hash_scalar(qt, C.byref(v))


def combine_hashes(a, b):
    return _lib.qso_combine_hashes(a, b)


def next_prime(n):
    return _lib.qso_next_prime(n)


def prev_prime(n):
    return _lib.qso_prev_prime(n)


def partition_id(h, p):
    return _lib.qso_partition_id(h, p)


def ref_combine_hashes(pairs):
    """CombineHashes evaluated by the REFERENCE's own utility/HashPair.hpp (oracle/_ref/ref_pin)."""
    args = [REF_PIN]
    for a, b in pairs:
        args += [str(a), str(b)]
    out = subprocess.run(args, check=True, capture_output=True, text=True).stdout.split()
    return [int(x, 16) for x in out]


# ---- select --------------------------------------------------------------------
def _literal(qt, literal):
    return C.c_int64(literal) if qt == T.DATE else _C_SCALAR[qt](literal)


def select_cmp(col, op, literal, filter_bitmap=None, qt=None):
    """qt=T.DATE: col holds raw DateLit bytes as int64, literal = T.date_raw(...)."""
    n = col.size
    out = np.zeros(max(words(n), 1), dtype=np.uint64)
    qt = qtype(col) if qt is None else qt
    lit = _literal(qt, literal)
    _lib.qso_select_cmp(qt, _p(col), n, op, C.byref(lit), _p(filter_bitmap), _p(out))
    return out


def eval_expression(cols, instrs, consts, result):
    keep = [np.ascontiguousarray(c) for c in cols]
    n = keep[0].size
    out = np.zeros(max(n, 1), dtype=np.float64)
    ptrs = (C.c_void_p * max(len(keep), 1))(*[c.ctypes.data for c in keep])
    types = (C.c_int32 * max(len(keep), 1))(*[_NP_TYPE[c.dtype] for c in keep])
    prog = (T.ExprInstr * max(len(instrs), 1))(*[T.ExprInstr(op, dst, a, b) for op, dst, a, b in instrs])
    cs = (C.c_double * T.MAX_CONSTS)(*list(consts))
    _lib.qso_eval_expression(len(keep), ptrs, types, len(instrs), prog, cs, result, n, _p(out))
    return out[:n]


def select_cmp_char(col, op, literal, filter_bitmap=None):
    """CHAR(width) column (uint8 array of shape (n, width)) OP literal (bytes): strcmpHelper semantics."""
    n, width = col.shape
    out = np.zeros(max(words(n), 1), dtype=np.uint64)
    _lib.qso_select_cmp_char(_p(col), width, n, op, C.c_char_p(literal), len(literal), _p(filter_bitmap), _p(out))
    return out


def select_cmp_sorted(col, op, literal, filter_bitmap=None, qt=None):
    """The same predicate evaluated by binary search on a sorted stripe (SortColumnPredicateEvaluator)."""
    n = col.size
    out = np.zeros(max(words(n), 1), dtype=np.uint64)
    qt = qtype(col) if qt is None else qt
    lit = _literal(qt, literal)
    _lib.qso_select_cmp_sorted(qt, _p(col), n, op, C.byref(lit), _p(filter_bitmap), _p(out))
    return out


def bitmap_count(bitmap, n):
    return _lib.qso_bitmap_count(_p(bitmap), n)


def compact_gather(col, bitmap):
    out = np.empty_like(col)
    k = _lib.qso_compact_gather(col.itemsize, _p(col), _p(bitmap), col.size, _p(out))
    return out[:k]


def bitmap_to_tids(bitmap, n, base_tid=0):
    out = np.empty(max(n, 1), dtype=np.int32)
    k = _lib.qso_bitmap_to_tids(_p(bitmap), n, base_tid, _p(out))
    return out[:k]


def gather(src, tids):
    out = np.empty(tids.size, dtype=src.dtype)
    _lib.qso_gather(src.itemsize, _p(src), _p(tids), tids.size, _p(out))
    return out


def bitmap_from_bools(bools):
    """MSB-first TupleIdSequence words from a boolean array (test helper)."""
    n = bools.size
    padded = np.zeros(words(n) * 64, dtype=np.uint8)
    padded[:n] = bools
    return np.packbits(padded.reshape(-1, 64), axis=1, bitorder="big").view(">u8").astype(np.uint64).reshape(-1)


def bools_from_bitmap(bitmap, n):
    be = bitmap.astype(">u8").view(np.uint8)
    return np.unpackbits(be, bitorder="big")[:n].astype(bool)


# ---- join ------------------------------------------------------------------------
class JoinTable:
    def __init__(self, key_type, est_entries):
        self._h = C.c_void_p(_lib.qso_join_table_create(key_type, est_entries))
        assert self._h.value

    def close(self):
        if self._h is not None:
            _lib.qso_join_table_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:  # interpreter shutdown
            pass

    def info(self):
        out = (C.c_uint64 * 4)()
        _lib.qso_join_table_info(self._h, out)
        return dict(num_slots=out[0], num_buckets=out[1], buckets_allocated=out[2], blob_bytes=out[3])

    def build(self, keys, block_id=0, base_tid=0, filter_bitmap=None):
        _lib.qso_join_build(self._h, _p(keys), keys.size, block_id, base_tid, _p(filter_bitmap))

    def probe(self, keys, capacity=None, probe_base_tid=0, filter_bitmap=None, with_blocks=False):
        n = keys.size
        cap = n if capacity is None else capacity
        while True:
            op = np.empty(max(cap, 1), dtype=np.int32)
            ob = np.empty(max(cap, 1), dtype=np.int32)
            blocks = np.empty(max(cap, 1), dtype=np.uint64) if with_blocks else None
            k = _lib.qso_join_probe(self._h, _p(keys), n, probe_base_tid, _p(filter_bitmap), _p(op), _p(ob),
                                    _p(blocks), cap)
            if k <= cap or capacity is not None:
                break
            cap = k
        k = min(k, cap)
        if with_blocks:
            return op[:k], ob[:k], blocks[:k]
        return op[:k], ob[:k]

    def probe_exists(self, keys, anti=False, filter_bitmap=None):
        out = np.zeros(max(words(keys.size), 1), dtype=np.uint64)
        _lib.qso_join_probe_exists(self._h, _p(keys), keys.size, _p(filter_bitmap), 1 if anti else 0, _p(out))
        return out


class CompositeJoinTable:
    """Join table over several fixed-width key components (SeparateChainingHashTable restated)."""

    def __init__(self, key_types, est_entries):
        self.key_types = list(key_types)
        arr = (C.c_int32 * len(self.key_types))(*self.key_types)
        self._h = C.c_void_p(_lib.qso_cjoin_table_create(len(self.key_types), arr, est_entries))
        assert self._h.value

    def close(self):
        if self._h is not None:
            _lib.qso_cjoin_table_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:  # interpreter shutdown
            pass

    @staticmethod
    def _cols(cols):
        keep = [np.ascontiguousarray(c) for c in cols]
        return keep, (C.c_void_p * len(keep))(*[c.ctypes.data for c in keep])

    def build(self, cols, block_id=0, base_tid=0, filter_bitmap=None):
        keep, ptrs = self._cols(cols)
        _lib.qso_cjoin_build(self._h, ptrs, keep[0].size, block_id, base_tid, _p(filter_bitmap))

    def probe(self, cols, probe_base_tid=0, filter_bitmap=None):
        keep, ptrs = self._cols(cols)
        n = keep[0].size
        k = _lib.qso_cjoin_probe(self._h, ptrs, n, probe_base_tid, _p(filter_bitmap), None, None, 0)
        op = np.empty(max(k, 1), dtype=np.int32)
        ob = np.empty(max(k, 1), dtype=np.int32)
        _lib.qso_cjoin_probe(self._h, ptrs, n, probe_base_tid, _p(filter_bitmap), _p(op), _p(ob), k)
        return op[:k], ob[:k]

    def hash_rows(self, cols):
        keep, ptrs = self._cols(cols)
        return np.array([_lib.qso_cjoin_hash_row(self._h, ptrs, i) for i in range(keep[0].size)], dtype=np.uint64)


def select_cmp_columns(lhs, rhs, op, filter_bitmap=None, qt=None):
    out = np.zeros(max(words(lhs.size), 1), dtype=np.uint64)
    _lib.qso_select_cmp_columns(_NP_TYPE[lhs.dtype] if qt is None else qt, _p(lhs), _p(rhs), lhs.size, op, _p(filter_bitmap), _p(out))
    return out


def tids_to_bitmap(tids, num_bits, base_tid=0):
    out = np.zeros(max(words(num_bits), 1), dtype=np.uint64)
    _lib.qso_tids_to_bitmap(_p(tids), tids.size, base_tid, num_bits, _p(out))
    return out


# ---- compressed attributes --------------------------------------------------------------
PRED_ALL, PRED_NONE, PRED_BASIC, PRED_RANGE = range(4)
_CODE_DTYPE = {1: np.uint8, 2: np.uint16, 4: np.uint32}


class CompressedColumn:
    """One attribute of a CompressedColumnStore block as CompressedBlockBuilder would store it:
    kind 0 uncompressed (codes = the values), 1 truncated, 2 dictionary-coded."""

    def __init__(self, values):
        values = np.ascontiguousarray(values)
        self.type = _NP_TYPE[values.dtype]
        self.dtype = values.dtype
        self.n = values.size
        self.info = CompressedInfo()
        raw = np.zeros(max(values.size, 1) * 8, dtype=np.uint8)
        dictionary = np.zeros(max(values.size, 1), dtype=values.dtype)
        _lib.qso_compress_column(self.type, _p(values), values.size, C.byref(self.info), _p(raw), _p(dictionary))
        self.kind, self.code_width = self.info.kind, self.info.code_width
        if self.kind == 0:
            self.codes = raw[: values.size * values.dtype.itemsize].view(values.dtype).copy()
            self.dictionary = None
        else:
            self.codes = raw[: values.size * self.code_width].view(_CODE_DTYPE[self.code_width]).copy()
            self.dictionary = dictionary[: self.info.num_codes].copy() if self.kind == 2 else None

    def transform(self, op, literal):
        """TransformPredicateOnCompressedAttribute -> CodePredicate (kinds 1 and 2 only)."""
        lit = np.array([literal], dtype=self.dtype)
        out = CodePredicate()
        _lib.qso_transform_predicate(C.byref(self.info), self.type, _p(self.dictionary), op, _p(lit), C.byref(out))
        return out

    def matches(self, op, literal, filter_bitmap=None):
        """getMatchesForPredicate on the code stripe (CompressedTupleStorageSubBlock.cpp:160-250)."""
        pred = self.transform(op, literal)
        if pred.result == PRED_NONE:
            return np.zeros(max(words(self.n), 1), dtype=np.uint64)
        if pred.result == PRED_ALL:
            return bitmap_from_bools(np.ones(self.n, dtype=bool)) if filter_bitmap is None else filter_bitmap.copy()
        return select_codes(self.codes, pred.comp, pred.first, pred.second, filter_bitmap)

    def decode(self):
        if self.kind == 0:
            return self.codes.copy()
        out = np.zeros(self.n, dtype=self.dtype)
        _lib.qso_decode_codes(self.code_width, _p(self.codes), self.n, _p(self.dictionary), self.dtype.itemsize, _p(out))
        return out


def select_codes(codes, op, first, second=0, filter_bitmap=None):
    out = np.zeros(max(words(codes.size), 1), dtype=np.uint64)
    _lib.qso_select_codes(codes.dtype.itemsize, _p(codes), codes.size, op, first, second, _p(filter_bitmap), _p(out))
    return out


def sort_permutation(key_cols, descending=None, types=None):
    """ORDER BY key_cols[0], key_cols[1], ...: stable, comparator semantics of SortConfiguration."""
    keep = [np.ascontiguousarray(c) for c in key_cols]
    n = keep[0].size
    ptrs = (C.c_void_p * len(keep))(*[c.ctypes.data for c in keep])
    types = (C.c_int32 * len(keep))(*(types if types is not None else [_NP_TYPE[c.dtype] for c in keep]))
    desc = (C.c_int32 * len(keep))(*[1 if (descending and descending[i]) else 0 for i in range(len(keep))])
    out = np.zeros(max(n, 1), dtype=np.int32)
    _lib.qso_sort_permutation(len(keep), ptrs, types, desc, n, _p(out))
    return out[:n]


def distinct_rows(cols, filter_bitmap=None, types=None):
    """Row numbers of the first occurrence of every distinct tuple, in tuple order (the distinctify table)."""
    keep = [np.ascontiguousarray(c) for c in cols]
    n = keep[0].size
    ptrs = (C.c_void_p * len(keep))(*[c.ctypes.data for c in keep])
    types = (C.c_int32 * len(keep))(*(types if types is not None else [_NP_TYPE[c.dtype] for c in keep]))
    out = np.zeros(max(n, 1), dtype=np.int32)
    count = _lib.qso_distinct_rows(len(keep), ptrs, types, n, _p(filter_bitmap), _p(out))
    return out[:count]


# ---- aggregation --------------------------------------------------------------------
class AggState:
    def __init__(self, config, handle=None):
        self.config = config
        self._h = C.c_void_p(_lib.qso_agg_state_create(C.byref(config))) if handle is None else handle

    def close(self):
        if self._h is not None:
            _lib.qso_agg_state_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:  # interpreter shutdown
            pass

    def update(self, cols, n=None, filter_bitmap=None):
        if n is None:
            n = cols[0].size
        _lib.qso_agg_update(self._h, _ptr_array(cols), n, _p(filter_bitmap))

    def update_nullable(self, cols, null_bitmaps, n=None, filter_bitmap=None):
        """null_bitmaps[c]: uint64 MSB-first null bitmap of column c, or None."""
        if n is None:
            n = cols[0].size
        keep = [np.ascontiguousarray(b, dtype=np.uint64) if b is not None else None for b in null_bitmaps]
        ptrs = (C.c_void_p * len(cols))(*[b.ctypes.data if b is not None else None for b in keep])
        _lib.qso_agg_update_nullable(self._h, _ptr_array(cols), ptrs, n, _p(filter_bitmap))

    def update_coded(self, cols, dictionaries, n=None, filter_bitmap=None):
        if n is None:
            n = len(cols[0])
        keep = [np.ascontiguousarray(d) if d is not None else None for d in dictionaries]
        dicts = (C.c_void_p * len(cols))(*[d.ctypes.data if d is not None else None for d in keep])
        _lib.qso_agg_update_coded(self._h, _ptr_array(cols), dicts, n, _p(filter_bitmap))

    def mark_existence(self, keys, filter_bitmap=None):
        _lib.qso_agg_mark_existence(self._h, qtype(keys), _p(keys), len(keys), _p(filter_bitmap))

    def merge(self, other):
        _lib.qso_agg_merge(self._h, other._h)

    def num_groups(self):
        return _lib.qso_agg_num_groups(self._h)

    def finalize(self, partition=0, num_partitions=1):
        cfg = self.config
        cap = max(self.num_groups(), 1)
        keys = []
        for k in range(cfg.num_keys):
            w = cfg.column_width[cfg.key_column[k]]
            keys.append(np.zeros(cap, dtype={1: np.uint8, 2: np.int16, 4: np.int32, 8: np.int64}[w]))
        vals, nulls = [], []
        for a in range(cfg.num_aggs):
            vals.append(np.zeros(cap, dtype=np.dtype(T.agg_output_dtype(cfg, a))))
            nulls.append(np.zeros(cap, dtype=np.uint8))
        rows = _lib.qso_agg_finalize(self._h, partition, num_partitions, _ptr_array(keys), _ptr_array(vals),
                                     _ptr_array(nulls), cap)
        return [k[:rows] for k in keys], [v[:rows] for v in vals], [z[:rows] for z in nulls]


# ---- LIP -------------------------------------------------------------------------------
class LipFilter:
    def __init__(self, kind, cardinality, min_value=0, is_anti=False):
        self._h = C.c_void_p(_lib.qso_lip_filter_create(kind, cardinality, min_value, 1 if is_anti else 0))

    def close(self):
        if self._h is not None:
            _lib.qso_lip_filter_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:  # interpreter shutdown
            pass

    def build(self, keys, filter_bitmap=None):
        _lib.qso_lip_build(self._h, qtype(keys), _p(keys), keys.size, _p(filter_bitmap))

    def probe(self, keys, in_bitmap=None):
        out = np.zeros(max(words(keys.size), 1), dtype=np.uint64)
        _lib.qso_lip_probe(self._h, qtype(keys), _p(keys), keys.size, _p(in_bitmap), _p(out))
        return out


# ---- partition ----------------------------------------------------------------------------
def partition_offsets(keys, num_partitions):
    out = np.zeros(num_partitions + 1, dtype=np.int64)
    _lib.qso_partition_offsets(qtype(keys), _p(keys), keys.size, num_partitions, _p(out))
    return out


def partition_scatter(keys, num_partitions, col):
    out = np.empty_like(col)
    _lib.qso_partition_scatter_col(qtype(keys), _p(keys), keys.size, num_partitions, col.itemsize, _p(col), _p(out))
    return out


# ---- CPU baseline drivers ---------------------------------------------------------------------
def bench_join(build_keys, probe_keys, block_rows, num_threads):
    r = JoinBenchResult()
    _lib.qso_bench_join(qtype(build_keys), _p(build_keys), build_keys.size, _p(probe_keys), probe_keys.size,
                        block_rows, num_threads, C.byref(r))
    return dict(build_seconds=r.build_seconds, probe_seconds=r.probe_seconds, matches=r.matches, checksum=r.checksum)


def bench_agg(config, cols, n, block_rows, num_threads):
    h = C.c_void_p()
    secs = _lib.qso_bench_agg(C.byref(config), _ptr_array(cols), n, block_rows, num_threads, C.byref(h))
    return secs, AggState(config, handle=h)


def bench_select(col, op, literal, block_rows, num_threads):
    out = np.empty_like(col)
    rows = C.c_int64()
    lit = _C_SCALAR[qtype(col)](literal)
    secs = _lib.qso_bench_select(qtype(col), _p(col), col.size, op, C.byref(lit), block_rows, num_threads, _p(out),
                                 C.byref(rows))
    return secs, out[:rows.value]


def bench_agg_coded(config, cols, dictionaries, n, block_rows, num_threads):
    """cols: code stripes for the columns with column_code_width != 0, plain stripes otherwise; dictionaries: one per column or None."""
    h = C.c_void_p()
    secs = _lib.qso_bench_agg_coded(C.byref(config), _ptr_array(cols), _ptr_array(dictionaries), n, block_rows, num_threads, C.byref(h))
    return secs, AggState(config, handle=h)


def bench_partitioned_join(o_key, o_payload, l_key, l_payload, num_partitions, block_rows, num_threads):
    """BASELINE config 4 in one process (repartition both sides, per-partition build / probe / materialise)."""
    assert o_key.dtype == np.int32 and l_key.dtype == np.int32 and o_payload.dtype == np.int64 and l_payload.dtype == np.int64
    r = PartitionedJoinResult()
    _lib.qso_bench_partitioned_join(_p(o_key), _p(o_payload), o_key.size, _p(l_key), _p(l_payload), l_key.size,
                                    num_partitions, block_rows, num_threads, C.byref(r))
    return {f: getattr(r, f) for f, _ in PartitionedJoinResult._fields_}


def bench_q3(inputs, segment, date_cut, block_rows, num_threads):
    """BASELINE config 5 in one process.  inputs: dict of numpy columns (int32 keys / dates / segment codes, float64 price and
    discount) plus customers_total, orders_total."""
    a = Q3Inputs()
    keep = []
    for name, dt in (("c_custkey", np.int32), ("c_mktsegment", np.int32), ("o_orderkey", np.int32), ("o_custkey", np.int32),
                     ("o_orderdate", np.int32), ("l_orderkey", np.int32), ("l_shipdate", np.int32),
                     ("l_extendedprice", np.float64), ("l_discount", np.float64)):
        arr = np.ascontiguousarray(inputs[name], dtype=dt)
        keep.append(arr)
        setattr(a, name, arr.ctypes.data)
    a.n_customer, a.n_orders, a.n_lineitem = inputs["c_custkey"].size, inputs["o_orderkey"].size, inputs["l_orderkey"].size
    a.customers_total, a.orders_total = int(inputs["customers_total"]), int(inputs["orders_total"])
    a.segment, a.date_cut = int(segment), int(date_cut)
    r = Q3Result()
    _lib.qso_bench_q3(C.byref(a), block_rows, num_threads, C.byref(r))
    out = {f: getattr(r, f) for f, _ in Q3Result._fields_}
    out["top_revenue"], out["top_orderkey"] = list(r.top_revenue), list(r.top_orderkey)
    return out
