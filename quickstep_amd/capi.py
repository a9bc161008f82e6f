"""ctypes binding of the C ABI in include/qsx.h (quickstep_amd/lib/libqsx.so).

This is the *product* path as seen from Python: tests marked ``gpu``, bench.py
and ``__graft_entry__.smoke()`` all go through here, i.e. through the same
``extern "C"`` entry points a Quickstep GPU work order would call.  There is no
CPU implementation behind these functions; if the shared library is missing
the import of this module fails, and on a machine without a gfx950 device
every compute call raises ``QsxError(QSX_ERR_NO_DEVICE)``.

Device memory, streams and collectives are torch's (plumbing): functions take
``torch`` CUDA tensors and pass ``tensor.data_ptr()`` through.
"""
import ctypes as C
import os

import torch  # noqa: F401  (imported first so that libqsx binds to the HIP runtime torch already loaded)

from . import types as T

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("QSX_LIB_PATH") or os.path.join(_HERE, "lib", "libqsx.so")   # override: A/B runs of two builds


class QsxError(RuntimeError):
    def __init__(self, status, where):
        self.status = status
        msg = _lib.qsx_status_string(status).decode()
        detail = _lib.qsx_last_error().decode()
        super().__init__(f"{where}: {msg} ({status})" + (f" — {detail}" if detail and status in (T.ERR_HIP, T.ERR_COMM) else ""))


def _load():
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(or make -C quickstep_amd/csrc). There is no CPU fallback for the HIP execution kernel.")
    return C.CDLL(LIB_PATH)


_lib = _load()

_vp, _i64, _i32, _int, _sz = C.c_void_p, C.c_int64, C.c_int32, C.c_int, C.c_size_t
_pp = C.POINTER(C.c_void_p)

_SIGNATURES = {
    "qsx_status_string": (C.c_char_p, [_int]),
    "qsx_abi_version": (_int, []),
    "qsx_abi_sizeof_agg_config": (_sz, []),
    "qsx_device_count": (_int, []),
    "qsx_current_device": (_int, [C.POINTER(C.c_int)]),
    "qsx_set_current_device": (_int, [_int]),
    "qsx_last_error": (C.c_char_p, []),
    "qsx_device_alloc": (_int, [_sz, _pp]),
    "qsx_device_free": (_int, [_vp]),
    "qsx_copy_to_device": (_int, [_vp, _vp, _sz, _vp]),
    "qsx_copy_to_host": (_int, [_vp, _vp, _sz, _vp]),
    "qsx_copy_on_device": (_int, [_vp, _vp, _sz, _vp]),
    "qsx_memset_device": (_int, [_vp, _int, _sz, _vp]),
    "qsx_stream_synchronize": (_int, [_vp]),
    "qsx_stream_create": (_int, [_pp]),
    "qsx_stream_destroy": (_int, [_vp]),
    "qsx_trim_scratch": (_int, [C.POINTER(_sz)]),
    "qsx_set_out_of_memory_hook": (_int, [_vp, _vp]),
    "qsx_select_cmp": (_int, [_int, _vp, _i64, _int, _vp, _vp, _vp, _vp, _vp]),
    "qsx_select_cmp_sorted": (_int, [_int, _vp, _i64, _int, _vp, _vp, _vp, _vp, _vp]),
    "qsx_select_cmp_char": (_int, [_vp, _int, _i64, _int, C.c_char_p, _int, _vp, _vp, _vp, _vp]),
    "qsx_select_cmp_columns": (_int, [_int, _vp, _vp, _i64, _int, _vp, _vp, _vp, _vp]),
    "qsx_select_codes": (_int, [_int, _vp, _i64, _int, C.c_uint32, C.c_uint32, _vp, _vp, _vp, _vp]),
    "qsx_select_codes_sorted": (_int, [_int, _vp, _i64, _int, C.c_uint32, C.c_uint32, _vp, _vp, _vp, _vp]),
    "qsx_decode_codes": (_int, [_int, _vp, _i64, _vp, _int, _vp, _vp]),
    "qsx_bitmap_combine": (_int, [_int, _vp, _vp, _i64, _vp, _vp]),
    "qsx_bitmap_count": (_int, [_vp, _i64, _vp, _vp]),
    "qsx_compact_workspace_bytes": (_sz, [_i64]),
    "qsx_copy_segments": (_int, [_i64, _pp, _pp, C.POINTER(_i64), _vp]),
    "qsx_compact_blocks_workspace_bytes": (_sz, [_i64, C.POINTER(_i64)]),
    "qsx_compact_gather_blocks": (_int, [_int, C.POINTER(_i32), _i64, C.POINTER(_i64), _pp, _pp, C.POINTER(_i32), _pp, _vp, _vp, _vp,
                                         _sz, _vp]),
    "qsx_compact_gather": (_int, [_int, _pp, C.POINTER(_i32), _vp, _i64, _pp, _vp, _vp, _sz, _vp]),
    "qsx_bitmap_to_tids": (_int, [_vp, _i64, _i32, _vp, _vp, _vp, _sz, _vp]),
    "qsx_tids_to_bitmap": (_int, [_vp, _i64, _i32, _i64, _vp, _vp]),
    "qsx_gather": (_int, [_int, _vp, _vp, _i64, _vp, _vp]),
    "qsx_gather_segmented": (_int, [_int, _int, _pp, C.POINTER(_i64), _vp, _i64, _vp, _vp]),
    "qsx_bitmap_gather_segmented": (_int, [_int, _pp, C.POINTER(_i64), _vp, _i64, _vp, _vp]),
    "qsx_sort_workspace_bytes": (_sz, [_i64]),
    "qsx_sort_permutation": (_int, [_int, _pp, C.POINTER(_i32), C.POINTER(_i32), _i64, _vp, _vp, _sz, _vp]),
    "qsx_distinct_rows": (_int, [_int, _pp, C.POINTER(_i32), _i64, _vp, _vp, _vp, _vp, _sz, _vp]),
    "qsx_sort_top_k": (_int, [_int, _pp, C.POINTER(_i32), C.POINTER(_i32), _i64, _i64, _vp, _vp, _sz, _vp]),
    "qsx_join_table_create": (_int, [_int, _i64, _pp]),
    "qsx_join_table_create_dense": (_int, [_int, _i64, _i64, _i64, _i64, _pp]),
    "qsx_join_table_destroy": (_int, [_vp]),
    "qsx_join_table_release": (_int, [_vp]),
    "qsx_join_key_pack": (_int, [_int, _pp, C.POINTER(_i32), _i64, _vp, C.POINTER(_int), _vp]),
    "qsx_join_key_pack_char": (_int, [_vp, _int, _i64, _vp, _vp]),
    "qsx_join_table_clear": (_int, [_vp, _vp]),
    "qsx_join_table_size": (_int, [_vp, C.POINTER(_i64), _vp]),
    "qsx_join_build": (_int, [_vp, _vp, _i64, _i32, _vp, _vp]),
    "qsx_join_probe": (_int, [_vp, _vp, _i64, _i32, _vp, _vp, _vp, _i64, _vp, _vp]),
    "qsx_join_probe_lip": (_int, [_vp, _vp, _i64, _i32, _vp, _int, _pp, _vp, _vp, _i64, _vp, _vp]),
    "qsx_join_key_pack_blocks": (_int, [_int, C.POINTER(_i32), _i64, C.POINTER(_i64), _pp, _vp, C.POINTER(_int), _vp]),
    "qsx_join_key_pack_blocks_coded": (_int, [_int, C.POINTER(_i32), _i64, C.POINTER(_i64), _pp, C.POINTER(_i32), _pp, _vp, C.POINTER(_int), _vp]),
    "qsx_join_build_blocks": (_int, [_vp, _i64, C.POINTER(_i64), _pp, C.POINTER(_i32), _pp, _vp]),
    "qsx_join_probe_blocks": (_int, [_vp, _i64, C.POINTER(_i64), _pp, C.POINTER(_i32), _pp, _vp, _vp, _i64, _vp, _vp]),
    "qsx_join_probe_count_blocks": (_int, [_vp, _i64, C.POINTER(_i64), _pp, _pp, _vp, _vp]),
    "qsx_join_probe_project_blocks": (_int, [_vp, _i64, C.POINTER(_i64), _pp, _pp, C.POINTER(T.JoinProjection), _i64, _vp, _vp]),
    "qsx_join_probe_exists_blocks": (_int, [_vp, _i64, C.POINTER(_i64), _pp, _pp, _int, _pp, _vp, _vp]),
    "qsx_join_probe_exists_lip": (_int, [_vp, _vp, _i64, _vp, _int, _pp, _vp, _vp, _vp]),
    "qsx_join_build_blocks_coded": (_int, [_vp, _i64, C.POINTER(_i64), _pp, _vp, C.POINTER(_i32), _pp, _vp]),
    "qsx_join_probe_blocks_coded": (_int, [_vp, _i64, C.POINTER(_i64), _pp, _vp, C.POINTER(_i32), _pp, _vp, _vp, _i64, _vp, _vp]),
    "qsx_join_probe_count_blocks_coded": (_int, [_vp, _i64, C.POINTER(_i64), _pp, _vp, _pp, _vp, _vp]),
    "qsx_join_probe_project_blocks_coded": (_int, [_vp, _i64, C.POINTER(_i64), _pp, _vp, _pp, C.POINTER(T.JoinProjection), _i64, _vp, _vp]),
    "qsx_join_probe_exists_blocks_coded": (_int, [_vp, _i64, C.POINTER(_i64), _pp, _vp, _pp, _int, _pp, _vp, _vp]),
    "qsx_join_probe_count": (_int, [_vp, _vp, _i64, _vp, _vp, _vp]),
    "qsx_join_probe_exists": (_int, [_vp, _vp, _i64, _vp, _int, _vp, _vp, _vp]),
    "qsx_eval_expression": (_int, [_int, _pp, C.POINTER(_i32), _int, C.POINTER(T.ExprInstr), C.POINTER(C.c_double), T.Operand, _i64, _vp, _vp]),
    "qsx_eval_expression_long": (_int, [_int, _pp, C.POINTER(_i32), _int, C.POINTER(T.ExprInstr), C.POINTER(_i64), T.Operand, _i64, _int, _vp, _vp]),
    "qsx_agg_state_create": (_int, [C.POINTER(T.AggConfig), _pp]),
    "qsx_agg_state_destroy": (_int, [_vp]),
    "qsx_select_cmp_sorted_blocks": (_int, [_int, _i64, C.POINTER(_i64), _pp, _int, _vp, _pp, _pp, _vp, _vp]),
    "qsx_select_cmp_char_blocks": (_int, [_int, _i64, C.POINTER(_i64), _pp, _int, _vp, _int, _pp, _pp, _vp, _vp]),
    "qsx_select_codes_blocks": (_int, [_int, _i64, C.POINTER(_i64), _pp, C.POINTER(_i32), C.POINTER(C.c_uint32), C.POINTER(C.c_uint32), _pp, _pp,
                                       _vp, _vp]),
    "qsx_select_codes_sorted_blocks": (_int, [_int, _i64, C.POINTER(_i64), _pp, C.POINTER(_i32), C.POINTER(C.c_uint32), C.POINTER(C.c_uint32), _pp, _pp,
                                              _vp, _vp]),
    "qsx_select_cmp_blocks": (_int, [_int, _i64, C.POINTER(_i64), _pp, _int, _vp, _pp, _pp, _vp, _vp]),
    "qsx_agg_state_clear": (_int, [_vp, _vp]),
    "qsx_agg_update": (_int, [_vp, _pp, _i64, _vp, _vp]),
    "qsx_agg_update_blocks": (_int, [_vp, _int, C.POINTER(_i64), _pp, _pp, _vp]),
    "qsx_agg_update_coded_blocks": (_int, [_vp, _int, C.POINTER(_i64), _pp, _pp, _pp, _vp]),
    "qsx_agg_update_coded_blocks_sized": (_int, [_vp, _int, C.POINTER(_i64), _pp, _pp, C.POINTER(C.c_int32), _pp, _vp]),
    "qsx_agg_mark_existence": (_int, [_vp, _int, _vp, _i64, _vp, _vp]),
    "qsx_agg_update_coded": (_int, [_vp, _pp, _pp, _i64, _vp, _vp]),
    "qsx_agg_update_coded_sized": (_int, [_vp, _pp, _pp, C.POINTER(C.c_int32), _i64, _vp, _vp]),
    "qsx_agg_update_nullable": (_int, [_vp, _pp, _pp, _i64, _vp, _vp]),
    "qsx_agg_merge": (_int, [_vp, _vp, _vp]),
    "qsx_agg_state_export_bytes": (_int, [_vp, C.POINTER(_sz), _vp]),
    "qsx_agg_state_export": (_int, [_vp, _vp, _sz, _vp]),
    "qsx_agg_state_image_layout": (_int, [_vp, C.POINTER(_int), C.POINTER(_i64), C.POINTER(_i64), C.POINTER(_int), C.POINTER(_i32), _int]),
    "qsx_agg_state_import_merge": (_int, [_vp, _vp, _sz, _vp]),
    "qsx_agg_num_groups": (_int, [_vp, C.POINTER(_i64), _vp]),
    "qsx_agg_finalize": (_int, [_vp, _int, _int, _pp, _pp, _pp, _i64, _vp, _vp]),
    "qsx_lip_filter_create": (_int, [_int, _i64, _i64, _int, _pp]),
    "qsx_lip_filter_destroy": (_int, [_vp]),
    "qsx_lip_build": (_int, [_vp, _int, _vp, _i64, _vp, _vp]),
    "qsx_lip_build_from_join_table": (_int, [_vp, _vp, _i64, _vp]),
    "qsx_lip_probe": (_int, [_vp, _int, _vp, _i64, _vp, _vp, _vp, _vp]),
    "qsx_lip_build_blocks": (_int, [_vp, _int, _i64, C.POINTER(_i64), _pp, _pp, _vp]),
    "qsx_lip_probe_blocks": (_int, [_vp, _int, _i64, C.POINTER(_i64), _pp, _pp, _pp, _vp, _vp]),
    "qsx_lip_build_blocks_coded": (_int, [_vp, _int, _i64, C.POINTER(_i64), _pp, _vp, _pp, _vp]),
    "qsx_lip_probe_blocks_coded": (_int, [_vp, _int, _i64, C.POINTER(_i64), _pp, _vp, _pp, _pp, _vp, _vp]),
    "qsx_lip_filter_words": (_int, [_vp, _pp, C.POINTER(_i64)]),
    "qsx_comm_unique_id": (_int, [_vp]),
    "qsx_comm_create": (_int, [_int, _int, _vp, _pp]),
    "qsx_comm_destroy": (_int, [_vp]),
    "qsx_comm_rank": (_int, [_vp, C.POINTER(_int), C.POINTER(_int)]),
    "qsx_comm_agree": (_int, [_vp, _int, _vp]),
    "qsx_comm_synchronize": (_int, [_vp, _vp]),
    "qsx_comm_abort": (_int, [_vp]),
    "qsx_exchange_counts": (_int, [_vp, _vp, _vp, _vp]),
    "qsx_alltoallv": (_int, [_vp, _int, _vp, C.POINTER(_i64), _vp, C.POINTER(_i64), _vp]),
    "qsx_allgather": (_int, [_vp, _vp, _sz, _vp, _vp]),
    "qsx_bitmap_allreduce_or": (_int, [_vp, _vp, _i64, _vp]),
    "qsx_agg_reduce_scatter": (_int, [_vp, _vp, _vp]),
    "qsx_agg_allgather_merge": (_int, [_vp, _vp, _vp]),
    "qsx_agg_dense_partition_range": (_int, [_i64, _int, _int, C.POINTER(_i64), C.POINTER(_i64), C.POINTER(_i64), C.POINTER(_i64),
                                             C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]),
    "qsx_partition_workspace_bytes": (_sz, [_i64, _int]),
    "qsx_partition_scatter": (_int, [_int, _vp, _i64, _int, _int, _pp, C.POINTER(_i32), _pp, _vp, _vp, _sz, _vp]),
    "qsx_partition_blocks_workspace_bytes": (_sz, [_i64, _i64, _int]),
    "qsx_partition_scatter_blocks": (_int, [_int, _i64, C.POINTER(_i64), _pp, _int, _int, _pp, C.POINTER(_i32), _pp, _vp, _vp, _sz, _vp]),
}

# every symbol include/qsx.h declares must resolve (tests/test_abi.py checks the header against this table)
for _name, (_res, _args) in _SIGNATURES.items():
    _fn = getattr(_lib, _name)
    _fn.restype = _res
    _fn.argtypes = _args

if _lib.qsx_abi_sizeof_agg_config() != C.sizeof(T.AggConfig):
    raise ImportError("quickstep_amd.types.AggConfig does not match qsx_agg_config_t of libqsx.so")

lib = _lib
EXPORTED = tuple(_SIGNATURES)


def _check(status, where):
    if status != T.OK:
        raise QsxError(status, where)


def trim_scratch():
    """Release what the calling thread keeps inside libqsx.so between calls (qsx_trim_scratch); returns the bytes."""
    v = C.c_size_t()
    _check(_lib.qsx_trim_scratch(C.byref(v)), "qsx_trim_scratch")
    return v.value


def device_count():
    return _lib.qsx_device_count()


def current_device():
    """The HIP device current in the calling thread (qsx_current_device): the one every call of this thread works on."""
    v = C.c_int(-1)
    _check(_lib.qsx_current_device(C.byref(v)), "qsx_current_device")
    return v.value


def set_current_device(device):
    _check(_lib.qsx_set_current_device(int(device)), "qsx_set_current_device")


def _ptr(t):
    return None if t is None else C.c_void_p(t.data_ptr())


def _stream(stream=None):
    s = torch.cuda.current_stream() if stream is None else stream
    return C.c_void_p(s.cuda_stream)


_TORCH_TYPE = {torch.int32: T.INT, torch.int64: T.LONG, torch.float32: T.FLOAT, torch.float64: T.DOUBLE,
               torch.uint8: T.CHAR}   # CHAR(1)
_C_SCALAR = {T.INT: C.c_int32, T.LONG: C.c_int64, T.FLOAT: C.c_float, T.DOUBLE: C.c_double}


def qsx_type_of(t):
    return _TORCH_TYPE[t.dtype]


def bitmap_words(n):
    return (n + 63) // 64


def new_bitmap(n, device):
    return torch.empty(max(bitmap_words(n), 1), dtype=torch.int64, device=device)


def _ptr_array(tensors):
    arr = (C.c_void_p * max(len(tensors), 1))()
    for i, t in enumerate(tensors):
        arr[i] = None if t is None else t.data_ptr()
    return arr


# --------------------------------------------------------------------------- select
def _literal(qt, literal):
    return C.c_int64(literal) if qt == T.DATE else _C_SCALAR[qt](literal)


def select_cmp(col, op, literal, filter_bitmap=None, out_bitmap=None, out_count=None, stream=None, qtype=None):
    """K1: returns (bitmap int64[ceil(n/64)] MSB-first, count int64[1]) device tensors.  qtype=T.DATE: col is an int64
    tensor of raw DateLit bytes, literal = T.date_raw(...)."""
    n = col.numel()
    qt = qsx_type_of(col) if qtype is None else qtype
    lit = _literal(qt, literal)
    if out_bitmap is None:
        out_bitmap = new_bitmap(n, col.device)
    if out_count is None:
        out_count = torch.zeros(1, dtype=torch.int64, device=col.device)
    _check(_lib.qsx_select_cmp(qt, _ptr(col), n, op, C.byref(lit), _ptr(filter_bitmap), _ptr(out_bitmap),
                               _ptr(out_count), _stream(stream)), "qsx_select_cmp")
    return out_bitmap, out_count


def select_cmp_blocks(cols, op, literal, filters=None, stream=None, qtype=None, out_bitmaps=None):
    """K1 over a run of blocks in one launch: cols = one stripe per block; returns (list of per-block bitmaps, counts int64[nb])."""
    nb = len(cols)
    qt = qsx_type_of(cols[0]) if qtype is None else qtype
    lit = _literal(qt, literal)
    dev = cols[0].device
    outs = out_bitmaps if out_bitmaps is not None else [new_bitmap(c.numel(), dev) for c in cols]
    counts = torch.zeros(max(nb, 1), dtype=torch.int64, device=dev)
    rows = (C.c_int64 * max(nb, 1))(*[c.numel() for c in cols])
    cptr = (C.c_void_p * max(nb, 1))(*[c.data_ptr() if c.numel() else None for c in cols])
    optr = (C.c_void_p * max(nb, 1))(*[o.data_ptr() if o.numel() else None for o in outs])
    fptr = None
    if filters is not None:
        fptr = (C.c_void_p * max(nb, 1))(*[f.data_ptr() if f is not None and f.numel() else None for f in filters])
    _check(_lib.qsx_select_cmp_blocks(qt, nb, rows, cptr, op, C.byref(lit), fptr, optr, _ptr(counts), _stream(stream)),
           "qsx_select_cmp_blocks")
    return outs, counts[:nb]


def _run_outputs(cols, filters):
    nb = len(cols)
    dev = cols[0].device
    outs = [new_bitmap(c.numel(), dev) for c in cols]
    counts = torch.zeros(max(nb, 1), dtype=torch.int64, device=dev)
    rows = (C.c_int64 * max(nb, 1))(*[c.numel() for c in cols])
    cptr = (C.c_void_p * max(nb, 1))(*[c.data_ptr() if c.numel() else None for c in cols])
    optr = (C.c_void_p * max(nb, 1))(*[o.data_ptr() if o.numel() else None for o in outs])
    fptr = None
    if filters is not None:
        fptr = (C.c_void_p * max(nb, 1))(*[f.data_ptr() if f is not None and f.numel() else None for f in filters])
    return nb, outs, counts, rows, cptr, optr, fptr


def select_cmp_sorted_blocks(cols, op, literal, filters=None, stream=None, qtype=None):
    """K1 on the sort column of a run of sorted blocks (every block sorted on its own): (per-block bitmaps, counts int64[nb])."""
    qt = qsx_type_of(cols[0]) if qtype is None else qtype
    lit = _literal(qt, literal)
    nb, outs, counts, rows, cptr, optr, fptr = _run_outputs(cols, filters)
    _check(_lib.qsx_select_cmp_sorted_blocks(qt, nb, rows, cptr, op, C.byref(lit), fptr, optr, _ptr(counts), _stream(stream)),
           "qsx_select_cmp_sorted_blocks")
    return outs, counts[:nb]


def select_cmp_char_blocks(cols, op, literal, filters=None, stream=None):
    """K1 on CHAR(width) stripes of a run of blocks: cols = uint8 tensors of shape (n_b, width), literal a bytes object."""
    width = cols[0].shape[1]
    flat = [c.reshape(-1) for c in cols]
    nb = len(cols)
    dev = cols[0].device
    outs = [new_bitmap(c.shape[0], dev) for c in cols]
    counts = torch.zeros(max(nb, 1), dtype=torch.int64, device=dev)
    rows = (C.c_int64 * max(nb, 1))(*[c.shape[0] for c in cols])
    cptr = (C.c_void_p * max(nb, 1))(*[f.data_ptr() if f.numel() else None for f in flat])
    optr = (C.c_void_p * max(nb, 1))(*[o.data_ptr() if o.numel() else None for o in outs])
    fptr = None
    if filters is not None:
        fptr = (C.c_void_p * max(nb, 1))(*[f.data_ptr() if f is not None and f.numel() else None for f in filters])
    _check(_lib.qsx_select_cmp_char_blocks(width, nb, rows, cptr, op, C.c_char_p(literal), len(literal), fptr, optr, _ptr(counts),
                                           _stream(stream)), "qsx_select_cmp_char_blocks")
    return outs, counts[:nb]


def select_codes_blocks(code_blocks, ops, firsts, seconds, filters=None, stream=None):
    """K1 on the code stripes of a run of compressed blocks: per-block code comparison (QSX_CODE_* op, first, second)."""
    nb, outs, counts, rows, cptr, optr, fptr = _run_outputs(code_blocks, filters)
    _check(_lib.qsx_select_codes_blocks(code_blocks[0].element_size(), nb, rows, cptr, (C.c_int32 * max(nb, 1))(*ops),
                                        (C.c_uint32 * max(nb, 1))(*firsts), (C.c_uint32 * max(nb, 1))(*seconds), fptr, optr,
                                        _ptr(counts), _stream(stream)), "qsx_select_codes_blocks")
    return outs, counts[:nb]


def select_codes_sorted_blocks(code_blocks, ops, firsts, seconds, filters=None, stream=None):
    """K1 on the compressed sort column of a run of blocks: per-block code comparison (QSX_CODE_* op, first, second)."""
    nb, outs, counts, rows, cptr, optr, fptr = _run_outputs(code_blocks, filters)
    _check(_lib.qsx_select_codes_sorted_blocks(code_blocks[0].element_size(), nb, rows, cptr, (C.c_int32 * max(nb, 1))(*ops),
                                               (C.c_uint32 * max(nb, 1))(*firsts), (C.c_uint32 * max(nb, 1))(*seconds), fptr, optr,
                                               _ptr(counts), _stream(stream)), "qsx_select_codes_sorted_blocks")
    return outs, counts[:nb]


def select_cmp_char(col, op, literal, filter_bitmap=None, stream=None):
    """K1 on a CHAR(width) stripe: col is a uint8 tensor of shape (n, width), literal a bytes object."""
    n, width = col.shape
    assert col.dtype == torch.uint8 and col.is_contiguous()
    out_bitmap = new_bitmap(n, col.device)
    out_count = torch.zeros(1, dtype=torch.int64, device=col.device)
    _check(_lib.qsx_select_cmp_char(_ptr(col), width, n, op, C.c_char_p(literal), len(literal), _ptr(filter_bitmap), _ptr(out_bitmap),
                                    _ptr(out_count), _stream(stream)), "qsx_select_cmp_char")
    return out_bitmap, out_count


def select_cmp_sorted(col, op, literal, filter_bitmap=None, stream=None, qtype=None):
    """K1 on the sort column of a sorted column store (binary search): same result surface as select_cmp."""
    n = col.numel()
    qt = qsx_type_of(col) if qtype is None else qtype
    lit = _literal(qt, literal)
    out_bitmap = new_bitmap(n, col.device)
    out_count = torch.zeros(1, dtype=torch.int64, device=col.device)
    _check(_lib.qsx_select_cmp_sorted(qt, _ptr(col), n, op, C.byref(lit), _ptr(filter_bitmap), _ptr(out_bitmap),
                                      _ptr(out_count), _stream(stream)), "qsx_select_cmp_sorted")
    return out_bitmap, out_count


def select_cmp_columns(lhs, rhs, op, filter_bitmap=None, stream=None, qtype=None):
    """K1, attribute OP attribute: returns (bitmap, count) like select_cmp."""
    n = lhs.numel()
    assert rhs.numel() == n and rhs.dtype == lhs.dtype
    out_bitmap = new_bitmap(n, lhs.device)
    out_count = torch.zeros(1, dtype=torch.int64, device=lhs.device)
    _check(_lib.qsx_select_cmp_columns(qsx_type_of(lhs) if qtype is None else qtype, _ptr(lhs), _ptr(rhs), n, op, _ptr(filter_bitmap),
                                       _ptr(out_bitmap), _ptr(out_count), _stream(stream)), "qsx_select_cmp_columns")
    return out_bitmap, out_count


def select_codes(codes, op, first, second=0, filter_bitmap=None, stream=None):
    """K1 on a compressed attribute's code stripe (uint8 / int16 / int32 tensors holding unsigned codes)."""
    n = codes.numel()
    out_bitmap = new_bitmap(n, codes.device)
    out_count = torch.zeros(1, dtype=torch.int64, device=codes.device)
    _check(_lib.qsx_select_codes(codes.element_size(), _ptr(codes), n, op, first, second, _ptr(filter_bitmap),
                                 _ptr(out_bitmap), _ptr(out_count), _stream(stream)), "qsx_select_codes")
    return out_bitmap, out_count


def eval_expression(cols, instrs, consts, result, stream=None):
    """K11 standalone: the value of `result` (T.col / T.const / T.temp) after the program, per row, as float64."""
    n = cols[0].numel()
    out = torch.empty(n, dtype=torch.float64, device=cols[0].device)
    ptrs = (C.c_void_p * max(len(cols), 1))(*[c.data_ptr() for c in cols])
    types = (C.c_int32 * max(len(cols), 1))(*[qsx_type_of(c) for c in cols])
    prog = (T.ExprInstr * max(len(instrs), 1))(*[T.ExprInstr(op, dst, a, b) for op, dst, a, b in instrs])
    cs = (C.c_double * T.MAX_CONSTS)(*list(consts))
    _check(_lib.qsx_eval_expression(len(cols), ptrs, types, len(instrs), prog, cs, result, n, _ptr(out), _stream(stream)),
           "qsx_eval_expression")
    return out


def eval_expression_long(cols, instrs, consts, result, out_dtype=torch.int64, stream=None):
    """The expression program over INT / LONG columns in integer arithmetic (qsx_eval_expression_long)."""
    n = cols[0].numel()
    out = torch.empty(n, dtype=out_dtype, device=cols[0].device)
    ptrs = (C.c_void_p * max(len(cols), 1))(*[c.data_ptr() for c in cols])
    types = (C.c_int32 * max(len(cols), 1))(*[qsx_type_of(c) for c in cols])
    prog = (T.ExprInstr * max(len(instrs), 1))(*[T.ExprInstr(op, dst, a, b) for op, dst, a, b in instrs])
    cs = (C.c_int64 * T.MAX_CONSTS)(*[int(c) for c in consts])
    _check(_lib.qsx_eval_expression_long(len(cols), ptrs, types, len(instrs), prog, cs, result, n, out.element_size(), _ptr(out),
                                         _stream(stream)), "qsx_eval_expression_long")
    return out


def select_codes_sorted(codes, op, first, second=0, filter_bitmap=None, stream=None):
    """K1 on the code stripe of a compressed SORT column (ascending codes): binary search instead of a scan."""
    n = codes.numel()
    out_bitmap = new_bitmap(n, codes.device)
    out_count = torch.zeros(1, dtype=torch.int64, device=codes.device)
    _check(_lib.qsx_select_codes_sorted(codes.element_size(), _ptr(codes), n, op, first, second, _ptr(filter_bitmap),
                                        _ptr(out_bitmap), _ptr(out_count), _stream(stream)), "qsx_select_codes_sorted")
    return out_bitmap, out_count


def copy_segments(srcs, dsts, stream=None):
    """qsx_copy_segments: srcs[i] -> dsts[i] (flat tensors of equal byte size), one launch."""
    n = len(srcs)
    sp = (C.c_void_p * max(n, 1))(*[s.data_ptr() if s.numel() else None for s in srcs])
    dp = (C.c_void_p * max(n, 1))(*[d.data_ptr() if d.numel() else None for d in dsts])
    nbytes = (C.c_int64 * max(n, 1))(*[s.numel() * s.element_size() for s in srcs])
    _check(_lib.qsx_copy_segments(n, sp, dp, nbytes, _stream(stream)), "qsx_copy_segments")


def decode_codes(codes, dictionary, value_dtype, stream=None, out=None):
    """codes -> values: dictionary lookup, or zero-extension when dictionary is None (truncated attribute)."""
    if out is None:
        out = torch.empty(codes.numel(), dtype=value_dtype, device=codes.device)
    _check(_lib.qsx_decode_codes(codes.element_size(), _ptr(codes), codes.numel(), _ptr(dictionary), out.element_size(),
                                 _ptr(out), _stream(stream)), "qsx_decode_codes")
    return out


def bitmap_combine(op, a, b, n, out=None, stream=None):
    if out is None:
        out = torch.empty_like(a)
    _check(_lib.qsx_bitmap_combine(op, _ptr(a), _ptr(b), n, _ptr(out), _stream(stream)), "qsx_bitmap_combine")
    return out


def bitmap_count(bitmap, n, stream=None):
    out = torch.zeros(1, dtype=torch.int64, device=bitmap.device)
    _check(_lib.qsx_bitmap_count(_ptr(bitmap), n, _ptr(out), _stream(stream)), "qsx_bitmap_count")
    return out


def compact_gather(cols, bitmap, n, out_cols=None, stream=None):
    """K2: returns (list of output columns sized n, count int64[1])."""
    device = bitmap.device
    if out_cols is None:
        out_cols = [torch.empty_like(c) for c in cols]
    widths = (C.c_int32 * max(len(cols), 1))(*[c.element_size() for c in cols])
    ws_bytes = _lib.qsx_compact_workspace_bytes(n)
    ws = torch.empty(max(ws_bytes, 8), dtype=torch.uint8, device=device)
    count = torch.zeros(1, dtype=torch.int64, device=device)
    _check(_lib.qsx_compact_gather(len(cols), _ptr_array(cols), widths, _ptr(bitmap), n, _ptr_array(out_cols),
                                   _ptr(count), _ptr(ws), ws_bytes, _stream(stream)), "qsx_compact_gather")
    return out_cols, count


def compact_gather_blocks(blocks, bitmaps, base_tids=None, want_tids=False, stream=None):
    """K2 over a run of blocks in one launch: blocks = per-block column lists, bitmaps = per-block TupleIdSequences.  Returns
    (output columns sized for all rows of the run, tids int32 or None, count int64[1]); the selected rows of block 0 come
    first, then block 1's, ... in row order."""
    nb = len(blocks)
    ncols = len(blocks[0]) if nb else 0
    device = bitmaps[0].device if nb else torch.device("cuda:0")
    total = sum(b[0].numel() if ncols else 0 for b in blocks) if ncols else 0
    rows_list = [b[0].numel() for b in blocks] if ncols else [0] * nb
    out_cols = [torch.empty(max(total, 1), dtype=blocks[0][c].dtype, device=device) for c in range(ncols)]
    widths = (C.c_int32 * max(ncols, 1))(*[blocks[0][c].element_size() for c in range(ncols)])
    rows = (C.c_int64 * max(nb, 1))(*rows_list)
    cptr = (C.c_void_p * max(nb * ncols, 1))()
    for i, b in enumerate(blocks):
        for c in range(ncols):
            cptr[i * ncols + c] = b[c].data_ptr() if b[c].numel() else None
    bptr = (C.c_void_p * max(nb, 1))(*[m.data_ptr() if m is not None and m.numel() else None for m in bitmaps])
    tptr = None if base_tids is None else (C.c_int32 * max(nb, 1))(*base_tids)
    tids = torch.empty(max(total, 1), dtype=torch.int32, device=device) if want_tids else None
    ws_bytes = _lib.qsx_compact_blocks_workspace_bytes(nb, rows)
    ws = torch.empty(max(ws_bytes, 8), dtype=torch.uint8, device=device)
    count = torch.zeros(1, dtype=torch.int64, device=device)
    _check(_lib.qsx_compact_gather_blocks(ncols, widths, nb, rows, cptr, bptr, tptr, _ptr_array(out_cols), _ptr(tids), _ptr(count),
                                          _ptr(ws), ws_bytes, _stream(stream)), "qsx_compact_gather_blocks")
    return out_cols, tids, count


def bitmap_to_tids(bitmap, n, base_tid=0, stream=None):
    device = bitmap.device
    out = torch.empty(max(n, 1), dtype=torch.int32, device=device)
    ws_bytes = _lib.qsx_compact_workspace_bytes(n)
    ws = torch.empty(max(ws_bytes, 8), dtype=torch.uint8, device=device)
    count = torch.zeros(1, dtype=torch.int64, device=device)
    _check(_lib.qsx_bitmap_to_tids(_ptr(bitmap), n, base_tid, _ptr(out), _ptr(count), _ptr(ws), ws_bytes,
                                   _stream(stream)), "qsx_bitmap_to_tids")
    return out, count


def tids_to_bitmap(tids, num_bits, base_tid=0, stream=None):
    out = new_bitmap(num_bits, tids.device)
    _check(_lib.qsx_tids_to_bitmap(_ptr(tids), tids.numel(), base_tid, num_bits, _ptr(out), _stream(stream)),
           "qsx_tids_to_bitmap")
    return out


def join_key_pack(cols, stream=None):
    """Composite join key -> (int64 key per row, exact flag); see qsx_join_key_pack."""
    n = cols[0].numel()
    out = torch.empty(n, dtype=torch.int64, device=cols[0].device)
    ptrs = (C.c_void_p * len(cols))(*[c.data_ptr() for c in cols])
    types = (C.c_int32 * len(cols))(*[qsx_type_of(c) for c in cols])
    exact = C.c_int(0)
    _check(_lib.qsx_join_key_pack(len(cols), ptrs, types, n, _ptr(out), C.byref(exact), _stream(stream)),
           "qsx_join_key_pack")
    return out, bool(exact.value)


def join_key_pack_char(col, stream=None):
    """CHAR(n <= 8) stripe (uint8 tensor of shape (rows, n)) -> LONG join keys."""
    n, width = col.shape
    out = torch.empty(n, dtype=torch.int64, device=col.device)
    _check(_lib.qsx_join_key_pack_char(_ptr(col), width, n, _ptr(out), _stream(stream)), "qsx_join_key_pack_char")
    return out


def join_key_pack_blocks(blocks, stream=None):
    """Composite join keys of a run of blocks (blocks[b] = the key component stripes of block b) -> (one int64 stripe of all
    blocks' packed keys, exact flag)."""
    nb, ncols = len(blocks), len(blocks[0])
    total = sum(b[0].numel() for b in blocks)
    out = torch.empty(max(total, 1), dtype=torch.int64, device=blocks[0][0].device)
    rows = (C.c_int64 * max(nb, 1))(*[b[0].numel() for b in blocks])
    ptrs = (C.c_void_p * max(nb * ncols, 1))()
    for i, b in enumerate(blocks):
        for k in range(ncols):
            ptrs[i * ncols + k] = b[k].data_ptr() if b[k].numel() else None
    types = (C.c_int32 * ncols)(*[qsx_type_of(c) for c in blocks[0]])
    exact = C.c_int(0)
    _check(_lib.qsx_join_key_pack_blocks(ncols, types, nb, rows, ptrs, _ptr(out), C.byref(exact), _stream(stream)), "qsx_join_key_pack_blocks")
    return out[:total], bool(exact.value)


def join_key_pack_blocks_coded(blocks, coding, types, stream=None):
    """qsx_join_key_pack_blocks_coded: blocks[b][k] = block b's stripe of component k as it lies, coding[b][k] = (code width or 0,
    dictionary or None), types[k] = T.INT / T.LONG of the component."""
    nb, ncols = len(blocks), len(types)
    rows_of = [(b[0].numel() if coding[i][0][0] == 0 else b[0].numel() * b[0].element_size() // coding[i][0][0]) for i, b in enumerate(blocks)]
    total = sum(rows_of)
    dev = blocks[0][0].device
    out = torch.empty(max(total, 1), dtype=torch.int64, device=dev)
    rows = (C.c_int64 * max(nb, 1))(*rows_of)
    ptrs = (C.c_void_p * max(nb * ncols, 1))()
    widths = (C.c_int32 * max(nb * ncols, 1))()
    dicts = (C.c_void_p * max(nb * ncols, 1))()
    for i, b in enumerate(blocks):
        for k in range(ncols):
            ptrs[i * ncols + k] = b[k].data_ptr() if b[k].numel() else None
            widths[i * ncols + k] = coding[i][k][0]
            dicts[i * ncols + k] = coding[i][k][1].data_ptr() if coding[i][k][1] is not None else None
    tarr = (C.c_int32 * ncols)(*types)
    exact = C.c_int(0)
    _check(_lib.qsx_join_key_pack_blocks_coded(ncols, tarr, nb, rows, ptrs, widths, dicts, _ptr(out), C.byref(exact), _stream(stream)),
           "qsx_join_key_pack_blocks_coded")
    return out[:total], bool(exact.value)


def sort_permutation(key_cols, descending=None, stream=None, types=None):
    """ORDER BY key_cols[0], key_cols[1], ... -> int32 row numbers in output order (stable)."""
    n = key_cols[0].numel()
    device = key_cols[0].device
    out = torch.empty(max(n, 1), dtype=torch.int32, device=device)
    ws_bytes = _lib.qsx_sort_workspace_bytes(n)
    ws = torch.empty(max(ws_bytes, 8), dtype=torch.uint8, device=device)
    ptrs = (C.c_void_p * len(key_cols))(*[c.data_ptr() for c in key_cols])
    types = (C.c_int32 * len(key_cols))(*(types if types is not None else [qsx_type_of(c) for c in key_cols]))
    desc = (C.c_int32 * len(key_cols))(*[1 if (descending and descending[i]) else 0 for i in range(len(key_cols))])
    _check(_lib.qsx_sort_permutation(len(key_cols), ptrs, types, desc, n, _ptr(out), _ptr(ws), ws_bytes, _stream(stream)),
           "qsx_sort_permutation")
    return out[:n]


def distinct_rows(cols, filter_bitmap=None, stream=None, types=None):
    """First row of every distinct tuple over cols (restricted to filter_bitmap), in tuple order."""
    n = cols[0].numel()
    device = cols[0].device
    out = torch.empty(max(n, 1), dtype=torch.int32, device=device)
    count = torch.zeros(1, dtype=torch.int64, device=device)
    ws_bytes = _lib.qsx_sort_workspace_bytes(n)
    ws = torch.empty(max(ws_bytes, 8), dtype=torch.uint8, device=device)
    ptrs = (C.c_void_p * len(cols))(*[c.data_ptr() for c in cols])
    types = (C.c_int32 * len(cols))(*(types if types is not None else [qsx_type_of(c) for c in cols]))
    _check(_lib.qsx_distinct_rows(len(cols), ptrs, types, n, _ptr(filter_bitmap), _ptr(out), _ptr(count), _ptr(ws), ws_bytes,
                                  _stream(stream)), "qsx_distinct_rows")
    return out[:int(count.item())]


def sort_top_k(key_cols, k, descending=None, stream=None, types=None):
    """ORDER BY ... LIMIT k -> the first min(k, n) row numbers of sort_permutation's output."""
    n = key_cols[0].numel()
    k = min(int(k), n)
    device = key_cols[0].device
    out = torch.empty(max(k, 1), dtype=torch.int32, device=device)
    ws_bytes = _lib.qsx_sort_workspace_bytes(n)
    ws = torch.empty(max(ws_bytes, 8), dtype=torch.uint8, device=device)
    ptrs = (C.c_void_p * len(key_cols))(*[c.data_ptr() for c in key_cols])
    types = (C.c_int32 * len(key_cols))(*(types if types is not None else [qsx_type_of(c) for c in key_cols]))
    desc = (C.c_int32 * len(key_cols))(*[1 if (descending and descending[i]) else 0 for i in range(len(key_cols))])
    _check(_lib.qsx_sort_top_k(len(key_cols), ptrs, types, desc, n, k, _ptr(out), _ptr(ws), ws_bytes, _stream(stream)),
           "qsx_sort_top_k")
    return out[:k]


def gather(src, tids, out=None, stream=None):
    n = tids.numel()
    if out is None:
        out = torch.empty(n, dtype=src.dtype, device=src.device)
    _check(_lib.qsx_gather(src.element_size(), _ptr(src), _ptr(tids), n, _ptr(out), _stream(stream)), "qsx_gather")
    return out


def gather_segmented(segments, first_rows, tids, out=None, stream=None):
    """K5 over a relation stored as several blocks (tids = relation-global row numbers)."""
    n = tids.numel()
    if out is None:
        out = torch.empty(n, dtype=segments[0].dtype, device=tids.device)
    starts = (C.c_int64 * len(segments))(*first_rows)
    _check(_lib.qsx_gather_segmented(segments[0].element_size(), len(segments), _ptr_array(segments), starts,
                                     _ptr(tids), n, _ptr(out), _stream(stream)), "qsx_gather_segmented")
    return out


def bitmap_gather_segmented(segment_bitmaps, first_rows, tids, stream=None):
    """Null bits of the rows `tids` (relation-global; negative = NULL padding) as a TupleIdSequence-ordered bitmap.
    segment_bitmaps[s]: int64 tensor (the segment's null bitmap words) or None (no NULLs in that segment)."""
    n = tids.numel()
    out = torch.zeros((n + 63) // 64 + 1, dtype=torch.int64, device=tids.device)
    ptrs = (C.c_void_p * len(segment_bitmaps))(*[b.data_ptr() if b is not None else None for b in segment_bitmaps])
    starts = (C.c_int64 * len(segment_bitmaps))(*first_rows)
    _check(_lib.qsx_bitmap_gather_segmented(len(segment_bitmaps), ptrs, starts, _ptr(tids), n, _ptr(out), _stream(stream)),
           "qsx_bitmap_gather_segmented")
    return out[:(n + 63) // 64]


# --------------------------------------------------------------------------- join
class KeyCoding(C.Structure):
    """qsx_key_coding_t (include/qsx.h)."""
    _fields_ = [("block_code_width", C.POINTER(C.c_int32)), ("block_dictionaries", C.POINTER(C.c_void_p))]


def _key_coding(coding, nb):
    """coding: None, or one (code_width, dictionary tensor or None) per block -> (pointer for the call, keep-alive)."""
    if coding is None:
        return None, None
    assert len(coding) == nb
    widths = (C.c_int32 * max(nb, 1))(*[w for w, _ in coding])
    dicts = (C.c_void_p * max(nb, 1))(*[d.data_ptr() if d is not None else None for _, d in coding])
    kc = KeyCoding(C.cast(widths, C.POINTER(C.c_int32)), C.cast(dicts, C.POINTER(C.c_void_p)))
    return C.byref(kc), (kc, widths, dicts)


def _rows_of(key_blocks, coding):
    """Rows per block: a plain stripe's elements, or a code stripe's bytes over its code width."""
    if coding is None:
        return [k.numel() for k in key_blocks]
    return [k.numel() if w == 0 else k.numel() * k.element_size() // w for k, (w, _) in zip(key_blocks, coding)]


class JoinTable:
    """JoinHashTable handle (qsx_join_table_t)."""

    def __init__(self, key_type, est_entries, key_range=None, key_stride=1):
        """key_range = (min_key, max_key): exact build-side statistics -> the directly addressed
        flavour (qsx_join_table_create_dense; key_stride = P for one hash partition of a dense
        domain); None -> the hashed table."""
        self.key_type = key_type
        self._h = None
        h = C.c_void_p()
        if key_range is None:
            _check(_lib.qsx_join_table_create(key_type, est_entries, C.byref(h)), "qsx_join_table_create")
        else:
            _check(_lib.qsx_join_table_create_dense(key_type, int(key_range[0]), int(key_range[1]), int(key_stride),
                                                    est_entries, C.byref(h)), "qsx_join_table_create_dense")
        self._h = h

    def close(self):
        if self._h is not None:
            _lib.qsx_join_table_destroy(self._h)
            self._h = None

    def release(self):
        """qsx_join_table_release: destroy without the wait for the device (every stream that used the table has been synchronised)."""
        if self._h is not None:
            _check(_lib.qsx_join_table_release(self._h), "qsx_join_table_release")
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def clear(self, stream=None):
        _check(_lib.qsx_join_table_clear(self._h, _stream(stream)), "qsx_join_table_clear")

    def size(self, stream=None):
        v = C.c_int64()
        _check(_lib.qsx_join_table_size(self._h, C.byref(v), _stream(stream)), "qsx_join_table_size")
        return v.value

    def build(self, keys, base_tid=0, filter_bitmap=None, stream=None):
        _check(_lib.qsx_join_build(self._h, _ptr(keys), keys.numel(), base_tid, _ptr(filter_bitmap),
                                   _stream(stream)), "qsx_join_build")

    def build_blocks(self, key_blocks, base_tids, filters=None, stream=None, coding=None):
        """K3 over a run of build blocks in one launch.  coding: one (code_width, dictionary or None) per block — the key
        stripes as a CompressedColumnStore holds them (qsx_join_build_blocks_coded)."""
        nb = len(key_blocks)
        rows = (C.c_int64 * max(nb, 1))(*_rows_of(key_blocks, coding))
        kptr = (C.c_void_p * max(nb, 1))(*[k.data_ptr() if k.numel() else None for k in key_blocks])
        bptr = (C.c_int32 * max(nb, 1))(*base_tids)
        fptr = None
        if filters is not None:
            fptr = (C.c_void_p * max(nb, 1))(*[f.data_ptr() if f is not None and f.numel() else None for f in filters])
        if coding is not None:
            cptr, keep = _key_coding(coding, nb)
            _check(_lib.qsx_join_build_blocks_coded(self._h, nb, rows, kptr, cptr, bptr, fptr, _stream(stream)), "qsx_join_build_blocks_coded")
            return
        _check(_lib.qsx_join_build_blocks(self._h, nb, rows, kptr, bptr, fptr, _stream(stream)), "qsx_join_build_blocks")

    def probe(self, keys, capacity=None, probe_base_tid=0, filter_bitmap=None, out=None, stream=None):
        """K4: returns (probe_tid int32[capacity], build_tid int32[capacity], count int64[1])."""
        n = keys.numel()
        if out is None:
            capacity = n if capacity is None else capacity
            out_p = torch.empty(max(capacity, 1), dtype=torch.int32, device=keys.device)
            out_b = torch.empty(max(capacity, 1), dtype=torch.int32, device=keys.device)
            count = torch.zeros(1, dtype=torch.int64, device=keys.device)
        else:
            out_p, out_b, count = out
            capacity = out_p.numel() if capacity is None else capacity
        _check(_lib.qsx_join_probe(self._h, _ptr(keys), n, probe_base_tid, _ptr(filter_bitmap), _ptr(out_p),
                                   _ptr(out_b), capacity, _ptr(count), _stream(stream)), "qsx_join_probe")
        return out_p, out_b, count

    def probe_exists_lip(self, keys, lip_filters, filter_bitmap=None, stream=None):
        """qsx_join_probe_exists_lip: the semi probe with the work order's LIP filters tested inside: (bitmap, count int64[1])."""
        n = keys.numel()
        out = new_bitmap(n, keys.device)
        count = torch.zeros(1, dtype=torch.int64, device=keys.device)
        fptr = (C.c_void_p * max(len(lip_filters), 1))(*[f._h for f in lip_filters])
        _check(_lib.qsx_join_probe_exists_lip(self._h, _ptr(keys), n, _ptr(filter_bitmap), len(lip_filters), fptr, _ptr(out), _ptr(count),
                                              _stream(stream)), "qsx_join_probe_exists_lip")
        return out, count

    def probe_lip(self, keys, lip_filters, capacity=None, probe_base_tid=0, filter_bitmap=None, out=None, stream=None):
        """qsx_join_probe_lip: the probe with the work order's LIP filters (LipFilter objects over the probe key) tested inside."""
        n = keys.numel()
        if out is None:
            capacity = n if capacity is None else capacity
            out_p = torch.empty(max(capacity, 1), dtype=torch.int32, device=keys.device)
            out_b = torch.empty(max(capacity, 1), dtype=torch.int32, device=keys.device)
            count = torch.zeros(1, dtype=torch.int64, device=keys.device)
        else:
            out_p, out_b, count = out
            capacity = out_p.numel() if capacity is None else capacity
        handles = (C.c_void_p * max(len(lip_filters), 1))(*[f._h for f in lip_filters])
        _check(_lib.qsx_join_probe_lip(self._h, _ptr(keys), n, probe_base_tid, _ptr(filter_bitmap), len(lip_filters), handles, _ptr(out_p),
                                       _ptr(out_b), capacity, _ptr(count), _stream(stream)), "qsx_join_probe_lip")
        return out_p, out_b, count

    def probe_count(self, keys, filter_bitmap=None, stream=None):
        count = torch.zeros(1, dtype=torch.int64, device=keys.device)
        _check(_lib.qsx_join_probe_count(self._h, _ptr(keys), keys.numel(), _ptr(filter_bitmap), _ptr(count),
                                         _stream(stream)), "qsx_join_probe_count")
        return count

    def probe_exists(self, keys, anti=False, filter_bitmap=None, stream=None):
        n = keys.numel()
        out = new_bitmap(n, keys.device)
        count = torch.zeros(1, dtype=torch.int64, device=keys.device)
        _check(_lib.qsx_join_probe_exists(self._h, _ptr(keys), n, _ptr(filter_bitmap), 1 if anti else 0, _ptr(out),
                                          _ptr(count), _stream(stream)), "qsx_join_probe_exists")
        return out, count

    def probe_blocks(self, key_blocks, capacity=None, base_tids=None, filters=None, out=None, stream=None, coding=None):
        """K4 over a run of probe blocks in one launch: key_blocks = one key stripe per block.  Probe tids are
        base_tids[b] + row, or run-global row numbers when base_tids is None.  coding: as in build_blocks."""
        nb = len(key_blocks)
        dev = key_blocks[0].device if nb else torch.device("cuda:0")
        total = sum(_rows_of(key_blocks, coding))
        if out is None:
            capacity = total if capacity is None else capacity
            out_p = torch.empty(max(capacity, 1), dtype=torch.int32, device=dev)
            out_b = torch.empty(max(capacity, 1), dtype=torch.int32, device=dev)
            count = torch.zeros(1, dtype=torch.int64, device=dev)
        else:
            out_p, out_b, count = out
            capacity = out_p.numel() if capacity is None else capacity
        rows = (C.c_int64 * max(nb, 1))(*_rows_of(key_blocks, coding))
        kptr = (C.c_void_p * max(nb, 1))(*[k.data_ptr() if k.numel() else None for k in key_blocks])
        bptr = None if base_tids is None else (C.c_int32 * max(nb, 1))(*base_tids)
        fptr = None
        if filters is not None:
            fptr = (C.c_void_p * max(nb, 1))(*[f.data_ptr() if f is not None and f.numel() else None for f in filters])
        if coding is not None:
            cptr, keep = _key_coding(coding, nb)
            _check(_lib.qsx_join_probe_blocks_coded(self._h, nb, rows, kptr, cptr, bptr, fptr, _ptr(out_p), _ptr(out_b), capacity, _ptr(count),
                                                    _stream(stream)), "qsx_join_probe_blocks_coded")
            return out_p, out_b, count
        _check(_lib.qsx_join_probe_blocks(self._h, nb, rows, kptr, bptr, fptr, _ptr(out_p), _ptr(out_b), capacity, _ptr(count),
                                          _stream(stream)), "qsx_join_probe_blocks")
        return out_p, out_b, count

    def probe_project_blocks(self, key_blocks, probe_columns, build_columns, build_first_tids=None, capacity=None, filters=None,
                             stream=None, coding=None, key_dtype=None):
        """K4 + K5 in one pass: the output relation of the inner join over a run of probe blocks.
        probe_columns: list of per-attribute lists of per-block stripes (probe_columns[a][b]); build_columns: list of
        per-attribute lists of per-segment stripes (build_columns[a][s]), segment s starting at build tuple id
        build_first_tids[s].  Returns (output columns — the probe attributes first, then the build attributes —, count)."""
        nb = len(key_blocks)
        dev = key_blocks[0].device if nb else torch.device("cuda:0")
        total = sum(_rows_of(key_blocks, coding))
        capacity = total if capacity is None else capacity
        sources = [(0, c) for c in probe_columns] + [(1, c) for c in build_columns]
        nc = len(sources)
        nseg = len(build_columns[0]) if build_columns else 0
        if build_first_tids is None:
            build_first_tids, at = [], 0
            for sg in range(nseg):
                build_first_tids.append(at)
                at += build_columns[0][sg].numel()
        proj = T.JoinProjection()
        proj.num_columns = nc
        outs = []
        pstripes = (C.c_void_p * max(nb * nc, 1))()
        bstripes = (C.c_void_p * max(nseg * nc, 1))()
        for c, (side, stripes) in enumerate(sources):
            ref = stripes[0]
            # (coded runs: a probe column given as the key stripes themselves is the join key, emitted as its value)
            is_key = coding is not None and side == 0 and all(a is b for a, b in zip(stripes, key_blocks))
            dtype = key_dtype if is_key else ref.dtype
            proj.width[c] = torch.empty(0, dtype=dtype).element_size()
            proj.on_build[c] = side
            outs.append(torch.empty(max(capacity, 1), dtype=dtype, device=dev))
            for i, stripe in enumerate(stripes):
                (bstripes if side else pstripes)[i * nc + c] = stripe.data_ptr() if stripe.numel() else None
        optr = (C.c_void_p * nc)(*[o.data_ptr() for o in outs])
        first = (C.c_int64 * max(nseg, 1))(*build_first_tids)
        proj.probe_stripes = C.cast(pstripes, C.POINTER(C.c_void_p))
        proj.num_build_segments = nseg
        proj.build_first_tids = C.cast(first, C.POINTER(C.c_int64))
        proj.build_stripes = C.cast(bstripes, C.POINTER(C.c_void_p))
        proj.out_columns = C.cast(optr, C.POINTER(C.c_void_p))
        count = torch.zeros(1, dtype=torch.int64, device=dev)
        rows = (C.c_int64 * max(nb, 1))(*_rows_of(key_blocks, coding))
        kptr = (C.c_void_p * max(nb, 1))(*[k.data_ptr() if k.numel() else None for k in key_blocks])
        fptr = None
        if filters is not None:
            fptr = (C.c_void_p * max(nb, 1))(*[f.data_ptr() if f is not None and f.numel() else None for f in filters])
        if coding is not None:
            cptr, keep = _key_coding(coding, nb)
            _check(_lib.qsx_join_probe_project_blocks_coded(self._h, nb, rows, kptr, cptr, fptr, C.byref(proj), capacity, _ptr(count),
                                                            _stream(stream)), "qsx_join_probe_project_blocks_coded")
            return outs, count
        _check(_lib.qsx_join_probe_project_blocks(self._h, nb, rows, kptr, fptr, C.byref(proj), capacity, _ptr(count), _stream(stream)),
               "qsx_join_probe_project_blocks")
        return outs, count

    def probe_count_blocks(self, key_blocks, filters=None, stream=None, coding=None):
        nb = len(key_blocks)
        dev = key_blocks[0].device if nb else torch.device("cuda:0")
        count = torch.zeros(1, dtype=torch.int64, device=dev)
        rows = (C.c_int64 * max(nb, 1))(*_rows_of(key_blocks, coding))
        kptr = (C.c_void_p * max(nb, 1))(*[k.data_ptr() if k.numel() else None for k in key_blocks])
        fptr = None
        if filters is not None:
            fptr = (C.c_void_p * max(nb, 1))(*[f.data_ptr() if f is not None and f.numel() else None for f in filters])
        if coding is not None:
            cptr, keep = _key_coding(coding, nb)
            _check(_lib.qsx_join_probe_count_blocks_coded(self._h, nb, rows, kptr, cptr, fptr, _ptr(count), _stream(stream)),
                   "qsx_join_probe_count_blocks_coded")
            return count
        _check(_lib.qsx_join_probe_count_blocks(self._h, nb, rows, kptr, fptr, _ptr(count), _stream(stream)), "qsx_join_probe_count_blocks")
        return count

    def probe_exists_blocks(self, key_blocks, anti=False, filters=None, out_bitmaps=None, stream=None, coding=None):
        """Semi / anti probe over a run of blocks: returns (per-block bitmaps, total count int64[1])."""
        nb = len(key_blocks)
        dev = key_blocks[0].device if nb else torch.device("cuda:0")
        outs = out_bitmaps if out_bitmaps is not None else [new_bitmap(r, dev) for r in _rows_of(key_blocks, coding)]
        count = torch.zeros(1, dtype=torch.int64, device=dev)
        rows = (C.c_int64 * max(nb, 1))(*_rows_of(key_blocks, coding))
        kptr = (C.c_void_p * max(nb, 1))(*[k.data_ptr() if k.numel() else None for k in key_blocks])
        optr = (C.c_void_p * max(nb, 1))(*[o.data_ptr() if o.numel() else None for o in outs])
        fptr = None
        if filters is not None:
            fptr = (C.c_void_p * max(nb, 1))(*[f.data_ptr() if f is not None and f.numel() else None for f in filters])
        if coding is not None:
            cptr, keep = _key_coding(coding, nb)
            _check(_lib.qsx_join_probe_exists_blocks_coded(self._h, nb, rows, kptr, cptr, fptr, 1 if anti else 0, optr, _ptr(count),
                                                           _stream(stream)), "qsx_join_probe_exists_blocks_coded")
            return outs, count
        _check(_lib.qsx_join_probe_exists_blocks(self._h, nb, rows, kptr, fptr, 1 if anti else 0, optr, _ptr(count),
                                                 _stream(stream)), "qsx_join_probe_exists_blocks")
        return outs, count


# --------------------------------------------------------------------------- aggregation
class AggState:
    """AggregationOperationState handle (qsx_agg_state_t)."""

    def __init__(self, config):
        self.config = config
        h = C.c_void_p()
        _check(_lib.qsx_agg_state_create(C.byref(config), C.byref(h)), "qsx_agg_state_create")
        self._h = h
        self.device = torch.device("cuda", torch.cuda.current_device())

    def close(self):
        if self._h is not None:
            _lib.qsx_agg_state_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def clear(self, stream=None):
        _check(_lib.qsx_agg_state_clear(self._h, _stream(stream)), "qsx_agg_state_clear")

    def update(self, cols, n=None, filter_bitmap=None, stream=None):
        if n is None:
            n = cols[0].numel()
        _check(_lib.qsx_agg_update(self._h, _ptr_array(cols), n, _ptr(filter_bitmap), _stream(stream)),
               "qsx_agg_update")

    def update_blocks(self, blocks, filters=None, stream=None):
        """One launch over a run of blocks: blocks = list of per-block column lists (every block its own stripes); filters =
        per-block bitmaps (None entries allowed) or None."""
        nb, ncols = len(blocks), self.config.num_columns
        rows = (C.c_int64 * max(nb, 1))(*[b[0].numel() if b else 0 for b in blocks])
        ptrs = (C.c_void_p * max(nb * ncols, 1))()
        for i, b in enumerate(blocks):
            for c in range(ncols):
                ptrs[i * ncols + c] = b[c].data_ptr() if c < len(b) and b[c] is not None else None
        fptr = None
        if filters is not None:
            fptr = (C.c_void_p * max(nb, 1))(*[f.data_ptr() if f is not None else None for f in filters])
        _check(_lib.qsx_agg_update_blocks(self._h, nb, rows, ptrs, fptr, _stream(stream)), "qsx_agg_update_blocks")

    def update_coded_blocks(self, blocks, dictionaries, filters=None, stream=None, sized=True):
        """qsx_agg_update_coded over a run of blocks: blocks[b][c] = code / value stripe, dictionaries[b][c] = dictionary or None.
        sized: every block's dictionary sizes travel with the call (qsx_agg_update_coded_blocks_sized)."""
        nb, ncols = len(blocks), self.config.num_columns
        rows = (C.c_int64 * max(nb, 1))(*[b[0].numel() if b else 0 for b in blocks])
        ptrs = (C.c_void_p * max(nb * ncols, 1))()
        dptr = (C.c_void_p * max(nb * ncols, 1))()
        for i, b in enumerate(blocks):
            for c in range(ncols):
                ptrs[i * ncols + c] = b[c].data_ptr() if c < len(b) and b[c] is not None else None
                d = dictionaries[i][c] if c < len(dictionaries[i]) else None
                dptr[i * ncols + c] = d.data_ptr() if d is not None else None
        fptr = None
        if filters is not None:
            fptr = (C.c_void_p * max(nb, 1))(*[f.data_ptr() if f is not None else None for f in filters])
        if sized:
            entries = (C.c_int32 * max(nb * ncols, 1))()
            for i in range(nb):
                for c in range(ncols):
                    d = dictionaries[i][c] if c < len(dictionaries[i]) else None
                    entries[i * ncols + c] = d.numel() if d is not None else 0
            _check(_lib.qsx_agg_update_coded_blocks_sized(self._h, nb, rows, ptrs, dptr, entries, fptr, _stream(stream)),
                   "qsx_agg_update_coded_blocks_sized")
            return
        _check(_lib.qsx_agg_update_coded_blocks(self._h, nb, rows, ptrs, dptr, fptr, _stream(stream)), "qsx_agg_update_coded_blocks")

    def update_nullable(self, cols, null_bitmaps, n=None, filter_bitmap=None, stream=None):
        """null_bitmaps[c]: int64 tensor with the null bitmap words of column c (TupleIdSequence bit order) or None."""
        if n is None:
            n = cols[0].numel()
        nulls = (C.c_void_p * len(cols))(*[b.data_ptr() if b is not None else None for b in null_bitmaps])
        _check(_lib.qsx_agg_update_nullable(self._h, _ptr_array(cols), nulls, n, _ptr(filter_bitmap), _stream(stream)),
               "qsx_agg_update_nullable")

    def update_coded(self, cols, dictionaries, n=None, filter_bitmap=None, stream=None, sized=True):
        """cols[c] = code stripe for columns with column_code_width != 0; dictionaries[c] = dictionary tensor or None."""
        if n is None:
            n = cols[0].numel()
        dicts = (C.c_void_p * len(cols))(*[d.data_ptr() if d is not None else None for d in dictionaries])
        if sized:      # the dictionaries' sizes travel with the call (small ones are then decoded from LDS)
            entries = (C.c_int32 * len(cols))(*[d.numel() if d is not None else 0 for d in dictionaries])
            _check(_lib.qsx_agg_update_coded_sized(self._h, _ptr_array(cols), dicts, entries, n, _ptr(filter_bitmap), _stream(stream)),
                   "qsx_agg_update_coded_sized")
            return
        _check(_lib.qsx_agg_update_coded(self._h, _ptr_array(cols), dicts, n, _ptr(filter_bitmap), _stream(stream)),
               "qsx_agg_update_coded")

    def mark_existence(self, keys, filter_bitmap=None, stream=None):
        _check(_lib.qsx_agg_mark_existence(self._h, qsx_type_of(keys), _ptr(keys), keys.numel(), _ptr(filter_bitmap),
                                           _stream(stream)), "qsx_agg_mark_existence")

    def merge(self, other, stream=None):
        _check(_lib.qsx_agg_merge(self._h, other._h, _stream(stream)), "qsx_agg_merge")

    def export_bytes(self, stream=None):
        """Size of the image right now (a hash-strategy table may have grown); synchronises."""
        v = C.c_size_t()
        _check(_lib.qsx_agg_state_export_bytes(self._h, C.byref(v), _stream(stream)), "qsx_agg_state_export_bytes")
        return v.value

    def export(self, device, stream=None):
        """Raw image as an int64 tensor (8-byte words)."""
        nbytes = self.export_bytes(stream)
        out = torch.empty(nbytes // 8, dtype=torch.int64, device=device)
        _check(_lib.qsx_agg_state_export(self._h, _ptr(out), nbytes, _stream(stream)), "qsx_agg_state_export")
        return out

    def image_layout(self):
        """(dense, header_words, words_per_column, column_kinds): how an exported image is laid out and how two partial values
        of each column combine (T.ACC_SUM_F64 / ACC_SUM_I64 / ACC_MIN_I64 / ACC_MAX_I64)."""
        dense, header, per_col, ncols = C.c_int(), C.c_int64(), C.c_int64(), C.c_int()
        kinds = (C.c_int32 * 32)()
        _check(_lib.qsx_agg_state_image_layout(self._h, C.byref(dense), C.byref(header), C.byref(per_col), C.byref(ncols), kinds, 32),
               "qsx_agg_state_image_layout")
        return bool(dense.value), header.value, per_col.value, [int(kinds[i]) for i in range(ncols.value)]

    def import_merge(self, image, stream=None):
        """image: int64 tensor holding exactly one exported image (its length tells the source table's capacity)."""
        _check(_lib.qsx_agg_state_import_merge(self._h, _ptr(image), image.numel() * image.element_size(), _stream(stream)),
               "qsx_agg_state_import_merge")

    def num_groups(self, stream=None):
        v = C.c_int64()
        _check(_lib.qsx_agg_num_groups(self._h, C.byref(v), _stream(stream)), "qsx_agg_num_groups")
        return v.value

    def finalize(self, device, partition=0, num_partitions=1, capacity=None, stream=None):
        """K10: returns (key columns, value columns, null columns (uint8), groups int64[1])."""
        cfg = self.config
        if capacity is None:
            capacity = max(self.num_groups(stream), 1)
        keys = []
        for k in range(cfg.num_keys):
            w = cfg.column_width[cfg.key_column[k]]
            dt = {1: torch.uint8, 2: torch.int16, 4: torch.int32, 8: torch.int64}[w]
            keys.append(torch.zeros(capacity, dtype=dt, device=device))
        vals, nulls = [], []
        for a in range(cfg.num_aggs):
            dt = getattr(torch, T.agg_output_dtype(cfg, a))
            vals.append(torch.zeros(capacity, dtype=dt, device=device))
            nulls.append(torch.zeros(capacity, dtype=torch.uint8, device=device))
        groups = torch.zeros(1, dtype=torch.int64, device=device)
        _check(_lib.qsx_agg_finalize(self._h, partition, num_partitions, _ptr_array(keys), _ptr_array(vals),
                                     _ptr_array(nulls), capacity, _ptr(groups), _stream(stream)), "qsx_agg_finalize")
        return keys, vals, nulls, groups


# --------------------------------------------------------------------------- LIP
class LipFilter:
    def __init__(self, kind, cardinality, min_value=0, is_anti=False):
        h = C.c_void_p()
        _check(_lib.qsx_lip_filter_create(kind, cardinality, min_value, 1 if is_anti else 0, C.byref(h)),
               "qsx_lip_filter_create")
        self._h = h

    def close(self):
        if self._h is not None:
            _lib.qsx_lip_filter_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def build(self, keys, filter_bitmap=None, stream=None):
        _check(_lib.qsx_lip_build(self._h, qsx_type_of(keys), _ptr(keys), keys.numel(), _ptr(filter_bitmap),
                                  _stream(stream)), "qsx_lip_build")

    def build_from_table(self, table, num_new_keys=-1, stream=None):
        """qsx_lip_build_from_join_table: the bits of every key a directly addressed JoinTable holds, read off the table.  False —
        nothing done, call build() — when the table / filter are not of those kinds or reading the key range would cost more
        than num_new_keys atomics (< 0: whatever it costs)."""
        rc = _lib.qsx_lip_build_from_join_table(self._h, table._h, num_new_keys, _stream(stream))
        if rc == T.ERR_UNSUPPORTED:
            return False
        _check(rc, "qsx_lip_build_from_join_table")
        return True

    def probe(self, keys, in_bitmap=None, stream=None):
        n = keys.numel()
        out = new_bitmap(n, keys.device)
        count = torch.zeros(1, dtype=torch.int64, device=keys.device)
        _check(_lib.qsx_lip_probe(self._h, qsx_type_of(keys), _ptr(keys), n, _ptr(in_bitmap), _ptr(out), _ptr(count),
                                  _stream(stream)), "qsx_lip_probe")
        return out, count

    def build_blocks(self, key_blocks, filters=None, stream=None, coding=None, key_type=None):
        """qsx_lip_build over a run of blocks in one launch (coding / key_type: compressed key stripes, JoinTable.build_blocks)."""
        nb = len(key_blocks)
        rows = (C.c_int64 * max(nb, 1))(*_rows_of(key_blocks, coding))
        kptr = (C.c_void_p * max(nb, 1))(*[k.data_ptr() if k.numel() else None for k in key_blocks])
        fptr = None
        if filters is not None:
            fptr = (C.c_void_p * max(nb, 1))(*[f.data_ptr() if f is not None and f.numel() else None for f in filters])
        if coding is not None:
            cptr, keep = _key_coding(coding, nb)
            _check(_lib.qsx_lip_build_blocks_coded(self._h, key_type, nb, rows, kptr, cptr, fptr, _stream(stream)), "qsx_lip_build_blocks_coded")
            return
        _check(_lib.qsx_lip_build_blocks(self._h, qsx_type_of(key_blocks[0]) if nb else T.INT, nb, rows, kptr, fptr, _stream(stream)),
               "qsx_lip_build_blocks")

    def probe_blocks(self, key_blocks, in_bitmaps=None, stream=None, coding=None, key_type=None):
        """qsx_lip_probe over a run of blocks in one launch: (per-block output bitmaps, total count int64[1])."""
        nb = len(key_blocks)
        dev = key_blocks[0].device if nb else torch.device("cuda:0")
        outs = [new_bitmap(r, dev) for r in _rows_of(key_blocks, coding)]
        count = torch.zeros(1, dtype=torch.int64, device=dev)
        rows = (C.c_int64 * max(nb, 1))(*_rows_of(key_blocks, coding))
        kptr = (C.c_void_p * max(nb, 1))(*[k.data_ptr() if k.numel() else None for k in key_blocks])
        optr = (C.c_void_p * max(nb, 1))(*[o.data_ptr() if o.numel() else None for o in outs])
        iptr = None
        if in_bitmaps is not None:
            iptr = (C.c_void_p * max(nb, 1))(*[f.data_ptr() if f is not None and f.numel() else None for f in in_bitmaps])
        if coding is not None:
            cptr, keep = _key_coding(coding, nb)
            _check(_lib.qsx_lip_probe_blocks_coded(self._h, key_type, nb, rows, kptr, cptr, iptr, optr, _ptr(count), _stream(stream)),
                   "qsx_lip_probe_blocks_coded")
            return outs, count
        _check(_lib.qsx_lip_probe_blocks(self._h, qsx_type_of(key_blocks[0]) if nb else T.INT, nb, rows, kptr, iptr, optr, _ptr(count),
                                         _stream(stream)), "qsx_lip_probe_blocks")
        return outs, count

    def words(self):
        """(device pointer, number of 64-bit words) of the raw LSB-first bit array."""
        p = C.c_void_p()
        nw = C.c_int64()
        _check(_lib.qsx_lip_filter_words(self._h, C.byref(p), C.byref(nw)), "qsx_lip_filter_words")
        return p.value, nw.value

    def clear(self, stream=None):
        p, nw = self.words()
        _check(_lib.qsx_memset_device(C.c_void_p(p), 0, nw * 8, _stream(stream)), "qsx_memset_device")

    def export(self, device, stream=None):
        """Copy of the raw bit array as an int64 tensor (what a multi-GPU plan all-reduces with OR)."""
        p, nw = self.words()
        out = torch.empty(nw, dtype=torch.int64, device=device)
        _check(_lib.qsx_copy_on_device(_ptr(out), C.c_void_p(p), nw * 8, _stream(stream)), "qsx_copy_on_device")
        return out

    def merge_or(self, words, stream=None):
        """filter |= words (an exported image of a filter of the same kind and cardinality: another rank's bits)."""
        p, nw = self.words()
        assert words.numel() == nw and words.element_size() == 8
        _check(_lib.qsx_bitmap_combine(1, C.c_void_p(p), _ptr(words), nw * 64, C.c_void_p(p), _stream(stream)),
               "qsx_bitmap_combine")


# --------------------------------------------------------------------------- partition
def partition_scatter(keys, num_partitions, cols, stream=None):
    """K9: returns (scattered columns, offsets int64[P+1] on device)."""
    n = keys.numel()
    device = keys.device
    out_cols = [torch.empty_like(c) for c in cols]
    widths = (C.c_int32 * max(len(cols), 1))(*[c.element_size() for c in cols])
    ws_bytes = _lib.qsx_partition_workspace_bytes(n, num_partitions)
    ws = torch.empty(max(ws_bytes, 8), dtype=torch.uint8, device=device)
    offsets = torch.zeros(num_partitions + 1, dtype=torch.int64, device=device)
    _check(_lib.qsx_partition_scatter(qsx_type_of(keys), _ptr(keys), n, num_partitions, len(cols), _ptr_array(cols),
                                      widths, _ptr_array(out_cols), _ptr(offsets), _ptr(ws), ws_bytes,
                                      _stream(stream)), "qsx_partition_scatter")
    return out_cols, offsets


def partition_scatter_blocks(block_keys, num_partitions, block_cols, stream=None):
    """K9 over a run of blocks (qsx_partition_scatter_blocks): block_keys[b] the key stripe of block b, block_cols[b][c] column c's
    stripe of block b.  Returns (scattered columns over all rows, offsets int64[P+1] on device) — what partition_scatter returns for
    the blocks' rows laid end to end."""
    nb = len(block_keys)
    assert nb > 0 and len(block_cols) == nb
    device = block_keys[0].device
    ncols = len(block_cols[0])
    rows = [k.numel() for k in block_keys]
    n = sum(rows)
    out_cols = [torch.empty(n, dtype=block_cols[0][c].dtype, device=device) for c in range(ncols)]
    widths = (C.c_int32 * max(ncols, 1))(*[block_cols[0][c].element_size() for c in range(ncols)])
    flat = [block_cols[b][c] for b in range(nb) for c in range(ncols)]
    ws_bytes = _lib.qsx_partition_blocks_workspace_bytes(n, nb, num_partitions)
    ws = torch.empty(max(ws_bytes, 8), dtype=torch.uint8, device=device)
    offsets = torch.zeros(num_partitions + 1, dtype=torch.int64, device=device)
    _check(_lib.qsx_partition_scatter_blocks(qsx_type_of(block_keys[0]), nb, (C.c_int64 * nb)(*rows), _ptr_array(block_keys), num_partitions,
                                             ncols, _ptr_array(flat), widths, _ptr_array(out_cols), _ptr(offsets), _ptr(ws), ws_bytes,
                                             _stream(stream)), "qsx_partition_scatter_blocks")
    return out_cols, offsets


# --------------------------------------------------------------------------- multi-GPU (RCCL behind the C ABI)
def dense_partition_range(num_entries, num_partitions, partition):
    """qsx_agg_dense_partition_range (host arithmetic, needs no device): (begin, end, first_word, last_word, first_mask,
    last_mask) of finalize partition `partition` of a COLLISION_FREE state — the split qsx_agg_finalize and
    qsx_agg_reduce_scatter use."""
    b, e, fw, lw = _i64(), _i64(), _i64(), _i64()
    fm, lm = C.c_uint64(), C.c_uint64()
    _check(_lib.qsx_agg_dense_partition_range(num_entries, num_partitions, partition, C.byref(b), C.byref(e), C.byref(fw), C.byref(lw),
                                              C.byref(fm), C.byref(lm)), "qsx_agg_dense_partition_range")
    return b.value, e.value, fw.value, lw.value, fm.value, lm.value


class Comm:
    """qsx_comm_t: one per rank.  unique_id() on rank 0, carried to the other ranks by the caller's control plane
    (torch.distributed's store, an engine's message bus), then Comm(world, rank, id_bytes) on every rank's own device."""

    @staticmethod
    def unique_id():
        buf = C.create_string_buffer(128)
        _check(_lib.qsx_comm_unique_id(buf), "qsx_comm_unique_id")
        return buf.raw

    def __init__(self, world, rank, id_bytes):
        h = C.c_void_p()
        self._h = None
        _check(_lib.qsx_comm_create(world, rank, C.c_char_p(id_bytes), C.byref(h)), "qsx_comm_create")
        self._h = h
        self.world, self.rank = world, rank

    def close(self):
        if self._h is not None:
            _lib.qsx_comm_destroy(self._h)
            self._h = None

    def agree(self, local_status=0, stream=None):
        """Failure agreement before a collective step: raises on EVERY rank when any rank contributed a non-zero status."""
        _check(_lib.qsx_comm_agree(self._h, int(local_status), _stream(stream)), "qsx_comm_agree")

    def synchronize(self, stream=None):
        """Wait for the stream under the communicator's watchdog (QSX_COMM_TIMEOUT_MS)."""
        _check(_lib.qsx_comm_synchronize(self._h, _stream(stream)), "qsx_comm_synchronize")

    def abort(self):
        _check(_lib.qsx_comm_abort(self._h), "qsx_comm_abort")

    def exchange_counts(self, send_counts, stream=None):
        recv = torch.empty_like(send_counts)
        _check(_lib.qsx_exchange_counts(self._h, _ptr(send_counts), _ptr(recv), _stream(stream)), "qsx_exchange_counts")
        return recv

    def alltoallv(self, col, send_rows, recv_rows, stream=None, out=None):
        """col: rows for rank 0, then rank 1, ... (qsx_partition_scatter's layout); returns the rows received, in rank order."""
        if out is None:
            out = torch.empty(int(sum(recv_rows)), dtype=col.dtype, device=col.device)
        assert out.numel() == int(sum(recv_rows)) and col.numel() >= int(sum(send_rows)) and out.dtype == col.dtype
        assert col.is_contiguous() and out.is_contiguous()
        sr = (C.c_int64 * self.world)(*[int(x) for x in send_rows])
        rr = (C.c_int64 * self.world)(*[int(x) for x in recv_rows])
        _check(_lib.qsx_alltoallv(self._h, col.element_size(), _ptr(col), sr, _ptr(out), rr, _stream(stream)), "qsx_alltoallv")
        return out

    def allgather(self, t, stream=None, out=None):
        if out is None:
            out = torch.empty(self.world * t.numel(), dtype=t.dtype, device=t.device)
        assert out.numel() == self.world * t.numel() and out.dtype == t.dtype and out.is_contiguous() and t.is_contiguous()
        _check(_lib.qsx_allgather(self._h, _ptr(t), t.numel() * t.element_size(), _ptr(out), _stream(stream)), "qsx_allgather")
        return out

    def bitmap_allreduce_or(self, words, stream=None):
        _check(_lib.qsx_bitmap_allreduce_or(self._h, _ptr(words), words.numel(), _stream(stream)), "qsx_bitmap_allreduce_or")
        return words

    def agg_reduce_scatter(self, state, stream=None):
        _check(_lib.qsx_agg_reduce_scatter(self._h, state._h, _stream(stream)), "qsx_agg_reduce_scatter")

    def agg_allgather_merge(self, state, stream=None):
        _check(_lib.qsx_agg_allgather_merge(self._h, state._h, _stream(stream)), "qsx_agg_allgather_merge")
