"""ctypes mirror of the descriptor structs and enums of include/qsx.h.

Shared by the product binding (quickstep_amd.capi) and by the test-only CPU
checker's binding so that a test hands the *same* descriptor bytes to both
sides.  Pure data definitions: no library is loaded here.
"""
import ctypes as C

# qsx_type_t (numbering of types/TypeID.hpp:32-43 in the reference)
ACC_SUM_F64, ACC_SUM_I64, ACC_MIN_I64, ACC_MAX_I64 = 0, 1, 2, 3      # qsx_agg_state_image_layout column kinds
ABI_VERSION = 19                                                    # QSX_ABI_VERSION of include/qsx.h
INT, LONG, FLOAT, DOUBLE, CHAR = 0, 1, 2, 3, 4
DATE = 6   # the reference's 8-byte DateLit {int32 year; uint8 month, day; 2 bytes padding}, carried as int64 raw bytes
# qsx_cmp_t (types/operations/comparisons/ComparisonID.hpp:36-42)
EQ, NE, LT, LE, GT, GE = range(6)
CODE_EQ, CODE_NE, CODE_LT, CODE_GE, CODE_RANGE = range(5)            # qsx_code_cmp_t
# qsx_agg_strategy_t
AGG_SINGLE_STATE, AGG_COMPACT_KEY, AGG_COLLISION_FREE, AGG_GENERIC = range(4)
# qsx_agg_fn_t
AGG_COUNT_STAR, AGG_SUM, AGG_AVG, AGG_MIN, AGG_MAX, AGG_COUNT = range(6)
# qsx_operand_kind_t
OPD_COLUMN, OPD_CONST, OPD_TEMP = range(3)
# qsx_expr_op_t
EX_ADD, EX_SUB, EX_MUL, EX_DIV = range(4)
# qsx_lip_kind_t
LIP_SINGLE_IDENTITY_HASH, LIP_BITVECTOR_EXACT = range(2)

MAX_COLUMNS, MAX_KEYS, MAX_AGGS, MAX_INSTRS, MAX_TEMPS, MAX_CONSTS, MAX_PRED_TERMS = 16, 4, 8, 16, 8, 8, 4

TYPE_WIDTH = {INT: 4, LONG: 8, FLOAT: 4, DOUBLE: 8, DATE: 8}


def date_raw(year, month, day, padding=0):
    """The 8 bytes of a DateLit as a (signed) 64-bit integer; `padding` fills the two unused bytes."""
    v = (int(year) & 0xFFFFFFFF) | (int(month) & 0xFF) << 32 | (int(day) & 0xFF) << 40 | (int(padding) & 0xFFFF) << 48
    return v - (1 << 64) if v >= 1 << 63 else v

# status codes
OK = 0
ERR_INVALID_ARGUMENT, ERR_NO_DEVICE, ERR_OUT_OF_MEMORY, ERR_HIP = -1, -2, -3, -4
ERR_CAPACITY, ERR_UNSUPPORTED, ERR_TOO_MANY_GROUPS, ERR_HASH_COLLISION, ERR_COMM = -5, -6, -7, -8, -9
GROUPS_HASH_COLLISION = -1   # qsx_agg_finalize's group count when a wide-key state saw two keys under one hash


class Operand(C.Structure):
    _fields_ = [("kind", C.c_int32), ("index", C.c_int32)]


class ExprInstr(C.Structure):
    _fields_ = [("op", C.c_int32), ("dst", C.c_int32), ("a", Operand), ("b", Operand)]


class AggDesc(C.Structure):
    _fields_ = [("fn", C.c_int32), ("arg", Operand)]


class PredLiteral(C.Union):
    _fields_ = [("i32", C.c_int32), ("i64", C.c_int64), ("f32", C.c_float), ("f64", C.c_double)]


class PredTerm(C.Structure):
    _fields_ = [("column", C.c_int32), ("op", C.c_int32), ("literal", PredLiteral)]


class AggConfig(C.Structure):
    _fields_ = [
        ("strategy", C.c_int32),
        ("num_columns", C.c_int32),
        ("column_type", C.c_int32 * MAX_COLUMNS),
        ("column_width", C.c_int32 * MAX_COLUMNS),
        ("num_keys", C.c_int32),
        ("key_column", C.c_int32 * MAX_KEYS),
        ("num_instrs", C.c_int32),
        ("instrs", ExprInstr * MAX_INSTRS),
        ("consts", C.c_double * MAX_CONSTS),
        ("num_aggs", C.c_int32),
        ("aggs", AggDesc * MAX_AGGS),
        ("num_pred_terms", C.c_int32),
        ("pred", PredTerm * MAX_PRED_TERMS),
        ("est_groups", C.c_int64),
        ("num_entries", C.c_int64),
        ("column_code_width", C.c_int32 * MAX_COLUMNS),
        ("column_nullable", C.c_int32 * MAX_COLUMNS),
    ]


def col(i):
    return Operand(OPD_COLUMN, i)


def const(i):
    return Operand(OPD_CONST, i)


def temp(i):
    return Operand(OPD_TEMP, i)


def make_agg_config(strategy, columns, keys=(), instrs=(), consts=(), aggs=(), pred=(),
                    est_groups=0, num_entries=0, code_widths=None, nullable=()):
    """Build an AggConfig.

    columns: list of (type, width) — width may be None for numeric types
    keys:    list of column indices (GROUP BY order)
    instrs:  list of (op, dst, Operand a, Operand b)
    aggs:    list of (fn, Operand-or-None)
    pred:    list of (column, cmp, literal)
    nullable: column indices whose attribute type is nullable (their null bitmaps go to update_nullable)
    """
    cfg = AggConfig()
    cfg.strategy = strategy
    cfg.num_columns = len(columns)
    for i, (t, w) in enumerate(columns):
        cfg.column_type[i] = t
        cfg.column_width[i] = TYPE_WIDTH[t] if w is None else w
    cfg.num_keys = len(keys)
    for i, k in enumerate(keys):
        cfg.key_column[i] = k
    cfg.num_instrs = len(instrs)
    for i, (op, dst, a, b) in enumerate(instrs):
        cfg.instrs[i] = ExprInstr(op, dst, a, b)
    for i, v in enumerate(consts):
        cfg.consts[i] = v
    cfg.num_aggs = len(aggs)
    for i, (fn, arg) in enumerate(aggs):
        cfg.aggs[i] = AggDesc(fn, arg if arg is not None else Operand(OPD_COLUMN, 0))
    cfg.num_pred_terms = len(pred)
    for i, (column, op, literal) in enumerate(pred):
        term = PredTerm()
        term.column = column
        term.op = op
        t = columns[column][0]
        if t == INT:
            term.literal.i32 = int(literal)
        elif t in (LONG, DATE):
            term.literal.i64 = int(literal)
        elif t == FLOAT:
            term.literal.f32 = float(literal)
        else:
            term.literal.f64 = float(literal)
        cfg.pred[i] = term
    cfg.est_groups = est_groups
    cfg.num_entries = num_entries
    for i, w in enumerate(code_widths or ()):      # 0 = plain column, 1 / 2 / 4 = compressed attribute (codes)
        cfg.column_code_width[i] = w
    for i in nullable:
        cfg.column_nullable[i] = 1
    return cfg


def agg_output_dtype(cfg, a):
    """numpy dtype name of the value column qsx_agg_finalize writes for aggregate `a`: int64 for COUNT
    and SUM over INT/LONG, float64 for the other SUMs and AVG, the argument's own type for MIN/MAX."""
    fn = cfg.aggs[a].fn
    if fn in (AGG_MIN, AGG_MAX):
        arg = cfg.aggs[a].arg
        if arg.kind != OPD_COLUMN:
            return "float64"
        return {INT: "int32", LONG: "int64", FLOAT: "float32", DOUBLE: "float64"}[cfg.column_type[arg.index]]
    return "int64" if agg_output_is_int(cfg, a) else "float64"


def agg_output_is_int(cfg, a):
    """True when aggregate `a` finalizes to int64 (COUNT, SUM over INT/LONG)."""
    fn = cfg.aggs[a].fn
    if fn in (AGG_COUNT_STAR, AGG_COUNT):
        return True
    if fn == AGG_AVG:
        return False
    arg = cfg.aggs[a].arg
    return arg.kind == OPD_COLUMN and cfg.column_type[arg.index] in (INT, LONG)


MAX_PROJECTED = 16                                                  # QSX_MAX_PROJECTED


class JoinProjection(C.Structure):                                   # qsx_join_projection_t
    _fields_ = [("num_columns", C.c_int32), ("width", C.c_int32 * MAX_PROJECTED), ("on_build", C.c_int32 * MAX_PROJECTED),
                ("probe_stripes", C.POINTER(C.c_void_p)), ("num_build_segments", C.c_int32),
                ("build_first_tids", C.POINTER(C.c_int64)), ("build_stripes", C.POINTER(C.c_void_p)),
                ("out_columns", C.POINTER(C.c_void_p))]
