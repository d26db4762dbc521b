"""ctypes loader for libmeepo_hip.so — the C-ABI declared in include/meepo_embedding.h.

There is no CPU fallback: if the HIP extension is missing this module raises, and every operator raises
`MeepoError` if the library reports an error (e.g. no gfx950 device).  torch is imported first on purpose:
the extension's `libamdhip64.so.7` dependency must resolve to the HIP runtime torch already loaded, so that
torch's streams and device pointers are valid inside the library.
"""
from __future__ import annotations

import ctypes as C
import os

import torch  # noqa: F401  (must precede CDLL: see module docstring)

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("MEE_LIB_PATH") or os.path.join(_HERE, "libmeepo_hip.so")  # MEE_LIB_PATH: A/B against another build

OK = 0
ERR_INVALID_ARG, ERR_OUT_OF_MEMORY, ERR_HIP, ERR_NO_DEVICE, ERR_BATCH_TOO_LARGE, ERR_UNSUPPORTED, ERR_RCCL = -1, -2, -3, -4, -5, -6, -7
OPT_NONE, OPT_ADAGRAD, OPT_ADAM = 0, 1, 2
INIT_CONSTANT, INIT_UNIFORM = 0, 1
STATUS_TABLE_FULL, STATUS_RESERVED_KEY, STATUS_STALE_HANDLE, STATUS_INTERNAL = 1, 2, 4, 8
FIND_DEFAULT, FIND_STREAM_STORES, FIND_CACHED_STORES, FIND_STREAM_ROWS, FIND_STREAM_BUCKETS = 0, 1, 2, 4, 8   # mee_find_ex flags
HANDLE_SLOT_MASK = (1 << 40) - 1   # a located-find handle: bits 0..39 the slot (as mee_locate reports it), bits 40..61 the table's layout epoch
MEM_HBM, MEM_HOST_PINNED = 0, 1
FLAG_TRACK_HITS, FLAG_ADMISSION = 1, 2
ABI_VERSION = 2   # MEE_ABI_VERSION of include/meepo_embedding.h this loader was written against
EMPTY_KEY = -(1 << 63)
RECLAIMED_KEY = EMPTY_KEY + 1
BUCKET_WIDTH = 16


class MeepoError(RuntimeError):
    def __init__(self, code: int, msg: str):
        super().__init__(f"meepo error {code}: {msg}")
        self.code = code


class Config(C.Structure):
    _fields_ = [
        ("struct_size", C.c_uint32), ("device", C.c_int32), ("capacity", C.c_uint64), ("dim", C.c_uint32),
        ("optimizer", C.c_uint32), ("max_batch", C.c_uint64), ("default_value", C.c_float),
        ("initial_accumulator", C.c_float), ("initializer", C.c_uint32), ("init_scale", C.c_float),
        ("init_seed", C.c_uint64), ("value_memory", C.c_uint32), ("flags", C.c_uint32),
    ]


class FindRequest(C.Structure):
    _fields_ = [("d_keys", C.c_void_p), ("n", C.c_size_t), ("d_out", C.c_void_p), ("d_found", C.c_void_p)]


class Calibration(C.Structure):
    """mee_calibration: what mee_device_calibration measured (the apply's bucket-pair split by block-index parity)"""
    _fields_ = [("struct_size", C.c_uint32), ("xcd_split", C.c_uint32), ("from_env", C.c_uint32), ("placement_consistent", C.c_uint32),
                ("odd_over_even", C.c_float * 2), ("block_us_by_index_mod_8", C.c_float * 8), ("xcc_of_index_mod_8", C.c_uint32 * 8)]


class ShardedOptions(C.Structure):
    _fields_ = [("struct_size", C.c_uint32), ("flags", C.c_uint32), ("max_batch", C.c_uint64), ("pad_slack", C.c_double),
                ("cold", C.c_void_p), ("hot_key_limit", C.c_uint64)]


SHARDED_DEDUP = 1


class TableInfo(C.Structure):
    _fields_ = [
        ("capacity", C.c_uint64), ("n_buckets", C.c_uint64), ("max_batch", C.c_uint64), ("dim", C.c_uint32),
        ("optimizer", C.c_uint32), ("table_bytes", C.c_uint64), ("workspace_bytes", C.c_uint64),
    ]


_vp, _sz, _u64, _u32, _i32, _f32 = C.c_void_p, C.c_size_t, C.c_uint64, C.c_uint32, C.c_int32, C.c_float

# name -> (restype, argtypes); this is the full export list of include/meepo_embedding.h
PROTOTYPES = {
    "mee_abi_version": (C.c_int, []),
    "mee_last_error": (C.c_char_p, []),
    "mee_table_create": (C.c_int, [C.POINTER(Config), C.POINTER(_vp)]),
    "mee_table_destroy": (C.c_int, [_vp]),
    "mee_table_info_get": (C.c_int, [_vp, C.POINTER(TableInfo)]),
    "mee_clear": (C.c_int, [_vp, _vp]),
    "mee_set_tuning": (C.c_int, [_vp, C.c_char_p, C.c_int]),
    "mee_find": (C.c_int, [_vp, _vp, _sz, _vp, _vp, _vp]),
    "mee_find_ex": (C.c_int, [_vp, _vp, _sz, _vp, _vp, C.c_uint32, _vp]),
    "mee_find_missing": (C.c_int, [_vp, _vp, _sz, _vp, _vp, _vp]),
    "mee_find_counted": (C.c_int, [_vp, _vp, _sz, _vp, _vp, C.c_int, _vp]),
    "mee_hits_scan": (C.c_int, [_vp, _u32, _u32, C.c_int, _vp, _sz, C.POINTER(_sz), _vp]),
    "mee_insert": (C.c_int, [_vp, _vp, _vp, _sz, _vp]),
    "mee_insert_missing": (C.c_int, [_vp, _vp, _vp, _sz, _vp, _vp]),
    "mee_find_or_insert_missing": (C.c_int, [_vp, _vp, _sz, _vp, _vp, _vp]),
    "mee_assign": (C.c_int, [_vp, _vp, _vp, _sz, _vp, _vp]),
    "mee_find_plane": (C.c_int, [_vp, _u32, _vp, _sz, _vp, _vp, _vp]),
    "mee_assign_plane": (C.c_int, [_vp, _u32, _vp, _vp, _sz, _vp, _vp]),
    "mee_remove": (C.c_int, [_vp, _vp, _sz, _vp, _vp]),
    "mee_find_or_insert": (C.c_int, [_vp, _vp, _sz, _vp, _vp, _vp]),
    "mee_export": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _sz, C.POINTER(_sz), _vp]),
    "mee_group_create": (C.c_int, [C.POINTER(_vp), _u32, _u64, C.POINTER(_vp)]),
    "mee_group_find_pooled": (C.c_int, [_vp, _vp, _sz, _vp, _sz, _vp, _vp, _vp, C.c_int, _vp]),
    "mee_group_apply_adagrad_pooled": (C.c_int, [_vp, _vp, _vp, _sz, _vp, _vp, _vp, _sz, _f32, _f32, _vp]),
    "mee_group_apply_adam_pooled": (C.c_int, [_vp, _vp, _vp, _sz, _vp, _vp, _vp, _sz, _f32, _f32, _f32, _f32, _u64, _vp]),
    "mee_group_find_or_insert": (C.c_int, [_vp, _vp, _vp, _sz, _vp, _vp, _vp]),
    "mee_group_apply_adagrad": (C.c_int, [_vp, _vp, _vp, _vp, _sz, _f32, _f32, _vp]),
    "mee_group_apply_adam": (C.c_int, [_vp, _vp, _vp, _vp, _sz, _f32, _f32, _f32, _f32, _u64, _vp]),
    "mee_group_destroy": (C.c_int, [_vp]),
    "mee_group_set_tuning": (C.c_int, [_vp, C.c_char_p, C.c_int]),
    "mee_find_grouped": (C.c_int, [_vp, _vp, _vp, _sz, _vp, _vp, _vp]),
    "mee_find_pooled": (C.c_int, [_vp, _vp, _sz, _vp, _sz, _vp, _vp, C.c_int, _vp]),
    "mee_apply_adagrad_indexed": (C.c_int, [_vp, _vp, _vp, _sz, _vp, _sz, _f32, _f32, _vp]),
    "mee_apply_adam_indexed": (C.c_int, [_vp, _vp, _vp, _sz, _vp, _sz, _f32, _f32, _f32, _f32, _u64, _vp]),
    "mee_dedup_keys": (C.c_int, [_vp, _vp, _sz, _vp, _vp, C.c_int64, _vp]),
    "mee_reserve": (C.c_int, [_vp, _u64, _vp]),
    "mee_export_range": (C.c_int, [_vp, _u64, _u64, _vp, _vp, _vp, _vp, _sz, C.POINTER(_sz), _vp]),
    "mee_size": (C.c_int, [_vp, C.POINTER(_sz), _vp]),
    "mee_status": (C.c_int, [_vp, C.POINTER(_u32), _vp]),
    "mee_clear_status": (C.c_int, [_vp, _vp]),
    "mee_find_unordered": (C.c_int, [_vp, _vp, _sz, _vp, _vp, _vp]),
    "mee_find_many": (C.c_int, [_vp, C.POINTER(FindRequest), _u32, _vp]),
    "mee_find_located": (C.c_int, [_vp, _vp, _sz, _vp, _vp, _vp, _vp]),
    "mee_find_or_insert_located": (C.c_int, [_vp, _vp, _sz, _vp, _vp, _vp, _vp]),
    "mee_find_located_prepare": (C.c_int, [_vp, _vp, _sz, _vp, _vp, _vp, _vp]),
    "mee_find_or_insert_located_prepare": (C.c_int, [_vp, _vp, _sz, _vp, _vp, _vp, _vp]),
    "mee_apply_adagrad_located": (C.c_int, [_vp, _vp, _vp, _vp, _sz, _f32, _f32, _vp]),
    "mee_apply_adam_located": (C.c_int, [_vp, _vp, _vp, _vp, _sz, _f32, _f32, _f32, _f32, _u64, _vp]),
    "mee_find_or_insert_admit": (C.c_int, [_vp, _vp, _sz, _vp, _vp, _u32, _vp]),
    "mee_admission_decay": (C.c_int, [_vp, _u32, _vp]),
    "mee_locate": (C.c_int, [_vp, _vp, _sz, _vp, _vp, _vp]),
    "mee_table_plane": (C.c_int, [_vp, _u32, C.POINTER(_vp), C.POINTER(_u64), C.POINTER(_u32)]),
    "mee_device_calibration": (C.c_int, [C.c_int32, C.POINTER(Calibration)]),
    "mee_probe_length": (C.c_int, [_vp, _vp, _sz, C.POINTER(_u64), _vp]),
    "mee_probe_histogram": (C.c_int, [_vp, _vp, _sz, C.POINTER(C.c_uint64), _vp]),
    "mee_apply_adagrad": (C.c_int, [_vp, _vp, _vp, _sz, _f32, _f32, _vp]),
    "mee_apply_adam": (C.c_int, [_vp, _vp, _vp, _sz, _f32, _f32, _f32, _f32, _u64, _vp]),
    "mee_apply_prepare": (C.c_int, [_vp, _vp, _sz, _vp]),
    "mee_apply_discard": (C.c_int, [_vp, _vp]),
    "mee_dedup_sum": (C.c_int, [_vp, _vp, _vp, _sz, _vp, _vp, _vp, _vp, C.c_int64, _vp]),
    "mee_hash_batch": (C.c_int, [_vp, _sz, _u64, _u32, _vp, _vp, _vp, _vp]),
    "mee_router_create": (C.c_int, [_i32, _u64, _u32, C.POINTER(_vp)]),
    "mee_router_destroy": (C.c_int, [_vp]),
    "mee_partition_padded": (C.c_int, [_vp, _vp, _sz, _vp, _vp, _vp, _vp]),
    "mee_partition": (C.c_int, [_vp, _vp, _sz, _vp, _vp, _vp, _vp]),
    "mee_p2p_create": (C.c_int, [_i32, _u32, _u32, _u64, _u64, _u32, C.c_int, C.POINTER(_vp)]),
    "mee_p2p_barrier": (C.c_int, [_vp, _vp]),
    "mee_p2p_push_rows": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _sz, _vp]),
    "mee_p2p_inbox": (C.c_int, [_vp, C.POINTER(_vp), C.POINTER(_vp), C.POINTER(_u64)]),
    "mee_p2p_destroy": (C.c_int, [_vp]),
    "mee_p2p_export": (C.c_int, [_vp, _vp]),
    "mee_p2p_connect": (C.c_int, [_vp, _vp]),
    "mee_p2p_buffers": (C.c_int, [_vp, C.POINTER(_vp), C.POINTER(_vp)]),
    "mee_p2p_push": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _sz, _vp]),
    "mee_p2p_find": (C.c_int, [_vp, _vp, _vp]),
    "mee_p2p_status": (C.c_int, [_vp, C.POINTER(_u32), _vp]),
    "mee_scatter_rows": (C.c_int, [_vp, _vp, _sz, _sz, _vp, _vp]),
    "mee_gather_rows": (C.c_int, [_vp, _vp, _sz, _sz, _vp, _vp]),
    # row-sharded table over RCCL (meepo_sharded.hip)
    "mee_comm_unique_id": (C.c_int, [_vp]),
    "mee_comm_create": (C.c_int, [_vp, _u32, _u32, _i32, C.POINTER(_vp)]),
    "mee_comm_destroy": (C.c_int, [_vp]),
    "mee_comm_aborted": (C.c_int, [_vp]),
    "mee_sharded_create": (C.c_int, [_vp, _vp, _u64, C.c_double, C.POINTER(_vp)]),
    "mee_sharded_create_ex": (C.c_int, [_vp, _vp, C.POINTER(ShardedOptions), C.POINTER(_vp)]),
    "mee_sharded_clear_status": (C.c_int, [_vp, _vp]),
    "mee_sharded_destroy": (C.c_int, [_vp]),
    "mee_sharded_info": (C.c_int, [_vp, C.POINTER(_u32), C.POINTER(_u32), C.POINTER(_u64)]),
    "mee_sharded_find": (C.c_int, [_vp, _vp, _sz, _vp, _vp, _vp]),
    "mee_sharded_find_or_insert": (C.c_int, [_vp, _vp, _sz, _vp, _vp, _vp]),
    "mee_sharded_insert": (C.c_int, [_vp, _vp, _vp, _sz, _vp]),
    "mee_sharded_assign": (C.c_int, [_vp, _vp, _vp, _sz, _vp, _vp]),
    "mee_sharded_remove": (C.c_int, [_vp, _vp, _sz, _vp, _vp]),
    "mee_sharded_apply_adagrad": (C.c_int, [_vp, _vp, _vp, _sz, _f32, _f32, _vp]),
    "mee_sharded_apply_adam": (C.c_int, [_vp, _vp, _vp, _sz, _f32, _f32, _f32, _f32, _u64, _vp]),
    "mee_sharded_size": (C.c_int, [_vp, C.POINTER(_sz), _vp]),
    "mee_sharded_status": (C.c_int, [_vp, C.POINTER(_u32), _vp]),
}

_lib = None


def lib() -> C.CDLL:
    """Load the HIP extension (once).  Raises ImportError loudly when it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(hipcc --offload-arch=gfx950). meepoembedding_amd has no CPU fallback.")
        L = C.CDLL(LIB_PATH)
        older = os.environ.get("MEE_LIB_OLDER_BUILD") == "1"   # A/B runs of tools/ against an earlier round's library (MEE_LIB_PATH): entry points it lacks stay unbound
        for name, (res, args) in PROTOTYPES.items():
            if older and not hasattr(L, name):
                continue
            fn = getattr(L, name)  # AttributeError here = header/library mismatch
            fn.restype = res
            fn.argtypes = args
        if L.mee_abi_version() != ABI_VERSION:  # noqa: PLR2004
            raise ImportError(f"{LIB_PATH}: ABI version {L.mee_abi_version()} != {ABI_VERSION}")
        _lib = L
    return _lib


def device_calibration(device: int = 0) -> dict:
    """mee_device_calibration as a dict (runs the probe if no table with an optimizer has been created on the device yet)"""
    c = Calibration(struct_size=C.sizeof(Calibration))
    check(lib().mee_device_calibration(int(device), C.byref(c)))
    return {"xcd_split": c.xcd_split, "from_env": bool(c.from_env), "placement_consistent": bool(c.placement_consistent),
            "odd_over_even": [round(float(x), 4) for x in c.odd_over_even], "block_us_by_index_mod_8": [round(float(x), 2) for x in c.block_us_by_index_mod_8],
            "xcc_of_index_mod_8": [int(x) for x in c.xcc_of_index_mod_8]}


def check(rc: int) -> None:
    if rc != OK:
        raise MeepoError(rc, lib().mee_last_error().decode("utf-8", "replace"))
