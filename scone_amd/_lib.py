"""ctypes binding of libscone_hip.so (include/scone_hip.h).

The library is the product path: there is no CPU fallback.  Importing this module
is cheap; :func:`lib` loads the shared object on first use and raises
``RuntimeError`` if it has not been built (``python -c "import __graft_entry__ as g; g.build()"``
or ``make -C scone_amd/csrc -j8``).
"""

import ctypes as C
import os
import threading

_HERE = os.path.dirname(os.path.abspath(__file__))
# SCONE_HIP_LIB: alternative build of the same library (A/B kernel experiments)
LIB_PATH = os.environ.get("SCONE_HIP_LIB") or os.path.join(_HERE, "csrc", "libscone_hip.so")

ABI_VERSION = 2

FMT_F32, FMT_F16, FMT_I8, FMT_I4 = 0, 1, 2, 3
PLACE_HBM, PLACE_PINNED_HOST = 0, 1
REDUCE_MEAN, REDUCE_SUM = 0, 1
MODE_COVER, MODE_LONGEST_SUFFIX = 0, 1
DT_F32, DT_F16, DT_BF16 = 0, 1, 2

OK, ESTATE, EHIP, ENOMEM, ENODEV, EINVAL, ERANGE = 0, -1, -5, -12, -19, -22, -34

ST_BAD_TOKEN, ST_BAD_ID, ST_INDEX_FULL, ST_STAGE_OVERFLOW = 1, 2, 4, 8


class SconeCfg(C.Structure):
    _fields_ = [
        ("struct_size", C.c_uint32),
        ("device", C.c_int32),
        ("max_n", C.c_int32),
        ("dim", C.c_int32),
        ("table_fmt", C.c_int32),
        ("placement", C.c_int32),
        ("n_rows", C.c_uint64),
        ("row_begin", C.c_uint64),
        ("row_end", C.c_uint64),
        ("index_capacity", C.c_uint64),
        ("hot_rows", C.c_uint64),
        ("lookup_mode", C.c_uint32),
        ("stage_tokens", C.c_uint32),
        ("cache_rows", C.c_uint64),
    ]


_P = C.c_void_p
_I32, _I64, _U32, _U64 = C.c_int32, C.c_int64, C.c_uint32, C.c_uint64

# name -> (restype, argtypes); must list every symbol include/scone_hip.h declares
SIGNATURES = {
    "scone_abi_version": (C.c_int, []),
    "scone_strerror": (C.c_char_p, [C.c_int]),
    "scone_create": (C.c_int, [C.POINTER(SconeCfg), C.POINTER(_P)]),
    "scone_destroy": (None, [_P]),
    "scone_last_error": (C.c_char_p, [_P]),
    "scone_status": (C.c_int, [_P, C.POINTER(_U32), _P]),
    "scone_stage_counters": (C.c_int, [_P, C.POINTER(_U64), C.POINTER(_U64), C.POINTER(_U64), C.POINTER(_U64)]),
    "scone_index_build": (C.c_int, [_P, _P, _P, _U64, _U64]),
    "scone_index_build_device": (C.c_int, [_P, _P, _P, _U64, _U64, _P]),
    "scone_index_stats": (C.c_int, [_P, C.POINTER(_U64), C.POINTER(_U64), C.POINTER(_U64)]),
    "scone_index_blob_sizes": (C.c_int, [_P, C.POINTER(_U64), C.POINTER(_U64), C.POINTER(_U64)]),
    "scone_index_export": (C.c_int, [_P, _P, _P, _P, C.POINTER(_U64)]),
    "scone_index_import": (C.c_int, [_P, _P, _P, _P, _U64]),
    "scone_fit": (C.c_int, [_I32, _P, _I64, _P, _I64, _I32, _U32, _U64, _P, _P, _P, _U64, C.POINTER(_U64),
                            C.POINTER(_U64), _P]),
    "scone_table_upload": (C.c_int, [_P, _P, _P, _U64, _U64, C.c_int, _P]),
    "scone_table_download": (C.c_int, [_P, _P, _P, _U64, _U64, C.c_int, _P]),
    "scone_table_store_f32": (C.c_int, [_P, _P, _U64, _U64, _P]),
    "scone_table_store_f32_ids": (C.c_int, [_P, _P, _P, _U64, _P]),
    "scone_table_fill_synthetic": (C.c_int, [_P, _U32, C.c_float, _P]),
    "scone_table_gather_rows": (C.c_int, [_P, _P, _U64, _P, _P]),
    "scone_match": (C.c_int, [_P, _P, _I32, _I32, _P, _P]),
    "scone_match_csr": (C.c_int, [_P, _P, _I32, _I32, _P, _P, _I64, C.POINTER(_I64), _P]),
    "scone_gather_reduce": (C.c_int, [_P, _P, _P, _I64, _P, _I32, _P, _I32, _P]),
    "scone_embed": (C.c_int, [_P, _P, _I32, _I32, _P, _I64, _P, _I64, _P, _I32, _P, _I32, _P]),
    "scone_embed_prefetch": (C.c_int, [_P, _P, _I32, _I32, _I32, _P]),
    "scone_reserve": (C.c_int, [_P, _I64]),
    "scone_set_cu_reserve": (C.c_int, [_P, _I32]),
    "scone_get_cu_reserve": (C.c_int, [_P, C.POINTER(_I32), C.POINTER(_I32)]),
    "scone_lookup_stream": (C.c_int, [_P, C.POINTER(_P)]),
    "scone_streams_overlap": (C.c_int, [_P, _P, _P, C.POINTER(_I32)]),
    "scone_profile_enable": (C.c_int, [_P, C.c_int]),
    "scone_profile_read": (C.c_int, [_P, C.POINTER(_U64), C.POINTER(C.c_double), C.c_int]),
    "scone_profile_samples": (C.c_int, [_P, C.POINTER(C.c_float), _U64, C.POINTER(_U64)]),
    "scone_embed_partial": (C.c_int, [_P, _P, _I32, _I32, _P, _P, _P]),
    "scone_shard_gather_plan": (C.c_int, [_P, _P, _I32, _I32, C.POINTER(_U64), _P]),
    "scone_shard_gather_pack": (C.c_int, [_P, _P, _P]),
    "scone_shard_gather_embed": (C.c_int, [_P, _P, _I32, _I32, _P, _U64, _P, _I64, _P, _I64, _P, _I32, _P, _I32, _P]),
    "scone_shard_select_slot": (C.c_int, [_P, _I32]),
    "scone_shard_gather_plan_chunks": (C.c_int, [_P, _P, _I32, _I32, _I32, _I32, C.POINTER(_U64), _P]),
    "scone_ell_width": (C.c_int, [_P, C.POINTER(_U32)]),
    "scone_shard_gather_match": (C.c_int, [_P, _P, _I32, _I32, _I32, _I32, _P, _P]),
    "scone_shard_gather_plan_ell": (C.c_int, [_P, _P, _I32, _I32, _I32, _I32, C.POINTER(_U64), _P]),
    "scone_shard_gather_pack_range": (C.c_int, [_P, _U64, _U64, _U64, _P, _P]),
    "scone_shard_gather_add_records": (C.c_int, [_P, _P, _U64, _U64, _U64, _P]),
    "scone_shard_gather_remap_range": (C.c_int, [_P, _I32, _I32, _P]),
    "scone_shard_gather_embed_range": (C.c_int, [_P, _P, _I32, _I32, _I32, _I32, _P, _U64, _P, _I64, _P, _I64, _P, _I32, _P,
                                                 _I64, _I32, _P]),
    "scone_shard_gather_plan_async": (C.c_int, [_P, _P, _I32, _I32, _P]),
    "scone_shard_gather_plan_ell_async": (C.c_int, [_P, _P, _I32, _I32, _P]),
    "scone_shard_cols_pack_cap": (C.c_int, [_P, _U64, _P, _P, _P, _U64, _P, _P]),
    "scone_shard_cols_frag_slots": (C.c_int, [_U64, C.POINTER(_U64)]),
    "scone_shard_cols_pack": (C.c_int, [_P, _U64, _U64, _P, _P, _P, _U64, _P]),
    "scone_shard_cols_build_frag": (C.c_int, [_P, _P, _U64, _P, _U64, _P]),
    "scone_shard_head_scales": (C.c_int, [_P, _P, _P]),
    "scone_shard_head_version": (C.c_int, [_P, C.POINTER(_U64)]),
    "scone_shard_cols_embed": (C.c_int, [_P, _P, _I32, _I32, _I32, _I32, _P, _U64, _P, _P, _U64, C.POINTER(_U64), C.POINTER(_U64),
                                         C.POINTER(_U64), C.POINTER(_U64), _I32, _P, _I64, _P, _I64, _P, _I32, _P, _I64, _I32, _P]),
    "scone_shard_set_head": (C.c_int, [_P, _U64]),
    "scone_shard_head_store_f32": (C.c_int, [_P, _P, _U64, _U64, _P]),
    "scone_shard_record_bytes": (C.c_int, [_P, C.POINTER(_U64)]),
    "scone_ipc_alloc": (C.c_int, [_P, _U64, C.POINTER(_P), _P]),
    "scone_ipc_free": (C.c_int, [_P, _P]),
    "scone_ipc_open": (C.c_int, [_P, _P, C.POINTER(_P)]),
    "scone_ipc_close": (C.c_int, [_P, _P]),
    "scone_ipc_event_create": (C.c_int, [_P, C.POINTER(_P), _P]),
    "scone_ipc_event_open": (C.c_int, [_P, _P, C.POINTER(_P)]),
    "scone_ipc_event_destroy": (C.c_int, [_P, _P]),
    "scone_ipc_event_record": (C.c_int, [_P, _P, _P]),
    "scone_ipc_event_wait": (C.c_int, [_P, _P, _P]),
    "scone_ipc_push": (C.c_int, [_P, _P, _P, _U64, _I32, _P]),
    "scone_finalize": (C.c_int, [_P, _P, _P, _P, _I32, _I32, _I64, _I64, _P, _I64, _P, _I64, _P, _I32, _P,
                                 _I32, _P]),
}

_lock = threading.Lock()
_lib = None


def lib() -> C.CDLL:
    """Load (once) and return the C-ABI library; fail loudly if it is missing."""
    global _lib
    if _lib is not None:
        return _lib
    with _lock:
        if _lib is not None:
            return _lib
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"scone_amd: HIP extension not built ({LIB_PATH} is missing). "
                "Build it with `make -C scone_amd/csrc -j8` or __graft_entry__.build(); "
                "there is no CPU fallback for the lookup path.")
        try:
            handle = C.CDLL(LIB_PATH)
        except OSError as e:
            raise RuntimeError(f"scone_amd: cannot load {LIB_PATH}: {e}") from e
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(handle, name)  # AttributeError if the symbol is not exported
            fn.restype = res
            fn.argtypes = args
        if handle.scone_abi_version() != ABI_VERSION:
            raise RuntimeError("scone_amd: libscone_hip.so ABI version mismatch; rebuild it")
        _lib = handle
        return _lib
