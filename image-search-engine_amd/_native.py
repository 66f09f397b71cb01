"""ctypes binding of the gfx950 kNN library (include/ise_knn.h).

The product path has no CPU fallback: if ``csrc/libise_knn.so`` is missing the
import of this module raises, and if no MI355X is visible every entry point
that computes raises ``RuntimeError`` with the library's message.
"""
from __future__ import annotations

import ctypes
import os

# PyTorch-ROCm bundles its own libamdhip64.so.7; importing torch first makes
# this library bind to the same HIP runtime, so torch device pointers and
# streams can be handed straight through the C ABI.
import torch  # noqa: F401  (must precede the CDLL below)

_HERE = os.path.dirname(os.path.abspath(__file__))
# $ISE_KNN_LIB selects a dev build of the same library (e.g. csrc/libise_knn_ablate.so)
LIB_PATH = os.environ.get("ISE_KNN_LIB") or os.path.join(_HERE, "csrc", "libise_knn.so")

METRIC_INNER_PRODUCT = 0
METRIC_L2 = 1
MAX_K = 2048
STORE_F32, STORE_BF16 = 0, 1

E_INVALID, E_HIP, E_NOMEM, E_NODEVICE = -1, -2, -3, -4

# every symbol include/ise_knn.h declares: (name, restype, argtypes)
_f32p = ctypes.POINTER(ctypes.c_float)
_i64p = ctypes.POINTER(ctypes.c_int64)
_u64p = ctypes.POINTER(ctypes.c_uint64)
_vp = ctypes.c_void_p
_i64 = ctypes.c_int64
_int = ctypes.c_int
PROTOTYPES = [
    ("ise_version", _int, []),
    ("ise_last_error", ctypes.c_char_p, []),
    ("ise_device_count", _int, [ctypes.POINTER(_int)]),
    ("ise_device_arch", _int, [_int, ctypes.c_char_p, _int]),
    ("ise_index_create", _int, [ctypes.POINTER(_vp), _int, _int, _int]),
    ("ise_index_create_ex", _int, [ctypes.POINTER(_vp), _int, _int, _int, _int]),
    ("ise_index_destroy", _int, [_vp]),
    ("ise_index_reset", _int, [_vp]),
    ("ise_index_info", _int, [_vp, ctypes.POINTER(_int), ctypes.POINTER(_int), _i64p, ctypes.POINTER(_int)]),
    ("ise_index_set_shift", _int, [_vp, _vp]),
    ("ise_index_get_shift", _int, [_vp, _vp]),
    ("ise_index_stats", _int, [_vp, _u64p]),
    ("ise_index_host_stats", _int, [_vp, _u64p]),
    ("ise_index_reserve_workspaces", _int, [_vp, _i64, _int]),
    ("ise_index_add_host", _int, [_vp, _vp, _i64]),
    ("ise_index_add_device", _int, [_vp, _vp, _i64, _vp]),
    ("ise_index_reconstruct_host", _int, [_vp, _i64, _i64, _vp]),
    ("ise_index_search_host", _int, [_vp, _vp, _i64, _int, _vp, _vp]),
    ("ise_index_search_device", _int, [_vp, _vp, _i64, _int, _vp, _vp, _vp]),
    ("ise_index_search_keys_device", _int, [_vp, _vp, _i64, _int, ctypes.c_uint32, _vp, _vp]),
    ("ise_index_assign_device", _int, [_vp, _vp, _i64, _vp, _vp, _vp]),
    ("ise_merge_keys_device", _int, [_vp, _int, _i64, _int, _int, _vp, _vp, _int, _vp]),
    ("ise_normalize_rows_device", _int, [_vp, _i64, _int, _int, _vp]),
    ("ise_normalize_rows_host", _int, [_vp, _i64, _int, _int]),
    ("ise_bovw_histogram_device", _int, [_vp, _vp, _i64, _int, _vp, _int, _vp]),
    ("ise_index_short_stats", _int, [_vp, _vp]),
    ("ise_refresh_env_knobs", _int, []),
    ("ise_comm_precheck", _int, [_int]),
    ("ise_comm_unique_id", _int, [_vp]),
    ("ise_comm_create", _int, [ctypes.POINTER(_vp), _vp, _int, _int, _int]),
    ("ise_comm_allgather_keys", _int, [_vp, _vp, _vp, _i64, _vp]),
    ("ise_comm_destroy", _int, [_vp]),
    ("ise_index_search_timed_device", _int,
     [_vp, _vp, _i64, _int, _vp, _vp, _vp, _int, _f32p, _f32p]),
]

if not os.path.exists(LIB_PATH):
    raise ImportError(
        f"{LIB_PATH} is missing: build it with `python __graft_entry__.py` (or "
        f"`make -C {os.path.join(_HERE, 'csrc')}`). There is no CPU fallback for the kNN path."
    )

lib = ctypes.CDLL(LIB_PATH)
for _name, _res, _args in PROTOTYPES:
    _fn = getattr(lib, _name)  # AttributeError here = the .so is stale
    _fn.restype = _res
    _fn.argtypes = _args


class IseError(RuntimeError):
    """Error reported through the C ABI (Faiss surfaces C++ errors as RuntimeError too)."""

    def __init__(self, code: int, msg: str):
        super().__init__(f"ise_knn error {code}: {msg}")
        self.code = code


def check(rc: int) -> None:
    if rc != 0:
        raise IseError(rc, lib.ise_last_error().decode("utf-8", "replace"))


def device_count() -> int:
    n = _int(0)
    rc = lib.ise_device_count(ctypes.byref(n))
    return int(n.value) if rc == 0 else 0
