"""ctypes binding of libtsearch.so (include/tsearch.h).  The only caller of the C ABI.

There is no CPU fallback: if the shared library is missing or no gfx950 device is visible,
every compute call raises.  Build the library with ``python -c "import __graft_entry__ as g; g.build()"``
or ``make -C theoremsearch_amd/csrc``.
"""
from __future__ import annotations

import ctypes as C
import os
import threading

import numpy as np

TS_OK = 0
TS_F32, TS_BF16 = 0, 1
TS_METRIC_IP, TS_METRIC_COS = 0, 1
TS_ALGO_AUTO, TS_ALGO_SCAN, TS_ALGO_MFMA = 0, 1, 2
TS_MAX_K = 256

# TS_LIB overrides the library file (A/B builds of the same ABI during kernel tuning)
_LIB_PATH = os.environ.get("TS_LIB") or os.path.join(os.path.dirname(os.path.abspath(__file__)), "libtsearch.so")
_lib = None
_lock = threading.Lock()


class TSearchError(RuntimeError):
    def __init__(self, code: int, message: str):
        super().__init__(f"libtsearch error {code}: {message}")
        self.code = code


class SearchStats(C.Structure):
    _fields_ = [("algo", C.c_int32), ("levels", C.c_int32), ("fallback_queries", C.c_int32),
                ("reserved", C.c_int32), ("candidates", C.c_int64)]


# name -> (restype, argtypes); kept in one table so tests can check it against the header
_SIGNATURES = {
    "ts_version": (C.c_int, []),
    "ts_last_error": (C.c_char_p, []),
    "ts_device_count": (C.c_int, [C.POINTER(C.c_int)]),
    "ts_device_info": (C.c_int, [C.c_int, C.c_char_p, C.c_int, C.POINTER(C.c_int64), C.POINTER(C.c_int32)]),
    "ts_device_synchronize": (C.c_int, [C.c_int]),
    "ts_index_create": (C.c_int, [C.c_int, C.c_int64, C.c_int32, C.c_int, C.c_int, C.POINTER(C.c_void_p)]),
    "ts_index_destroy": (C.c_int, [C.c_void_p]),
    "ts_index_set_row_offset": (C.c_int, [C.c_void_p, C.c_int64]),
    "ts_index_synchronize": (C.c_int, [C.c_void_p]),
    "ts_index_wait_order": (C.c_int, [C.c_void_p, C.c_void_p]),
    "ts_copy_device": (C.c_int, [C.c_int, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]),
    "ts_search_cpu": (C.c_int, [C.c_void_p, C.c_int, C.c_int64, C.c_int32, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_int32, C.c_int32,
                                C.c_void_p, C.c_void_p, C.c_int32]),
    "ts_index_stream": (C.c_int, [C.c_void_p, C.POINTER(C.c_void_p)]),
    "ts_index_info": (C.c_int, [C.c_void_p, C.POINTER(C.c_int64), C.POINTER(C.c_int32), C.POINTER(C.c_int32),
                                C.POINTER(C.c_int32), C.POINTER(C.c_int64), C.POINTER(C.c_int64),
                                C.POINTER(C.c_void_p)]),
    "ts_index_set_option": (C.c_int, [C.c_void_p, C.c_char_p, C.c_int32]),
    "ts_index_reset_option": (C.c_int, [C.c_void_p, C.c_char_p]),
    "ts_index_upload": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int64, C.c_int64]),
    "ts_index_upload_device": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int64, C.c_int64, C.c_int64, C.c_void_p]),
    "ts_index_reserve": (C.c_int, [C.c_void_p, C.c_int64]),
    "ts_index_append": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int64, C.POINTER(C.c_int64)]),
    "ts_index_append_device": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int64, C.c_int64, C.c_void_p,
                                         C.POINTER(C.c_int64)]),
    "ts_index_attach_device": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64]),
    "ts_index_download": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int64]),
    "ts_search": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p,
                            C.c_int, C.c_void_p]),
    "ts_search_ex": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p,
                               C.c_int, C.c_void_p, C.c_int, C.POINTER(SearchStats)]),
    "ts_parse_pgvector_text": (C.c_int, [C.c_char_p, C.c_int64, C.c_int32, C.c_void_p, C.c_int64, C.POINTER(C.c_int64),
                                         C.POINTER(C.c_int64)]),
    "ts_index_view": (C.c_int, [C.c_void_p, C.POINTER(C.c_void_p)]),
    "ts_index_subset": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.POINTER(C.c_void_p)]),
    "ts_search_filtered": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int32, C.c_int32, C.c_void_p, C.c_int,
                                     C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]),
    "ts_search_filtered_ex": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int32, C.c_int32, C.c_void_p, C.c_int,
                                        C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.POINTER(SearchStats)]),
    "ts_search_biased": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int32, C.c_int32, C.c_void_p, C.c_int, C.c_float,
                                   C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]),
    "ts_rank_of": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p,
                             C.c_void_p]),
    "ts_count_above": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p,
                                 C.c_void_p]),
    "ts_scores": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int32, C.c_void_p, C.c_int, C.c_void_p]),
    "ts_merge_topk": (C.c_int, [C.c_int, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32,
                                C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]),
    "ts_merge_topk_packed": (C.c_int, [C.c_int, C.c_void_p, C.c_int64, C.c_int64, C.c_int32, C.c_int32, C.c_int32, C.c_int32,
                                       C.c_void_p, C.c_void_p, C.c_void_p]),
    "ts_comm_unique_id": (C.c_int, [C.c_void_p, C.c_int32]),
    "ts_comm_create": (C.c_int, [C.c_int, C.c_int32, C.c_int32, C.c_void_p, C.c_int32, C.POINTER(C.c_void_p)]),
    "ts_comm_destroy": (C.c_int, [C.c_void_p]),
    "ts_comm_info": (C.c_int, [C.c_void_p, C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.POINTER(C.c_int32)]),
    "ts_comm_allgather": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]),
    "ts_comm_search": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int32, C.c_int32, C.c_void_p,
                                 C.c_void_p, C.c_int, C.c_void_p]),
    "ts_shards_create": (C.c_int, [C.c_int32, C.c_void_p, C.c_int64, C.c_int32, C.c_int, C.c_int, C.POINTER(C.c_void_p)]),
    "ts_shards_destroy": (C.c_int, [C.c_void_p]),
    "ts_shards_info": (C.c_int, [C.c_void_p, C.POINTER(C.c_int32), C.POINTER(C.c_int64), C.POINTER(C.c_int32)]),
    "ts_shards_shard": (C.c_int, [C.c_void_p, C.c_int32, C.POINTER(C.c_void_p), C.POINTER(C.c_int64), C.POINTER(C.c_int64)]),
    "ts_shards_upload": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int64, C.c_int64]),
    "ts_shards_search": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p]),
    "ts_pool_normalize": (C.c_int, [C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int64, C.c_int32, C.c_int32, C.c_int, C.c_int,
                                    C.c_void_p, C.c_int, C.c_int64, C.c_void_p]),
    "ts_add_layernorm": (C.c_int, [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_float, C.c_int64, C.c_int32, C.c_int,
                                   C.c_void_p, C.c_void_p]),
    "ts_embed_layernorm": (C.c_int, [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_int64,
                                     C.c_void_p, C.c_void_p, C.c_float, C.c_int64, C.c_int32, C.c_int32, C.c_int, C.c_void_p, C.c_void_p]),
    "ts_attention_short": (C.c_int, [C.c_int, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p]),
    "ts_attention_float": (C.c_int, [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int,
                                     C.c_float, C.c_void_p, C.c_void_p, C.c_void_p]),
    "ts_attention_gqa": (C.c_int, [C.c_int, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int,
                                   C.c_void_p, C.c_void_p]),
    "ts_add_rmsnorm": (C.c_int, [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_float, C.c_int64, C.c_int32, C.c_int, C.c_void_p,
                                 C.c_void_p, C.c_void_p]),
    "ts_qk_norm_rope": (C.c_int, [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_float, C.c_int64, C.c_int32,
                                  C.c_int32, C.c_int32, C.c_int32, C.c_int, C.c_void_p]),
    "ts_add_layernorm_pieces": (C.c_int, [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_float, C.c_int64, C.c_int32,
                                          C.c_void_p, C.c_void_p, C.c_void_p]),
    "ts_add_rmsnorm_pieces": (C.c_int, [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_float, C.c_int64, C.c_int32, C.c_void_p,
                                        C.c_void_p, C.c_void_p, C.c_void_p]),
    "ts_gemma_norm_pieces": (C.c_int, [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_float, C.c_int64, C.c_int32,
                                       C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "ts_act_pieces": (C.c_int, [C.c_int, C.c_void_p, C.c_void_p, C.c_int64, C.c_int32, C.c_int, C.c_void_p, C.c_void_p]),
    "ts_split_pieces": (C.c_int, [C.c_int, C.c_void_p, C.c_int64, C.c_int32, C.c_int, C.c_void_p, C.c_void_p]),
    "ts_swiglu": (C.c_int, [C.c_int, C.c_void_p, C.c_int64, C.c_int32, C.c_int, C.c_void_p, C.c_void_p]),
    "ts_geglu": (C.c_int, [C.c_int, C.c_void_p, C.c_int64, C.c_int32, C.c_int, C.c_void_p, C.c_void_p]),
    "ts_gemma_norm": (C.c_int, [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_float, C.c_int64, C.c_int32, C.c_int,
                                C.c_void_p, C.c_void_p, C.c_void_p]),
    "ts_gemma_qk_norm_rope": (C.c_int, [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_float, C.c_int64,
                                        C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int, C.c_void_p]),
    "ts_index_profile_enable": (C.c_int, [C.c_void_p, C.c_int]),
    "ts_index_profile_read": (C.c_int, [C.c_void_p, C.POINTER(C.c_int64), C.POINTER(C.c_double), C.POINTER(C.c_int64)]),
    "ts_index_probe_read": (C.c_int, [C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_double)]),
    "ts_timer_create": (C.c_int, [C.c_int, C.POINTER(C.c_void_p)]),
    "ts_timer_start": (C.c_int, [C.c_void_p, C.c_void_p]),
    "ts_timer_stop": (C.c_int, [C.c_void_p, C.c_void_p]),
    "ts_timer_elapsed_ms": (C.c_int, [C.c_void_p, C.POINTER(C.c_float)]),
    "ts_timer_destroy": (C.c_int, [C.c_void_p]),
}


def lib_path() -> str:
    return _LIB_PATH


def _preload_torch_hip_runtime() -> None:
    """One HIP runtime per process.  The PyTorch-ROCm wheel bundles its own libamdhip64 under torch/lib; if
    libtsearch.so pulls in the system runtime (/opt/rocm) first, a later ``import torch`` binds to that copy and
    its device discovery fails ("No HIP GPUs are available").  So when torch is installed but not imported yet,
    load its bundled runtime first (globally): libtsearch then resolves against it, exactly as it does when the
    application imported torch first."""
    import sys
    if "torch" in sys.modules:
        return
    try:
        import importlib.util
        spec = importlib.util.find_spec("torch")
        if spec is None or not spec.submodule_search_locations:
            return
        libdir = os.path.join(list(spec.submodule_search_locations)[0], "lib")
        for name in ("libamdhip64.so",):
            path = os.path.join(libdir, name)
            if os.path.exists(path):
                C.CDLL(path, mode=C.RTLD_GLOBAL)
    except OSError:
        pass   # fall back to the system runtime


def prefer_torch_rccl() -> None:
    """One RCCL per process: when PyTorch-ROCm is installed, point libtsearch's lazy RCCL binding (dlopen at the first
    ts_comm_* / ts_shards_* call) at the copy torch bundles and may already have mapped, unless TS_RCCL_LIB says otherwise."""
    if os.environ.get("TS_RCCL_LIB"):
        return
    try:
        import importlib.util
        spec = importlib.util.find_spec("torch")
        if spec is not None and spec.submodule_search_locations:
            path = os.path.join(list(spec.submodule_search_locations)[0], "lib", "librccl.so")
            if os.path.exists(path):
                os.environ["TS_RCCL_LIB"] = path
    except Exception:
        pass


def load() -> C.CDLL:
    """Load libtsearch.so (once).  Raises if it has not been built."""
    global _lib
    with _lock:
        if _lib is None:
            _preload_torch_hip_runtime()
            if not os.path.exists(_LIB_PATH):
                raise TSearchError(-4, f"{_LIB_PATH} not found: build it first (make -C theoremsearch_amd/csrc); "
                                       "there is no CPU fallback")
            lib = C.CDLL(_LIB_PATH)
            for name, (res, args) in _SIGNATURES.items():
                fn = getattr(lib, name)
                fn.restype = res
                fn.argtypes = args
            _lib = lib
    return _lib


def check(rc: int) -> None:
    if rc != TS_OK:
        raise TSearchError(rc, load().ts_last_error().decode("utf-8", "replace"))


def device_count() -> int:
    n = C.c_int(0)
    check(load().ts_device_count(C.byref(n)))
    return n.value


def device_info(device: int = 0) -> dict:
    name = C.create_string_buffer(256)
    mem, cus = C.c_int64(0), C.c_int32(0)
    check(load().ts_device_info(device, name, 256, C.byref(mem), C.byref(cus)))
    return {"name": name.value.decode(), "total_mem": mem.value, "compute_units": cus.value}


def np_dtype_code(arr: np.ndarray) -> int:
    if arr.dtype == np.float32:
        return TS_F32
    if arr.dtype == np.uint16:
        return TS_BF16
    raise TypeError(f"expected float32 or uint16 (bf16 bits), got {arr.dtype}")


def as_ptr(arr: np.ndarray) -> C.c_void_p:
    return C.c_void_p(arr.ctypes.data)
