"""
ctypes binding of libhiprag.so (include/hiprag.h).  No fallbacks: if the library or a symbol is missing this
module raises -- the product path never computes on the CPU.
"""
from __future__ import annotations

import ctypes
import os
from ctypes import POINTER, c_char_p, c_double, c_float, c_int32, c_int64, c_uint32, c_uint64, c_void_p

_PKG_DIR = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB_PATH = os.environ.get("HIPRAG_LIB", os.path.join(_PKG_DIR, "lib", "libhiprag.so"))

METRIC_IP = 0
METRIC_L2 = 1


class HipRagError(RuntimeError):
    """Raised for any non-zero status from libhiprag (message from hiprag_last_error)."""

    def __init__(self, fn: str, code: int, msg: str):
        super().__init__(f"{fn} failed (status {code}): {msg}")
        self.code = code


class HipIdxStats(ctypes.Structure):
    _fields_ = [("passes", c_int64), ("queries", c_int64), ("fallback_queries", c_int64),
                ("bytes_per_pass", c_int64), ("timed_passes", c_int64), ("avg_scan_ms", c_float),
                ("avg_scan_wall_ms", c_float), ("avg_scan_gap_ms", c_float), ("launches", c_int64), ("roundb_queries", c_int64),
                ("list_entries", c_int64), ("ranked_entries", c_int64), ("rescored_groups", c_int64)]


class HipBm25Stats(ctypes.Structure):
    _fields_ = [("queries", c_int64), ("postings_touched", c_int64), ("bytes_algorithmic", c_int64)]


f32p, f64p, i64p, i32p, u32p, u64p = (POINTER(c_float), POINTER(c_double), POINTER(c_int64), POINTER(c_int32),
                                      POINTER(c_uint32), POINTER(c_uint64))

# name -> argtypes; every function returns int32 status except the two noted below
SIGNATURES = {
    "hiprag_device_count": [i32p],
    "hiprag_device_sync": [c_int32],
    "hiprag_scan_stream": [c_int32, c_void_p],
    "hiprag_tail_stream": [c_int32, c_int32, c_void_p],
    "hiprag_init": [c_int32],
    "hiprag_shutdown": [],
    "hiphybrid_search": [c_uint64, c_uint64, c_void_p, c_void_p, c_void_p, c_int32, c_int32, c_int32, c_float, c_float,
                         c_float, c_void_p, c_void_p],
    "hiphybrid_search_dev": [c_uint64, c_uint64, c_void_p, c_void_p, c_void_p, c_int32, c_int32, c_int32, c_float, c_float,
                             c_float, c_void_p, c_void_p, c_void_p, c_void_p],
    "hiphybrid_shard_begin_dev": [c_uint64, c_uint64, c_void_p, c_void_p, c_void_p, c_int32, c_int32, c_int32, c_void_p, c_void_p,
                                  c_void_p, c_void_p],
    "hiphybrid_shard_end_dev": [c_void_p, c_int32, c_int32, c_int32, c_int32, c_int32, c_float, c_float, c_float, c_void_p,
                                c_void_p, c_void_p, c_void_p],
    "hiprag_event_create": [u64p],
    "hiprag_event_record": [c_uint64, c_void_p],
    "hiprag_probe_read_gbps": [c_int32, c_int64, c_int32, POINTER(c_double)],
    "hiprag_event_elapsed_ms": [c_uint64, c_uint64, f32p],
    "hiprag_event_destroy": [c_uint64],
    "hipidx_create": [c_int32, c_int32, c_int32, u64p],
    "hipidx_destroy": [c_uint64],
    "hipidx_add": [c_uint64, c_void_p, c_int64],
    "hipidx_add_dev": [c_uint64, c_void_p, c_int64, c_void_p],
    "hipidx_ntotal": [c_uint64, i64p],
    "hipidx_dim": [c_uint64, i32p],
    "hipidx_metric": [c_uint64, i32p],
    "hipidx_set_id_base": [c_uint64, c_int64],
    "hipidx_search": [c_uint64, c_void_p, c_int32, c_int32, c_void_p, c_void_p],
    "hipidx_search_dev": [c_uint64, c_void_p, c_int32, c_int32, c_void_p, c_void_p, c_void_p, c_void_p],
    "hipidx_search_begin_dev": [c_uint64, c_void_p, c_int32, c_int32, c_int32, c_void_p],
    "hipidx_search_finish_dev": [c_uint64, c_void_p, c_int32, c_int32, c_int32, c_void_p, c_void_p, c_void_p, c_void_p],
    "hipidx_pass_queries": [c_uint64, i32p],
    "hipidx_launch_queries": [c_uint64, i32p],
    "hipidx_set_spare_cus": [c_uint64, c_int32],
    "hipidx_get_spare_cus": [c_uint64, c_void_p],
    "hipidx_gate_tail_dev": [c_uint64, c_int32, c_void_p],
    "hipidx_reserve_search": [c_uint64, c_int32],
    "hipidx_reserve_rows": [c_uint64, c_int64],
    "hipidx_reconstruct": [c_uint64, c_int64, c_void_p],
    "hipidx_save": [c_uint64, c_char_p],
    "hipidx_load": [c_char_p, c_int32, u64p],
    "hipidx_get_stats": [c_uint64, POINTER(HipIdxStats)],
    "hipidx_enable_timing": [c_uint64, c_int32],
    "hipivf_create": [c_uint64, c_uint64, c_void_p, c_void_p, c_int32, u64p],
    "hipivf_destroy": [c_uint64],
    "hipivf_search_dev": [c_uint64, c_void_p, c_int32, c_int32, c_int32, c_void_p, c_void_p, c_void_p, c_void_p],
    "hipivf_info": [c_uint64, i32p, i64p, i64p],
    "hiprag_merge_topk_dev": [c_void_p, c_void_p, c_int32, c_int32, c_int32, c_int32, c_int64, c_int32, c_void_p,
                              c_void_p, c_void_p, c_void_p],
    "hipbm25_create": [c_int64, c_int64, c_void_p, c_void_p, c_void_p, c_int32, u64p],
    "hipbm25_destroy": [c_uint64],
    "hipbm25_set_id_base": [c_uint64, c_int64],
    "hipbm25_search": [c_uint64, c_void_p, c_void_p, c_int32, c_int32, c_void_p, c_void_p],
    "hipbm25_search_dev": [c_uint64, c_void_p, c_void_p, c_int32, c_int32, c_void_p, c_void_p, c_void_p, c_void_p],
    "hipbm25_get_stats": [c_uint64, POINTER(HipBm25Stats)],
    "hipenc_create": [c_void_p, c_void_p, c_int32, u64p],
    "hipenc_destroy": [c_uint64],
    "hipenc_forward": [c_uint64, c_void_p, c_void_p, c_int32, c_int32, c_void_p, c_void_p],
    "hipenc_score_pairs": [c_uint64, c_void_p, c_void_p, c_int32, c_int32, c_void_p, c_void_p],
    "hipenc_last_flops": [c_uint64, f64p],
    "hipenc_linear": [c_void_p, c_void_p, c_void_p, c_int32, c_int32, c_int32, c_int32, c_void_p, c_void_p, c_void_p,
                      c_void_p, c_int32, c_int32, c_int32, c_void_p],
    "hiprrf_fuse": [c_void_p, c_void_p, c_int32, c_int32, c_int32, c_int32, c_float, c_float, c_float, c_void_p,
                    c_void_p],
    "hiprrf_fuse_dev": [c_void_p, c_void_p, c_int32, c_int32, c_int32, c_int32, c_float, c_float, c_float, c_void_p,
                        c_void_p, c_void_p],
}
NON_STATUS = {"hiprag_version": ([], c_int32), "hiprag_last_error": ([], c_char_p)}

_lib = None


def load() -> ctypes.CDLL:
    """Load libhiprag.so and type every symbol the header declares.  Raises if anything is missing."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise HipRagError("load", -1, f"{LIB_PATH} not found: build it with `make -C intool-rag_amd` "
                                      "(__graft_entry__.build()); there is no CPU fallback")
    lib = ctypes.CDLL(LIB_PATH)
    for name, (args, res) in NON_STATUS.items():
        fn = getattr(lib, name)
        fn.argtypes, fn.restype = args, res
    for name, args in SIGNATURES.items():
        fn = getattr(lib, name)        # AttributeError here = header/library mismatch: fail loudly
        fn.argtypes, fn.restype = args, c_int32
    _lib = lib
    return lib


def call(name: str, *args) -> None:
    lib = load()
    rc = getattr(lib, name)(*args)
    if rc != 0:
        msg = lib.hiprag_last_error()
        raise HipRagError(name, rc, msg.decode("utf-8", "replace") if msg else "")


def exported_symbols():
    return list(NON_STATUS) + list(SIGNATURES)
