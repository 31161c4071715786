"""ctypes binding of libtcgnn_hip.so (the C ABI declared in include/tcgnn.h).

There is no CPU fallback: if the shared library has not been built the import fails loudly, and a
non-zero status from any entry point raises RuntimeError carrying tcgnn_last_error().
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("TCGNN_LIB_PATH") or os.path.join(_HERE, "lib", "libtcgnn_hip.so")  # override: A/B builds

if not os.path.exists(LIB_PATH):
    raise ImportError(
        "libtcgnn_hip.so is not built (%s). Build it with `make -C %s` or "
        "`python -c 'import __graft_entry__ as g; g.build()'`; there is no fallback path."
        % (LIB_PATH, os.path.join(_HERE, "csrc")))

lib = ctypes.CDLL(LIB_PATH)

_i32p = ctypes.c_void_p  # raw addresses (host or device) travel as integers
_vp = ctypes.c_void_p
_i32, _i64, _sz = ctypes.c_int32, ctypes.c_int64, ctypes.c_size_t


class PlanInfo(ctypes.Structure):
    _fields_ = [("num_nodes", _i32), ("num_windows", _i32), ("num_edges", _i64), ("tc_blocks", _i64),
                ("wide_blocks", _i64), ("plan_bytes", _i64), ("canonical", _i32), ("waves_per_window", _i32),
                ("column_buckets", _i32), ("lds_ranges", _i32)]


class TileStats(ctypes.Structure):
    _fields_ = [("windows", _i64), ("nonempty_windows", _i64), ("edges", _i64), ("unique_columns", _i64),
                ("sliding_tiles", _i64), ("condensed_tiles", _i64), ("max_sliding_per_window", _i64),
                ("max_condensed_per_window", _i64)]


# every symbol include/tcgnn.h declares, with its signature
SIGNATURES = {
    "tcgnn_abi_version": (ctypes.c_int, []),
    "tcgnn_status_string": (ctypes.c_char_p, [ctypes.c_int]),
    "tcgnn_last_error": (ctypes.c_char_p, []),
    "tcgnn_preprocess": (ctypes.c_int, [_i32p, _i32p, _i32, _i32, _i32, _i32p, _i64, _i32p, _i32p, ctypes.POINTER(_i64), _i32]),
    "tcgnn_preprocess_gpu": (ctypes.c_int, [_i32p, _i32p, _i32, _i64, _i32, _i32, _i32p, _i64, _i32p, _i32p, ctypes.POINTER(_i64), _vp]),
    "tcgnn_preprocess_gpu_workspace_bytes": (ctypes.c_int, [_i32, _i64, _i32, ctypes.POINTER(_sz)]),
    "tcgnn_preprocess_gpu_ws": (ctypes.c_int, [_i32p, _i32p, _i32, _i64, _i32, _i32, _i32p, _i64, _i32p, _i32p, _vp, _sz, ctypes.POINTER(_i64), _vp]),
    "tcgnn_tile_stats": (ctypes.c_int, [_i32p, _i32p, _i32, _i32, _i32, ctypes.POINTER(TileStats), _i32]),
    "tcgnn_plan_create": (ctypes.c_int, [_i32p, _i32p, _i32p, _i32p, _i32p, _i32, _i64, _i32, _vp, ctypes.POINTER(_vp)]),
    "tcgnn_plan_create_sharded": (ctypes.c_int, [_i32p, _i32p, _i32p, _i32p, _i32p, _i32, _i32, _i32, _i64, _i32, _vp, ctypes.POINTER(_vp)]),
    "tcgnn_plan_prepare": (ctypes.c_int, [_vp, _i32, _vp]),
    "tcgnn_plan_prepare_val": (ctypes.c_int, [_vp, _i32, _vp]),
    "tcgnn_plan_set_spmm_mode": (ctypes.c_int, [_vp, _i32]),
    "tcgnn_plan_set_range_guard": (ctypes.c_int, [_vp, _i32]),
    "tcgnn_plan_destroy": (ctypes.c_int, [_vp]),
    "tcgnn_plan_get_info": (ctypes.c_int, [_vp, ctypes.POINTER(PlanInfo)]),
    "tcgnn_set_spmm_mode": (ctypes.c_int, [_i32]),
    "tcgnn_set_range_guard": (ctypes.c_int, [_i32]),
    "tcgnn_range_mode": (ctypes.c_int, [_vp, _vp, ctypes.POINTER(_i32), ctypes.POINTER(_i32)]),
    "tcgnn_plan_set_timing": (ctypes.c_int, [_vp, _i32]),
    "tcgnn_plan_last_kernel": (ctypes.c_char_p, [_vp]),
    "tcgnn_plan_read_timing": (ctypes.c_int, [_vp, ctypes.POINTER(ctypes.c_float), _i32, ctypes.POINTER(_i32)]),
    "tcgnn_workspace_bytes": (_sz, [_vp, _i32]),
    "tcgnn_spmm": (ctypes.c_int, [_vp, _vp, _vp, _i32, _vp, _sz, _vp]),
    "tcgnn_spmm_val": (ctypes.c_int, [_vp, _vp, _vp, _vp, _i32, _vp, _sz, _vp]),
    "tcgnn_spmm_fused": (ctypes.c_int, [_vp, _vp, _vp, _vp, _i32, _i32, _vp, _sz, _vp]),
    "tcgnn_spmm_gemm": (ctypes.c_int, [_vp, _vp, _vp, _vp, _i32, _i32, _i32, _vp, _sz, _vp]),
    "tcgnn_x16_pitch": (ctypes.c_int, [_i32]),
    "tcgnn_stage_absmax": (ctypes.c_int, [_vp, _i64, _vp, _vp]),
    "tcgnn_stage_rows": (ctypes.c_int, [_vp, _i32, _i32, _vp, _vp, _vp]),
    "tcgnn_spmm_staged": (ctypes.c_int, [_vp, _vp, _vp, _i32, _vp]),
    "tcgnn_spmm_staged_layout": (ctypes.c_int, [_vp, _i32, _vp]),
    "tcgnn_stage_rows_planar": (ctypes.c_int, [_vp, _i32, _i32, _vp, _vp, _i64, _vp]),
    "tcgnn_spmm_staged_planar": (ctypes.c_int, [_vp, _vp, _vp, _i32, _vp]),
    "tcgnn_sddmm": (ctypes.c_int, [_vp, _vp, _vp, _i32, _vp, _sz, _vp]),
    "tcgnn_agnn_supported": (ctypes.c_int, [_vp, _i32]),
    "tcgnn_agnn_pair_forward": (ctypes.c_int, [_vp, _vp, _vp, _vp, _vp, _i64, _vp, _i32, _vp, _sz, _vp]),
    "tcgnn_agnn_pair_backward": (ctypes.c_int, [_vp, _vp, _vp, _vp, _vp, _i64, _vp, _vp, _i32, _vp, _sz, _vp]),
}
for _name, (_res, _args) in SIGNATURES.items():
    _fn = getattr(lib, _name)  # AttributeError here = header and library disagree
    _fn.restype = _res
    _fn.argtypes = _args


def check(status, what):
    if status != 0:
        msg = lib.tcgnn_last_error().decode("utf-8", "replace")
        kind = lib.tcgnn_status_string(status).decode()
        raise RuntimeError("%s failed: %s (%s)" % (what, kind, msg))


def build_id():
    """Identity of the kernels this process runs: SHA-256 over the library's sources (csrc/*.hip, *.inc, *.cpp, *.h, Makefile and
    include/tcgnn.h), 16 hex digits.  Profiles under profiles/ are keyed by it: bench.py quotes PMC numbers only when they were
    collected from the same sources (tools/collect_profiles.py), never a stale lookup."""
    import hashlib
    h = hashlib.sha256()
    src = os.path.join(_HERE, "csrc")
    files = sorted(f for f in os.listdir(src) if f.endswith((".hip", ".inc", ".cpp", ".h")) or f == "Makefile")
    for f in [os.path.join(src, f) for f in files] + [os.path.join(os.path.dirname(_HERE), "include", "tcgnn.h")]:
        h.update(os.path.basename(f).encode())
        with open(f, "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]
