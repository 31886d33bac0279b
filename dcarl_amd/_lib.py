"""ctypes binding of libdcarl_hip.so (include/dcarl.h).  There is NO CPU fallback: if the HIP library
cannot be loaded, or no gfx950 GPU is visible, every product entry point raises."""
from __future__ import annotations

import ctypes as C
import os
import threading

_HERE = os.path.dirname(os.path.abspath(__file__))
# DCARL_HIP_LIB selects another build of the same ABI (used by tools/experiments/ab_bench.sh for same-box A/B timing)
LIB_PATH = os.environ.get("DCARL_HIP_LIB") or os.path.join(_HERE, "libdcarl_hip.so")
# DCARL_LIB_VARIANT=ab: a whole process on the A/B variant (tools/ scripts); tests use use_variant() instead

DCARL_OK = 0
ABI_VERSION = 8
MAX_ACTIONS = 32
SLICE = 64


class DcarlError(RuntimeError):
    pass


class CParams(C.Structure):
    _fields_ = [("rule_act", C.c_int32), ("n_thres", C.c_int32), ("alpha", C.c_double), ("scale", C.c_double),
                ("cap", C.c_double), ("init_rule", C.c_double), ("init_other", C.c_double)]


class CTraceState(C.Structure):
    """dcarl_trace_state_t: device pointers of the online loop's sufficient statistic (include/dcarl.h)."""
    _fields_ = [("n", C.c_void_p), ("sum", C.c_void_p), ("sumsq", C.c_void_p), ("shift", C.c_void_p), ("V", C.c_void_p),
                ("act_step", C.c_void_p)]


class CRlsParams(C.Structure):
    _fields_ = [("visited_times_thres", C.c_int32), ("min_rl_visits", C.c_int32), ("rule_mean_gate", C.c_double),
                ("confidence_thres", C.c_double)]


class CFrenetGrid(C.Structure):
    _fields_ = [("n_d", C.c_int32), ("n_T", C.c_int32), ("n_v", C.c_int32), ("nt_max", C.c_int32),
                ("nt", C.c_int32 * 8), ("d", C.c_double * 16), ("T", C.c_double * 8), ("tv", C.c_double * 8),
                ("dt", C.c_double), ("target_speed", C.c_double), ("kj", C.c_double), ("kt", C.c_double),
                ("kd", C.c_double), ("klat", C.c_double), ("klon", C.c_double)]


class CFrenetLimits(C.Structure):
    _fields_ = [("max_speed", C.c_double), ("max_accel", C.c_double), ("max_curvature", C.c_double),
                ("check_radius", C.c_double), ("move_gap", C.c_double), ("n_predict", C.c_int32)]


class CDeviceInfo(C.Structure):
    _fields_ = [("arch", C.c_char * 32), ("compute_units", C.c_int32), ("wavefront", C.c_int32),
                ("hbm_bytes", C.c_int64)]


_vp, _i32, _i64, _u32, _u64, _f64 = C.c_void_p, C.c_int32, C.c_int64, C.c_uint32, C.c_uint64, C.c_double
_PP = C.POINTER(CParams)

# name -> (restype, argtypes); mirrors include/dcarl.h one to one (checked by tests/test_abi_surface.py)
SIGNATURES = {
    "dcarl_version": (_i32, []),
    "dcarl_build_id": (C.c_char_p, []),
    "dcarl_last_error": (C.c_char_p, []),
    "dcarl_device_info": (_i32, [_i32, C.POINTER(CDeviceInfo)]),
    "dcarl_default_params": (None, [_PP]),
    "dcarl_last_kernel": (C.c_char_p, []),
    "dcarl_workspace_bytes": (_i64, [_i32, _i64, _i32, _i64]),
    "dcarl_trace_f32": (_i32, [_vp, _vp, _vp, _vp, _vp, _i32, _i32, _PP, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "dcarl_trace_f64": (_i32, [_vp, _vp, _vp, _vp, _vp, _i32, _i32, _PP, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "dcarl_trace_resume_f32": (_i32, [_vp, _vp, _vp, _vp, _vp, _i32, _i32, _PP, C.POINTER(CTraceState), _i32, _vp, _vp, _vp, _vp, _vp]),
    "dcarl_trace_resume_f64": (_i32, [_vp, _vp, _vp, _vp, _vp, _i32, _i32, _PP, C.POINTER(CTraceState), _i32, _vp, _vp, _vp, _vp, _vp]),
    "dcarl_trace_status": (_i32, [_vp]),
    "dcarl_true_step_values_f32": (_i32, [_vp, _vp, _vp, _vp, _i32, _i32, _vp, _i32, _i64, _vp, _vp]),
    "dcarl_true_step_values_f64": (_i32, [_vp, _vp, _vp, _vp, _i32, _i32, _vp, _i32, _i64, _vp, _vp]),
    "dcarl_top2_census_trace_f32": (_i32, [_vp, _vp, _vp, _vp, _i32, _i32, _PP, _vp, _vp]),
    "dcarl_top2_census_trace_f64": (_i32, [_vp, _vp, _vp, _vp, _i32, _i32, _PP, _vp, _vp]),
    "dcarl_top2_census_table": (_i32, [_vp, _i32, _i32, _PP, _vp, _vp]),
    "dcarl_host_pin": (_i32, [_vp, _i64]),
    "dcarl_host_unpin": (_i32, [_vp]),
    "dcarl_copy_h2d": (_i32, [_vp, _vp, _i64, _vp]),
    "dcarl_copy_d2h": (_i32, [_vp, _vp, _i64, _vp]),
    "dcarl_debug_raise_trace_fault": (_i32, []),
    "dcarl_count_nonfinite": (_i32, [_vp, _i32, _i64, _vp, _vp]),
    "dcarl_bounds_csr_f32": (_i32, [_vp, _vp, _i64, _i64, _i32, _i32, _PP, _vp, _vp, _vp, _vp, _vp]),
    "dcarl_bounds_csr_f64": (_i32, [_vp, _vp, _i64, _i64, _i32, _i32, _PP, _vp, _vp, _vp, _vp, _vp]),
    "dcarl_count_records": (_i32, [_vp, _vp, _vp, _vp, _i32, _i32, _vp, _vp]),
    "dcarl_group_records_f32": (_i32, [_vp, _vp, _vp, _vp, _vp, _i32, _i32, _vp, _vp, _vp]),
    "dcarl_group_records_f64": (_i32, [_vp, _vp, _vp, _vp, _vp, _i32, _i32, _vp, _vp, _vp]),
    "dcarl_bucket_bounds_f32": (_i32, [_vp, _vp, _i64, _PP, _vp, _vp]),
    "dcarl_bucket_bounds_f64": (_i32, [_vp, _vp, _i64, _PP, _vp, _vp]),
    "dcarl_overall_delta_f32": (_i32, [_vp, _vp, _vp, _vp, _vp, _i64, _vp, _vp, _vp, _vp]),
    "dcarl_overall_delta_f64": (_i32, [_vp, _vp, _vp, _vp, _vp, _i64, _vp, _vp, _vp, _vp]),
    "dcarl_scan_workspace_bytes": (_i64, [_i64]),
    "dcarl_scan_f64": (_i32, [_vp, _vp, _i64, _vp, _vp]),
    "dcarl_ingest_workspace_bytes": (_i64, [_i64, _i32, _i32, _i32, _i32, _i32]),
    "dcarl_ingest_group_f32": (_i32, [_vp, _i64, _i32, _i32, _i32, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "dcarl_ingest_group_f64": (_i32, [_vp, _i64, _i32, _i32, _i32, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "dcarl_ingest_group_pairs_f32": (_i32, [_vp, _vp, _vp, _i64, _i32, _i32, _i32, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "dcarl_host_compact_rows_f32": (_i32, [_vp, _i64, _i32, _i32, _vp, _vp]),
    "dcarl_ingest_group_packed_f32": (_i32, [_vp, _i64, _i32, _i32, _i32, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "dcarl_ingest_pack_f32": (_i32, [_i64, _i32, _i32, _i32, _vp, _vp, _vp, _vp, _i64, _vp, _vp, _vp, _vp, _vp]),
    "dcarl_ingest_pack_f64": (_i32, [_i64, _i32, _i32, _i32, _vp, _vp, _vp, _vp, _i64, _vp, _vp, _vp, _vp, _vp]),
    "dcarl_slot_order_workspace_bytes": (_i64, [_i32]),
    "dcarl_slot_order": (_i32, [_vp, _i32, _i64, _i32, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "dcarl_export_records_f32": (_i32, [_vp, _vp, _vp, _vp, _vp, _i32, _i64, _i64, _vp, _vp, _i64, _vp, _vp]),
    "dcarl_export_records_f64": (_i32, [_vp, _vp, _vp, _vp, _vp, _i32, _i64, _i64, _vp, _vp, _i64, _vp, _vp]),
    "dcarl_ingest_buckets_f32": (_i32, [_vp, _i64, _i32, _i32, _vp, _vp, _vp, _vp, _vp]),
    "dcarl_ingest_buckets_f64": (_i32, [_vp, _i64, _i32, _i32, _vp, _vp, _vp, _vp, _vp]),
    "dcarl_sample_state_records": (_i32, [_vp, _i32, _i32, _i32, _i64, _f64, _u64, _u32, _vp, _vp, _vp]),
    "dcarl_sample_state_records_ragged": (_i32, [_vp, _i32, _i32, _i32, _vp, _i64, _vp, _vp, _vp, _f64, _u64, _u32, _u32, _vp, _vp,
                                                  _vp, _vp]),
    "dcarl_sample_buckets": (_i32, [_vp, _i32, _i32, _i32, _vp, _i64, _f64, _u64, _u32, _vp, _vp]),
    "dcarl_summary_stats": (_i32, [_vp, _vp, _vp, _i32, _i32, _vp, _vp, _vp]),
    "dcarl_comm_unique_id": (_i32, [_vp]),
    "dcarl_comm_init": (_i32, [_i32, _i32, _vp, C.POINTER(C.c_void_p)]),
    "dcarl_allgather_summary": (_i32, [_vp, _vp, _vp, _i64, _vp]),
    "dcarl_comm_destroy": (_i32, [_vp]),
    "dcarl_sample_pairs": (_i32, [_vp, _i32, _i32, _i64, _f64, _u64, _u64, _u32, _vp, _vp, _vp, _vp, _vp]),
    "dcarl_visit_index_f64": (_i32, [_vp, _i64, _i32, _vp, _vp]),
    "dcarl_visit_floor_f64": (_i32, [_vp, _i64, _i32, _vp, _vp]),
    "dcarl_state_manual_f64": (_i32, [_vp, _vp, _vp, _i64, _vp, _vp]),
    "dcarl_sample_from_noise_f64": (_i32, [_vp, _vp, _i64, _vp, _vp, _i32, _i32, _vp, _vp, _f64, _vp, _vp]),
    "dcarl_gamma_powers": (None, [_f64, _i32, C.POINTER(C.c_double)]),
    "dcarl_episode_returns_f64": (_i32, [_vp, _vp, _vp, _vp, _i64, _vp, _vp, _vp, _vp]),
    "dcarl_nstep_backup_f64": (_i32, [_vp, _vp, _vp, _i64, _vp, _i32, _vp, _vp, _vp]),
    "dcarl_state_cells_f64": (_i32, [_vp, _i64, _i32, _vp, _vp, _vp, _vp]),
    "dcarl_state_ids": (_i32, [_vp, _vp, _i64, _i32, _i64, _vp, _vp, _vp, _vp]),
    "dcarl_index_states_f64": (_i32, [_vp, _i64, _i32, _vp, _i64, _vp, _vp, _vp, _vp, _vp]),
    "dcarl_frenet_default_grid": (None, [C.POINTER(CFrenetGrid)]),
    "dcarl_frenet_candidates_f64": (_i32, [_vp, _i64, C.POINTER(CFrenetGrid), _vp, _vp, _vp]),
    "dcarl_frenet_default_limits": (None, [C.POINTER(CFrenetLimits)]),
    "dcarl_frenet_global_paths_f64": (_i32, [_vp, _i64, C.POINTER(CFrenetGrid), _vp, _vp, _i32, _vp, _vp, _vp]),
    "dcarl_frenet_select": (_i32, [_vp, _vp, _vp, _vp, _vp, _i32, _i64, C.POINTER(CFrenetGrid), C.POINTER(CFrenetLimits),
                                    _vp, _vp, _vp]),
    "dcarl_rls_default_params": (None, [C.POINTER(CRlsParams)]),
    "dcarl_rls_workspace_bytes": (_i64, [_i64, _i32]),
    "dcarl_rls_neighbour_stats_f64": (_i32, [_vp, _vp, _i64, _vp, _vp, _i32, _vp, _vp, _vp, _vp, _vp]),
    "dcarl_rls_gate_train": (_i32, [_vp, _vp, _vp, _vp, _i32, _i32, _vp, _vp, _vp]),
    "dcarl_rls_decide": (_i32, [_vp, _vp, _vp, _i32, _i32, C.POINTER(CRlsParams), _vp, _vp]),
}

_libs = {}                      # variant -> typed library ("" = the product library)
_variant = os.environ.get("DCARL_LIB_VARIANT", "")      # what load() hands out; otherwise only use_variant() changes it
_lock = threading.Lock()


def _load_variant(variant: str):
    # (re)build in-tree when the library is missing or older than its sources and a compiler is at hand; a stale
    # .so would otherwise pass the version check and run old kernels against new host code.  DCARL_HIP_LIB (an
    # explicitly chosen build of the PRODUCT variant's ABI) is never rebuilt.
    explicit = os.environ.get("DCARL_HIP_LIB") if not variant else None
    from . import build as _build
    path = explicit or _build.lib_path(variant)
    if not explicit:
        try:
            if _build.needs_build(variant) and (_build.have_hipcc() or not os.path.exists(path)):
                _build.build(variant=variant)
        except Exception as e:  # noqa: BLE001
            if not os.path.exists(path):
                raise DcarlError(f"{os.path.basename(path)} is missing at {path} and could not be built: {e}") from e
            raise DcarlError(f"{path} is older than its sources and the rebuild failed: {e}") from e
    try:
        lib = C.CDLL(path)
    except OSError as e:
        raise DcarlError(f"cannot load {path}: {e}") from e
    for name, (res, args) in SIGNATURES.items():
        try:
            fn = getattr(lib, name)
        except AttributeError as e:
            raise DcarlError(f"{path} does not export {name}") from e
        fn.restype = res
        fn.argtypes = args
    if lib.dcarl_version() != ABI_VERSION:
        raise DcarlError(f"ABI version mismatch: library {lib.dcarl_version()}, binding {ABI_VERSION}")
    if not explicit:
        want, have = _build.source_id(variant), lib.dcarl_build_id().decode()
        if have != want:
            raise DcarlError(f"{path} was built from other sources (build id {have}, sources {want}) and no "
                             f"compiler is available to rebuild it")
    return lib


def load():
    """Load (building in-tree with hipcc if necessary) and type the library.  Raises DcarlError loudly."""
    lib = _libs.get(_variant)
    if lib is not None:
        return lib
    with _lock:
        lib = _libs.get(_variant)
        if lib is None:
            lib = _libs[_variant] = _load_variant(_variant)
    return lib


class use_variant:
    """``with _lib.use_variant("ab"):`` — inside, ``load()`` hands out libdcarl_hip_ab.so (-DDCARL_AB_BUILD: the DCARL_* environment
    overrides of the launchers' choices and the measurement-only kernel instances exist there and only there).  For the tests of
    every kernel instance and tools/'s A/B scripts; objects that cached the library (``ConfidenceEstimator``) must be created
    inside.  Process-wide, not thread-local: not for product code."""

    def __init__(self, variant: str):
        self.variant, self.prev = variant, ""

    def __enter__(self):
        global _variant
        self.prev, _variant = _variant, self.variant
        try:
            return load()
        except BaseException:
            _variant = self.prev                              # a failed build / load must not leave the process on the variant
            raise

    def __exit__(self, *exc):
        global _variant
        _variant = self.prev
        return False


def check(rc, what=""):
    if rc != DCARL_OK:
        msg = load().dcarl_last_error().decode(errors="replace")
        raise DcarlError(f"{what or 'dcarl call'} failed with code {rc}: {msg}")


def require_gpu():
    """Returns the torch device to use; raises if there is no gfx950 GPU (no CPU fallback by design)."""
    import torch
    if not torch.cuda.is_available():
        raise DcarlError("no ROCm GPU visible: dcarl_amd has no CPU path (the CPU restatement lives in oracle/ "
                         "and is test infrastructure only)")
    dev = torch.cuda.current_device()
    info = CDeviceInfo()
    check(load().dcarl_device_info(dev, C.byref(info)), "dcarl_device_info")
    return torch.device("cuda", dev)


def device_info(dev=None):
    import torch
    info = CDeviceInfo()
    check(load().dcarl_device_info(torch.cuda.current_device() if dev is None else dev, C.byref(info)),
          "dcarl_device_info")
    return dict(arch=info.arch.decode(), compute_units=info.compute_units, wavefront=info.wavefront,
                hbm_bytes=info.hbm_bytes)


def last_kernel() -> str:
    """Template instance the last trace / bounds call of this thread launched (dcarl_last_kernel)."""
    return load().dcarl_last_kernel().decode()


def ptr(t):
    """Device pointer of a tensor (None -> NULL)."""
    return None if t is None else C.c_void_p(t.data_ptr())


def stream_ptr():
    import torch
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)
