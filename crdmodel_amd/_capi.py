"""ctypes declarations of the libcrd C ABI (include/crd.h).  Thin: every call goes straight to the HIP library.

There is no Python or CPU fallback: if libcrd.so is missing or cannot be loaded, importing this module's `lib()`
raises, and any device call without a GPU returns CRD_EHIP which `check()` turns into an exception.
"""
import ctypes as C
import importlib.util
import os
import sys

_PKG = os.path.dirname(os.path.abspath(__file__))
# CRD_LIBRARY points the binding at another build of the same ABI (tuning builds under tools/); default is the in-tree library.
LIB_PATH = os.environ.get("CRD_LIBRARY") or os.path.join(_PKG, "libcrd.so")

ABI_VERSION = 6  # CRD_ABI_VERSION of include/crd.h (tests/test_host_abi.py keeps the two in step)
OK, EINVAL, ENOMEM, EHIP, ERCCL, EIO, EPARSE, ESTATE = 0, -1, -2, -3, -4, -5, -6, -7
MODEL_FHN, MODEL_GOLDBETER = 0, 1
SURFACE_TORUS, SURFACE_FLAT = 0, 1
PRECISION_F64, PRECISION_F32 = 0, 1
STEPPER_AUTO, STEPPER_STAGED, STEPPER_FUSED = 0, 1, 2
ADAPT_RK43, ADAPT_ARKODE = 0, 1

MODELS = {"fhn": MODEL_FHN, "goldbeter": MODEL_GOLDBETER}
SURFACES = {"torus": SURFACE_TORUS, "flat": SURFACE_FLAT}
STEPPERS = {"auto": STEPPER_AUTO, "staged": STEPPER_STAGED, "fused": STEPPER_FUSED}


class Params(C.Structure):
    """crd_params"""

    _fields_ = [
        ("model", C.c_int32), ("surface", C.c_int32),
        ("nx", C.c_int64), ("ny", C.c_int64),
        ("surface_length", C.c_double), ("surface_width", C.c_double),
        ("diffusion", C.c_double), ("beta", C.c_double), ("beta_min", C.c_double), ("beta_max", C.c_double),
        ("vary_beta", C.c_int32), ("just_diffusion", C.c_int32),
        ("t_boundary", C.c_double),
        ("precision", C.c_int32), ("reserved", C.c_int32),
    ]


class Grid(C.Structure):
    """crd_grid"""

    _fields_ = [
        ("nx", C.c_int64), ("ny", C.c_int64), ("dx", C.c_double), ("dy", C.c_double),
        ("xmin", C.c_double), ("xmax", C.c_double), ("ymin", C.c_double), ("ymax", C.c_double),
        ("R", C.c_double), ("r", C.c_double),
    ]


class HaloOp(C.Structure):
    """crd_halo_op"""

    _fields_ = [("is_send", C.c_int32), ("peer", C.c_int32), ("row_begin", C.c_int64), ("row_count", C.c_int64)]


class AdaptiveOptions(C.Structure):
    """crd_adaptive_options"""

    _fields_ = [("rtol", C.c_double), ("atol", C.c_double), ("h0", C.c_double), ("safety", C.c_double), ("bias", C.c_double),
                ("growth", C.c_double), ("shrink", C.c_double), ("max_steps", C.c_int64), ("h_max", C.c_double), ("dense_output", C.c_int32),
                ("method", C.c_int32)]


class AdaptiveStats(C.Structure):
    """crd_adaptive_stats"""

    _fields_ = [("accepted", C.c_int64), ("rejected", C.c_int64), ("h_last", C.c_double), ("h_next", C.c_double), ("h_min", C.c_double),
                ("h_max", C.c_double), ("err_last", C.c_double), ("t", C.c_double), ("t_internal", C.c_double), ("h_first", C.c_double),
                ("launched_ahead", C.c_int64)]


class RunConfig(C.Structure):
    """crd_run_config"""

    _fields_ = [
        ("params", Params),
        ("wave_length", C.c_double), ("wave_width", C.c_double),
        ("wave_inside", C.c_int32), ("output_timestep", C.c_int32),
        ("t_final", C.c_double),
        ("include_all_vars", C.c_int32), ("ic_type", C.c_int32),
        ("dt", C.c_double), ("dt_safety", C.c_double),
        ("n_gpus", C.c_int32), ("stepper", C.c_int32),
        ("adaptive", C.c_int32), ("steady_state_decimals", C.c_int32),
        ("rtol", C.c_double), ("atol", C.c_double),
        ("exchange_period", C.c_int32), ("reserved", C.c_int32),
    ]


class LaunchPlan(C.Structure):
    """crd_launch_plan"""

    _fields_ = [("autotune", C.c_int32), ("tuned", C.c_int32), ("one_round", C.c_int32), ("xcd_mapping", C.c_int32), ("rows", C.c_int32),
                ("columns_per_lane", C.c_int32), ("nontemporal_stores", C.c_int32), ("steps_per_launch", C.c_int32), ("ms_default", C.c_double),
                ("ms_chosen", C.c_double)]


class LaunchGeometry(C.Structure):
    """crd_launch_geometry"""

    _fields_ = [(f, C.c_int32) for f in ("rows", "strips", "chunk_rows", "chunks", "workgroups", "wavefronts_per_workgroup", "fill_iterations", "iterations_per_trip",
                                         "lanes", "lanes_valid", "vgprs", "sgprs", "lds_bytes", "scratch_bytes", "wavefronts_per_simd", "loop_valu", "loop_salu",
                                         "loop_vmem", "loop_lds", "loop_instructions", "simds", "clock_khz", "exec_skipped_vmem")] + [("wavefront_iterations", C.c_int64), ("wavefront_iterations_effective", C.c_int64)]


class StepTiming(C.Structure):
    """crd_step_timing"""

    _fields_ = [("ms_total", C.c_double), ("kernel_ms", C.c_double), ("exposed_halo_ms", C.c_double), ("exchange_ms", C.c_double),
                ("steps", C.c_int64), ("halo_slack", C.c_int32), ("timed_steps_per_launch", C.c_int32), ("halo_waits", C.c_int32), ("exchanges", C.c_int32),
                ("agreement_restarts", C.c_int64)]


# name -> (restype, argtypes); the test suite checks this table against include/crd.h symbol by symbol.
_vp = C.c_void_p
_SIGNATURES = {
    "crd_abi_version": (C.c_int, []),
    "crd_status_string": (C.c_char_p, [C.c_int]),
    "crd_kernel_table_digest": (C.c_char_p, []),
    "crd_config_load_ini": (C.c_int, [C.c_char_p, C.c_int, C.c_int, C.POINTER(RunConfig), C.c_char_p, C.c_size_t]),
    "crd_grid_from_params": (C.c_int, [C.POINTER(Params), C.POINTER(Grid)]),
    "crd_slab_extents": (C.c_int, [C.c_int64, C.c_int, C.c_int, C.POINTER(C.c_int64), C.POINTER(C.c_int64)]),
    "crd_dims_create": (C.c_int, [C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "crd_block_extents": (C.c_int, [C.c_int64, C.c_int64, C.c_int, C.c_int, C.c_int, C.c_int] + [C.POINTER(C.c_int64)] * 4),
    "crd_steady_state": (C.c_int, [C.c_int, C.c_double, C.POINTER(C.c_double), C.POINTER(C.c_double)]),
    "crd_steady_state_as_printed": (C.c_int, [C.c_int, C.c_double, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_double)]),
    "crd_initial_conditions": (C.c_int, [C.POINTER(RunConfig), C.c_int64, C.c_int64, _vp]),
    "crd_initial_conditions_block": (C.c_int, [C.POINTER(RunConfig), C.c_int64, C.c_int64, C.c_int64, C.c_int64, _vp]),
    "crd_stable_dt": (C.c_double, [C.POINTER(Params)]),
    "crd_writer_open": (C.c_int, [C.POINTER(RunConfig), C.c_char_p, C.c_int, C.c_int, C.POINTER(_vp)]),
    "crd_writer_open_block": (C.c_int, [C.POINTER(RunConfig), C.c_char_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(_vp)]),
    "crd_writer_write_row": (C.c_int, [_vp, _vp]),
    "crd_writer_close": (C.c_int, [_vp]),
    "crd_npy_writer_open": (C.c_int, [C.POINTER(RunConfig), C.c_char_p, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(_vp)]),
    "crd_npy_writer_append": (C.c_int, [_vp, _vp]),
    "crd_npy_writer_close": (C.c_int, [_vp]),
    "crd_device_count": (C.c_int, []),
    "crd_create": (C.c_int, [C.POINTER(Params), C.c_int, C.c_int, C.c_int, C.POINTER(_vp)]),
    "crd_create_block": (C.c_int, [C.POINTER(Params), C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(_vp)]),
    "crd_get_block": (C.c_int, [_vp] + [C.POINTER(C.c_int64)] * 4),
    "crd_destroy": (None, [_vp]),
    "crd_last_error": (C.c_char_p, [_vp]),
    "crd_get_grid": (C.c_int, [_vp, C.POINTER(Grid)]),
    "crd_get_slab": (C.c_int, [_vp, C.POINTER(C.c_int64), C.POINTER(C.c_int64)]),
    "crd_comm_attach_local": (C.c_int, [C.POINTER(_vp), C.c_int]),
    "crd_comm_set_rccl_library": (C.c_int, [C.c_char_p]),
    "crd_comm_unique_id": (C.c_int, [_vp]),
    "crd_comm_init_rccl": (C.c_int, [_vp, _vp]),
    "crd_comm_info": (C.c_int, [_vp, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "crd_halo_exchange": (C.c_int, [_vp, C.c_int]),
    "crd_state_download_rows": (C.c_int, [_vp, C.c_int, C.c_int64, C.c_int64, _vp]),
    "crd_halo_plan": (C.c_int, [C.c_int, C.c_int, C.c_int64, C.c_int, C.POINTER(HaloOp)]),
    "crd_cycle_vote": (C.c_int, [C.c_int, C.POINTER(C.c_double)]),
    "crd_cycle_agreed": (C.c_int, [C.POINTER(C.c_double)]),
    "crd_state_upload": (C.c_int, [_vp, _vp, C.c_int]),
    "crd_state_download": (C.c_int, [_vp, _vp, C.c_int]),
    "crd_host_alloc": (_vp, [C.c_size_t]),
    "crd_host_free": (None, [_vp]),
    "crd_rhs_host": (C.c_int, [_vp, C.c_double, _vp, _vp]),
    "crd_rhs_device": (C.c_int, [_vp, C.c_double, _vp, _vp]),
    "crd_set_stepper": (C.c_int, [_vp, C.c_int]),
    "crd_step_rk4": (C.c_int, [_vp, C.c_double, C.c_double, C.c_int64]),
    "crd_synchronize": (C.c_int, [_vp]),
    "crd_adaptive_defaults": (C.c_int, [C.POINTER(AdaptiveOptions)]),
    "crd_integrate_adaptive": (C.c_int, [_vp, C.c_double, C.c_double, C.POINTER(AdaptiveOptions), C.POINTER(AdaptiveStats)]),
    "crd_group_integrate_adaptive": (C.c_int, [C.POINTER(_vp), C.c_int, C.c_double, C.c_double, C.POINTER(AdaptiveOptions),
                                             C.POINTER(AdaptiveStats)]),
    "crd_group_step_rk4": (C.c_int, [C.POINTER(_vp), C.c_int, C.c_double, C.c_double, C.c_int64]),
    "crd_group_step_rk4_timed": (C.c_int, [C.POINTER(_vp), C.c_int, C.c_double, C.c_double, C.c_int64]),
    "crd_group_rhs_device": (C.c_int, [C.POINTER(_vp), C.c_int, C.c_double, C.POINTER(_vp), C.POINTER(_vp)]),
    "crd_group_rhs_host": (C.c_int, [C.POINTER(_vp), C.c_int, C.c_double, C.POINTER(_vp), C.POINTER(_vp)]),
    "crd_step_rk4_timed": (C.c_int, [_vp, C.c_double, C.c_double, C.c_int64, C.POINTER(C.c_double),
                                     C.POINTER(C.c_double), C.POINTER(C.c_int)]),
    "crd_dominant_kernel_rows": (C.c_int, [_vp, C.POINTER(C.c_int64)]),
    "crd_dominant_kernel_name": (C.c_char_p, [_vp]),
    "crd_state_max_abs": (C.c_int, [_vp, C.POINTER(C.c_double)]),
    "crd_trace_range_push": (None, [C.c_char_p]),
    "crd_trace_range_pop": (None, []),
    "crd_set_autotune": (C.c_int, [_vp, C.c_int]),
    "crd_get_launch_plan": (C.c_int, [_vp, C.POINTER(LaunchPlan)]),
    "crd_get_launch_geometry": (C.c_int, [_vp, C.POINTER(LaunchGeometry)]),
    "crd_set_launch_plan": (C.c_int, [_vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]),
    "crd_plan_launches": (C.c_int, [_vp]),
    "crd_set_diagnostics": (C.c_int, [_vp, C.c_int]),
    "crd_set_halo_slack": (C.c_int, [_vp, C.c_int]),
    "crd_set_exchange_period": (C.c_int, [_vp, C.c_int]),
    "crd_get_exchange_period": (C.c_int, [_vp]),
    "crd_group_set_threads": (C.c_int, [C.POINTER(_vp), C.c_int, C.c_int]),
    "crd_launch_plan_candidate": (C.c_int, [C.c_int, C.POINTER(LaunchPlan)]),
    "crd_get_step_timing": (C.c_int, [_vp, C.POINTER(StepTiming)]),
}

_lib = None


class CrdError(RuntimeError):
    def __init__(self, status, where, detail=""):
        self.status = status
        msg = "%s failed: %s (%d)" % (where, lib().crd_status_string(status).decode(), status)
        if detail:
            msg += ": " + detail
        super().__init__(msg)


def _one_hip_runtime():
    """Keep ONE HIP runtime in the process.  PyTorch bundles its own libamdhip64 under the same SONAME as /opt/rocm's.
    When PyTorch is installed but not imported yet, map its copy first and let libcrd bind to it by SONAME -- the same
    pairing bench.py gets by importing torch first -- so a later `import torch` finds its own runtime already in place.
    (RCCL is not touched here: libcrd binds it lazily with dlopen at the first crd_comm_* call.)
    CRD_SYSTEM_ROCM=1 keeps /opt/rocm's runtime."""
    if "torch" in sys.modules or os.environ.get("CRD_SYSTEM_ROCM") == "1":
        return
    spec = importlib.util.find_spec("torch")
    if spec is None or not spec.origin:
        return
    libdir = os.path.join(os.path.dirname(spec.origin), "lib")
    for name in ("libamdhip64.so",):
        path = os.path.join(libdir, name)
        if os.path.exists(path):
            C.CDLL(path, mode=C.RTLD_GLOBAL)


def lib():
    """Load libcrd.so (built in-tree by crdmodel_amd.build / `make -C crdmodel_amd/csrc`)."""
    global _lib
    if _lib is None:
        _one_hip_runtime()
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                "libcrd.so not found at %s: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "or `make -C crdmodel_amd/csrc`; there is no CPU fallback" % LIB_PATH)
        L = C.CDLL(LIB_PATH, mode=C.RTLD_GLOBAL)
        for name, (res, args) in _SIGNATURES.items():
            fn = getattr(L, name)  # AttributeError here = header / library mismatch
            fn.restype = res
            fn.argtypes = args
        if L.crd_abi_version() != ABI_VERSION:
            raise ImportError("libcrd.so ABI version mismatch")
        if os.environ.get("CRD_RCCL_LIBRARY"):  # another RCCL build / the multi-process ring tests' stand-in (tests/native)
            if L.crd_comm_set_rccl_library(os.environ["CRD_RCCL_LIBRARY"].encode()) != OK:
                raise ImportError("crd_comm_set_rccl_library failed")
        _lib = L
    return _lib


def check(status, where, ctx=None):
    if status != OK:
        detail = lib().crd_last_error(ctx).decode() if ctx is not None or status in (EHIP, ERCCL, EINVAL, ENOMEM) else ""
        raise CrdError(status, where, detail)
