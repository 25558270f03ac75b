"""ctypes view of the C-ABI in include/mola_icp_amd.h (nothing else is bound).

The shared library is the product; if it is missing this module raises -- there
is no Python/CPU fallback for the hot path."""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# (MOLA_ICP_LIB_PATH: a sanitizer build of the host code -- tools/sanitize.sh; never needed in production)
LIB_PATH = os.environ.get("MOLA_ICP_LIB_PATH") or os.path.join(_HERE, "lib", "libmola_icp_amd.so")

NACC = 24
ABI_VERSION = 5   # MOLA_ICP_ABI_VERSION of include/mola_icp_amd.h

OK = 0
E_BADARG, E_CONFIG, E_HIP, E_OOM, E_NODEVICE, E_UNSUPPORTED, E_COMM, E_INTERNAL = -1, -2, -3, -4, -5, -6, -7, -8
TERM_UNDEFINED, TERM_NO_PAIRINGS, TERM_SOLVER_ERROR, TERM_MAX_ITERATIONS, TERM_STALLED = 0, 1, 2, 3, 4
MATCHER_POINTS_DISTANCE_THRESHOLD, MATCHER_POINT2PLANE = 0, 1
SOLVER_HORN, SOLVER_GAUSS_NEWTON = 0, 1
QUALITY_PAIRED_RATIO = 0
NN_AUTO, NN_VALU, NN_MFMA, NN_TILED = 0, 1, 2, 3


MAX_EXTRA_STAGES = 3


class CMatcherEntry(C.Structure):
    _fields_ = [("matcher_class", C.c_int32), ("matcher_threshold", C.c_double), ("plane_eigen_threshold", C.c_double),
                ("knn", C.c_uint32), ("run_from_iteration", C.c_uint32), ("run_up_to_iteration", C.c_uint32)]


class CSolverEntry(C.Structure):
    _fields_ = [("solver_class", C.c_int32), ("solver_max_iterations", C.c_uint32), ("run_from_iteration", C.c_uint32),
                ("run_up_to_iteration", C.c_uint32)]


class CQualityEntry(C.Structure):
    _fields_ = [("quality_class", C.c_int32), ("quality_threshold", C.c_double), ("weight", C.c_double)]


class CParams(C.Structure):
    _fields_ = [
        ("max_iterations", C.c_uint32),
        ("min_abs_step_trans", C.c_double),
        ("min_abs_step_rot", C.c_double),
        ("use_scale_outlier_detector", C.c_int32),
        ("scale_outlier_threshold", C.c_double),
        ("use_robust_kernel", C.c_int32),
        ("robust_kernel_param", C.c_double),
        ("robust_kernel_scale", C.c_double),
        ("solver_class", C.c_int32),
        ("solver_max_iterations", C.c_uint32),
        ("matcher_class", C.c_int32),
        ("matcher_threshold", C.c_double),
        ("plane_eigen_threshold", C.c_double),
        ("knn", C.c_uint32),
        ("run_from_iteration", C.c_uint32),
        ("run_up_to_iteration", C.c_uint32),
        ("quality_class", C.c_int32),
        ("quality_threshold", C.c_double),
        ("fixed_iterations", C.c_int32),
        ("nn_kernel", C.c_int32),
        ("skip_quality", C.c_int32),
        ("solver_run_from_iteration", C.c_uint32),
        ("solver_run_up_to_iteration", C.c_uint32),
        ("n_extra_matchers", C.c_uint32),
        ("n_extra_solvers", C.c_uint32),
        ("extra_matchers", CMatcherEntry * MAX_EXTRA_STAGES),
        ("extra_solvers", CSolverEntry * MAX_EXTRA_STAGES),
        ("quality_weight", C.c_double),
        ("n_extra_quality", C.c_uint32),
        ("extra_quality", CQualityEntry * MAX_EXTRA_STAGES),
        ("reading_outlier_single_pass", C.c_int32),
        ("reading_p2pl_all_inside_gate", C.c_int32),
        ("reading_quality_denominator_local", C.c_int32),
        ("reading_robust_kernel_skips_planes", C.c_int32),
    ]


class CResult(C.Structure):
    _fields_ = [
        ("T", C.c_double * 16),
        ("cov", C.c_double * 36),
        ("quality", C.c_double),
        ("n_iterations", C.c_uint32),
        ("termination", C.c_uint32),
        ("n_pairs", C.c_uint64),
        ("rmse", C.c_double),
        ("ms_upload", C.c_double),
        ("ms_iterations", C.c_double),
        ("ms_quality", C.c_double),
        ("ms_nn_kernel", C.c_double),
        ("n_nn_launches", C.c_uint32),
        ("nn_kernel_used", C.c_uint32),
        ("nn_pairs_evaluated", C.c_uint64),
    ]


class CLoParams(C.Structure):
    _fields_ = [
        ("min_time_between_scans", C.c_double),
        ("min_dist_xyz_between_keyframes", C.c_double),
        ("min_rotation_between_keyframes", C.c_double),
        ("min_icp_goodness", C.c_double),
        ("icp_with_vel", CParams),
        ("icp_without_vel", CParams),
        ("icp_loop_closure", CParams),
        ("min_icp_goodness_lc", C.c_double),
        ("min_dist_to_matching", C.c_double),
        ("max_dist_to_matching", C.c_double),
        ("max_dist_to_loop_closure", C.c_double),
        ("loop_closure_montecarlo_samples", C.c_uint32),
        ("max_nearby_align_checks", C.c_uint32),
        ("min_topo_dist_to_consider_loopclosure", C.c_uint32),
        ("max_kfs_local_graph", C.c_uint32),
    ]


class CLoStep(C.Structure):
    _fields_ = [
        ("status", C.c_int32),
        ("used_with_vel_params", C.c_int32),
        ("dt", C.c_double),
        ("rel_pose", C.c_double * 16),
        ("twist", C.c_double * 4),
        ("dist_since_last_kf", C.c_double),
        ("rot_since_last_kf", C.c_double),
        ("keyframe_created", C.c_int32),
        ("kf_factor_valid", C.c_int32),
        ("kf_factor_from", C.c_uint64),
        ("kf_factor_to", C.c_uint64),
        ("kf_factor_pose", C.c_double * 16),
        ("reference_kf", C.c_uint64),
        ("accum_since_last_kf", C.c_double * 16),
        ("icp", CResult),
    ]


class CLoKfCandidate(C.Structure):
    _fields_ = [("kf_id", C.c_uint64), ("eucl_dist", C.c_double), ("topo_dist", C.c_uint32), ("already_checked", C.c_int32)]


class CLoCheckResult(C.Structure):
    _fields_ = [("icp", CResult), ("best_guess", C.c_int32), ("n_attempts", C.c_uint32), ("init_guess_used", C.c_double * 6),
                ("correction_percent", C.c_double), ("edge_accepted", C.c_int32)]


LO_ALIGN_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(C.c_float), C.POINTER(C.c_float), C.POINTER(C.c_float), C.c_size_t,
                          C.POINTER(C.c_float), C.POINTER(C.c_float), C.POINTER(C.c_float), C.c_size_t,
                          C.POINTER(C.c_double), C.POINTER(CParams), C.POINTER(CResult))
LO_DROPPED_TOO_SOON, LO_FIRST_SCAN, LO_ICP_RAN, LO_EMPTY_CLOUD = 0, 1, 2, 3

ALLREDUCE_FN = C.CFUNCTYPE(C.c_int, C.POINTER(C.c_double), C.c_int, C.c_int, C.c_void_p)
MATCH_CB = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(C.c_double), C.c_double, C.POINTER(C.c_uint64))
ACCUM_CB = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(CParams), C.POINTER(C.c_double), C.c_int,
                       C.POINTER(C.c_double), C.POINTER(C.c_double), C.c_int, C.POINTER(C.c_double))


class CStageCallbacks(C.Structure):
    _fields_ = [
        ("match", MATCH_CB),
        ("accumulate", ACCUM_CB),
        ("allreduce", ALLREDUCE_FN),
        ("user", C.c_void_p),
        ("n_local_total", C.c_uint64),
        ("n_map_total", C.c_uint64),
    ]


_FP = C.POINTER(C.c_float)
_DP = C.POINTER(C.c_double)
_H = C.c_void_p

#: every symbol include/mola_icp_amd.h declares: name -> (restype, argtypes)
SIGNATURES = {
    "mola_icp_abi_version": (C.c_int, []),
    "mola_icp_set_profiling": (C.c_int, [_H, C.c_int]),
    "mola_icp_forget_warm_start": (C.c_int, [_H]),
    "mola_icp_forget_cloud_schedule": (C.c_int, [_H]),
    "mola_icp_set_local_shard_host": (C.c_int, [_H, _FP, _FP, _FP, C.c_size_t, C.c_int, C.c_int, C.POINTER(C.c_size_t)]),
    "mola_icp_set_local_shard_device": (C.c_int, [_H, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.POINTER(C.c_size_t)]),
    "mola_icp_set_local_shard_range_host": (C.c_int, [_H, _FP, _FP, _FP, C.c_size_t, C.c_size_t, C.c_size_t, C.POINTER(C.c_size_t)]),
    "mola_icp_set_local_shard_range_device": (C.c_int, [_H, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_size_t, C.c_size_t, C.POINTER(C.c_size_t)]),
    "mola_icp_local_shard_indices": (C.c_int, [_H, C.POINTER(C.c_int32)]),
    "mola_icp_shard_reach_box": (C.c_int, [_H, _DP, C.c_double, _DP, _DP]),
    "mola_icp_set_map_slab_host": (C.c_int, [_H, _FP, _FP, _FP, C.c_size_t, _DP, _DP, C.POINTER(C.c_size_t)]),
    "mola_icp_set_map_slab_device": (C.c_int, [_H, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, _DP, _DP, C.POINTER(C.c_size_t)]),
    "mola_icp_last_error": (C.c_char_p, []),
    "mola_icp_status_string": (C.c_char_p, [C.c_int]),
    "mola_icp_device_count": (C.c_int, [C.POINTER(C.c_int)]),
    "mola_icp_debug_reload_env": (C.c_int, []),
    "mola_icp_set_wait_policy": (C.c_int, [C.c_int]),
    "mola_icp_set_thread_priority": (C.c_int, [C.c_int]),
    "mola_icp_get_thread_priority": (C.c_int, [C.POINTER(C.c_int)]),
    "mola_icp_get_wait_policy": (C.c_int, [C.POINTER(C.c_int)]),
    "mola_icp_params_default": (C.c_int, [C.POINTER(CParams)]),
    "mola_icp_params_from_yaml": (C.c_int, [C.c_char_p, C.POINTER(CParams)]),
    "mola_icp_params_from_yaml_file": (C.c_int, [C.c_char_p, C.c_char_p, C.c_char_p, C.POINTER(CParams)]),
    "mola_icp_params_compose": (C.c_int, [C.POINTER(CParams), C.POINTER(CParams), C.POINTER(CParams)]),
    "mola_icp_create": (C.c_int, [C.c_int, C.POINTER(_H)]),
    "mola_icp_destroy": (C.c_int, [_H]),
    "mola_icp_set_stream": (C.c_int, [_H, C.c_void_p]),
    "mola_icp_set_allreduce": (C.c_int, [_H, ALLREDUCE_FN, C.c_void_p]),
    "mola_icp_comm_set_library": (C.c_int, [C.c_char_p]),
    "mola_icp_comm_unique_id": (C.c_int, [C.POINTER(C.c_uint8)]),
    "mola_icp_comm_init": (C.c_int, [_H, C.POINTER(C.c_uint8), C.c_int, C.c_int]),
    "mola_icp_comm_nranks": (C.c_int, [_H, C.POINTER(C.c_int)]),
    "mola_icp_comm_destroy": (C.c_int, [_H]),
    "mola_icp_local_comm_create": (C.c_int, [C.c_char_p, C.c_int, C.c_int, C.c_double, C.POINTER(C.c_void_p)]),
    "mola_icp_local_comm_allreduce": (C.c_int, [C.c_void_p, C.POINTER(C.c_double), C.c_int]),
    "mola_icp_local_comm_nranks": (C.c_int, [C.c_void_p, C.POINTER(C.c_int)]),
    "mola_icp_local_comm_abort": (C.c_int, [C.c_void_p]),
    "mola_icp_local_comm_destroy": (C.c_int, [C.c_void_p]),
    "mola_icp_comm_attach_local": (C.c_int, [_H, C.c_void_p]),
    "mola_icp_align": (C.c_int, [_H, _FP, _FP, _FP, C.c_size_t, _FP, _FP, _FP, C.c_size_t, _DP,
                                 C.POINTER(CParams), C.POINTER(CResult)]),
    "mola_icp_align_batch": (C.c_int, [_H, C.c_size_t, C.POINTER(_FP), C.POINTER(_FP), C.POINTER(_FP),
                                       C.POINTER(C.c_size_t), C.POINTER(_FP), C.POINTER(_FP), C.POINTER(_FP),
                                       C.POINTER(C.c_size_t), _DP, C.POINTER(CParams), C.POINTER(CResult)]),
    "mola_icp_pool_create": (C.c_int, [C.POINTER(C.c_int), C.c_int, C.POINTER(_H)]),
    "mola_icp_pool_destroy": (C.c_int, [_H]),
    "mola_icp_pool_size": (C.c_int, [_H, C.POINTER(C.c_int)]),
    "mola_icp_pool_handle": (C.c_int, [_H, C.c_int, C.POINTER(_H)]),
    "mola_icp_pool_assignment": (C.c_int, [C.c_size_t, C.c_int, C.POINTER(C.c_int)]),
    "mola_icp_pool_last_shares": (C.c_int, [_H, C.POINTER(C.c_size_t), C.c_int]),
    "mola_icp_pool_align_batch": (C.c_int, [_H, C.c_size_t, C.POINTER(_FP), C.POINTER(_FP), C.POINTER(_FP),
                                            C.POINTER(C.c_size_t), C.POINTER(_FP), C.POINTER(_FP), C.POINTER(_FP),
                                            C.POINTER(C.c_size_t), _DP, C.POINTER(CParams), C.POINTER(CResult)]),
    "mola_icp_align_multi_init": (C.c_int, [_H, _FP, _FP, _FP, C.c_size_t, _FP, _FP, _FP, C.c_size_t, C.c_size_t, _DP,
                                            C.POINTER(CParams), C.POINTER(CResult), C.POINTER(CResult),
                                            C.POINTER(C.c_int)]),
    "mola_icp_cloud_put": (C.c_int, [_H, C.c_uint64, _FP, _FP, _FP, C.c_size_t]),
    "mola_icp_cloud_drop": (C.c_int, [_H, C.c_uint64]),
    "mola_icp_cloud_count": (C.c_int, [_H, C.POINTER(C.c_size_t), C.POINTER(C.c_size_t)]),
    "mola_icp_device_pool_trim": (C.c_int, [C.c_int, C.c_size_t, C.POINTER(C.c_size_t)]),
    "mola_icp_align_cached": (C.c_int, [_H, C.c_uint64, C.c_uint64, _DP, C.POINTER(CParams), C.POINTER(CResult)]),
    "mola_icp_align_cached_put": (C.c_int, [_H, C.c_uint64, C.c_uint64, _FP, _FP, _FP, C.c_size_t, _DP, C.POINTER(CParams), C.POINTER(CResult),
                                            C.POINTER(C.c_int)]),
    "mola_icp_voxel_downsample": (C.c_int, [_H, _FP, _FP, _FP, C.c_size_t, C.c_double, _FP, _FP, _FP, C.c_size_t,
                                            C.POINTER(C.c_size_t)]),
    "mola_icp_set_map_host": (C.c_int, [_H, _FP, _FP, _FP, C.c_size_t]),
    "mola_icp_set_map_device": (C.c_int, [_H, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t]),
    "mola_icp_set_local_host": (C.c_int, [_H, _FP, _FP, _FP, C.c_size_t]),
    "mola_icp_set_local_device": (C.c_int, [_H, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t]),
    "mola_icp_set_global_sizes": (C.c_int, [_H, C.c_uint64, C.c_uint64]),
    "mola_icp_align_resident": (C.c_int, [_H, _DP, C.POINTER(CParams), C.POINTER(CResult)]),
    "mola_icp_match": (C.c_int, [_H, _DP, C.c_double, C.c_int, C.POINTER(C.c_int32), _FP, C.POINTER(C.c_uint64)]),
    "mola_icp_accumulate": (C.c_int, [_H, C.POINTER(CParams), _DP, C.c_int, _DP, _DP, C.c_int, _DP]),
    "mola_icp_match_planes": (C.c_int, [_H, _DP, C.POINTER(CParams), C.POINTER(C.c_uint8), _DP, _DP, C.POINTER(C.c_int32),
                                        C.POINTER(C.c_uint64)]),
    "mola_icp_accumulate_planes": (C.c_int, [_H, _DP]),
    "mola_icp_solve_gauss_newton_planes": (C.c_int, [_DP, _DP, C.c_uint32, _DP, _DP, C.POINTER(C.c_uint32)]),
    "mola_icp_mixed_form": (C.c_int, [_DP, _DP, _DP]),
    "mola_icp_solve_horn": (C.c_int, [_DP, _DP, _DP, _DP]),
    "mola_icp_stall_deltas": (C.c_int, [_DP, _DP, _DP, _DP]),
    "mola_icp_se3_log": (C.c_int, [_DP, _DP]),
    "mola_icp_pose_from_xyzypr": (C.c_int, [_DP, _DP]),
    "mola_icp_pose_to_xyzypr": (C.c_int, [_DP, _DP]),
    "mola_icp_run_loop": (C.c_int, [C.POINTER(CStageCallbacks), _DP, C.POINTER(CParams), C.POINTER(CResult)]),
    "mola_icp_run_loop_batch": (C.c_int, [C.POINTER(CStageCallbacks), C.c_size_t, _DP, C.POINTER(CParams), C.POINTER(CResult)]),
    "mola_lo_params_default": (C.c_int, [C.POINTER(CLoParams)]),
    "mola_lo_params_from_yaml_file": (C.c_int, [C.c_char_p, C.c_char_p, C.POINTER(CLoParams)]),
    "mola_lo_create": (C.c_int, [_H, LO_ALIGN_FN, C.c_void_p, C.POINTER(CLoParams), C.POINTER(_H)]),
    "mola_lo_destroy": (C.c_int, [_H]),
    "mola_lo_reset": (C.c_int, [_H]),
    "mola_lo_process_scan": (C.c_int, [_H, C.c_double, _FP, _FP, _FP, C.c_size_t, C.POINTER(CLoStep)]),
    "mola_lo_select_checks": (C.c_int, [C.POINTER(CLoParams), C.POINTER(CLoKfCandidate), C.c_size_t, C.POINTER(C.c_uint64),
                                        C.c_size_t, C.POINTER(C.c_size_t), C.POINTER(C.c_uint64), C.POINTER(C.c_int)]),
    "mola_lo_montecarlo_guesses": (C.c_int, [_DP, C.c_double, C.c_uint32, C.c_uint64, _DP, _DP]),
    "mola_lo_check_nonadjacent": (C.c_int, [_H, LO_ALIGN_FN, C.c_void_p, C.POINTER(CLoParams), C.c_int, _FP, _FP, _FP, C.c_size_t,
                                            _FP, _FP, _FP, C.c_size_t, _DP, C.c_uint64, C.POINTER(CLoCheckResult)]),
}

_lib = None


def _preload_shared_hip_runtime() -> None:
    """PyTorch wheels bundle their own libamdhip64/libhsa-runtime64.  A process may hold only ONE HIP
    runtime (device pointers and streams are exchanged with torch: ICP.set_map(tensor),
    set_stream), so when torch is installed its runtime is loaded first and this library's
    `libamdhip64.so.7` dependency resolves to it by SONAME.  Without torch the system ROCm is used."""
    import importlib.util
    import sys
    if "torch" in sys.modules or os.environ.get("MOLA_ICP_SYSTEM_HIP"):
        return
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        spec = None
    if spec is None or not spec.origin:
        return
    libdir = os.path.join(os.path.dirname(spec.origin), "lib")
    for name in ("libhsa-runtime64.so", "libamdhip64.so"):
        path = os.path.join(libdir, name)
        if os.path.exists(path):
            try:
                C.CDLL(path, mode=C.RTLD_GLOBAL)
            except OSError:
                return


class IcpError(RuntimeError):
    """A MOLA_ICP_E_* status, carrying the library's message (the reference's own
    error convention is exceptions: src/LidarOdometry.cpp:70-75, 860-861)."""

    def __init__(self, status: int, message: str):
        super().__init__(f"[{status}] {message}")
        self.status = status
        self.message = message


def lib() -> C.CDLL:
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(or `make -C mola-fe-lidar_amd/csrc`). There is no fallback implementation.")
        _preload_shared_hip_runtime()
        L = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            f = getattr(L, name)
            f.restype = res
            f.argtypes = args
        if L.mola_icp_abi_version() != ABI_VERSION:
            raise ImportError("libmola_icp_amd.so ABI version mismatch")
        _lib = L
    return _lib


def check(status: int) -> None:
    if status != OK:
        msg = lib().mola_icp_last_error().decode("utf-8", "replace")
        raise IcpError(status, msg or lib().mola_icp_status_string(status).decode())
