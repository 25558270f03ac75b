"""ctypes wrapper of oracle/_build/libicp_oracle.so -- TEST INFRASTRUCTURE ONLY.

Importable only from tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg
(see the header of oracle/icp_oracle.c).  PARITY UNPINNED: the reference ships no
golden vectors for this path; see DESIGN.md."""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "_build", "libicp_oracle.so")
NACC = 24


class OParams(C.Structure):
    _fields_ = [
        ("max_iterations", C.c_uint32), ("min_abs_step_trans", C.c_double), ("min_abs_step_rot", C.c_double),
        ("use_scale_outlier_detector", C.c_int32), ("scale_outlier_threshold", C.c_double),
        ("use_robust_kernel", C.c_int32), ("robust_kernel_param", C.c_double), ("robust_kernel_scale", C.c_double),
        ("matcher_threshold", C.c_double), ("run_from_iteration", C.c_uint32), ("run_up_to_iteration", C.c_uint32),
        ("quality_threshold", C.c_double), ("fixed_iterations", C.c_int32), ("use_kdtree", C.c_int32),
    ]


class OResult(C.Structure):
    _fields_ = [
        ("T", C.c_double * 16), ("quality", C.c_double), ("n_iterations", C.c_uint32), ("termination", C.c_uint32),
        ("n_pairs", C.c_uint64), ("rmse", C.c_double), ("kdtree_build_s", C.c_double), ("iter_s", C.c_double),
    ]


class OReadings(C.Structure):
    """alternative readings of the [EXT]-recalled mp2p_icp semantics (orc_readings in icp_oracle.c); all 0 = default"""
    _fields_ = [("stall_max_abs", C.c_int32), ("quality_denominator", C.c_int32), ("outlier_single_pass", C.c_int32),
                ("gn_right_perturbation", C.c_int32), ("p2pl_all_inside_gate", C.c_int32)]


def set_readings(**kw):
    """set_readings() restores the defaults; set_readings(stall_max_abs=1) switches one reading (process-global)."""
    r = OReadings()
    for k, v in kw.items():
        setattr(r, k, int(v))
    lib().orc_set_readings(C.byref(r))


def build(force: bool = False) -> str:
    src = os.path.join(_HERE, "icp_oracle.c")
    stale = (not os.path.exists(LIB_PATH)) or (os.path.exists(src) and os.path.getmtime(src) > os.path.getmtime(LIB_PATH))
    if force or stale:
        subprocess.check_call(["make", "-C", _HERE, "-s"])
    return LIB_PATH


def use_native() -> str:
    """bench.py's cpu_baseline legs: build the checker with -march=native on THIS machine (oracle/Makefile `native`) and
    switch to it; the portable x86-64-v3 build stays in use if that fails.  Returns the flags in effect."""
    global _lib, LIB_PATH
    native = os.path.join(_HERE, "_build", "libicp_oracle_native.so")
    try:
        subprocess.check_call(["make", "-C", _HERE, "-s", "-B", "native"], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        C.CDLL(native)          # loads on this CPU?
    except Exception:
        return "-O3 -march=x86-64-v3 (portable build; the -march=native build failed here)"
    LIB_PATH = native
    _lib = None
    return "-O3 -march=native"


_lib = None
_FP = C.POINTER(C.c_float)
_DP = C.POINTER(C.c_double)
_IP = C.POINTER(C.c_int32)


def lib() -> C.CDLL:
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            build()
        L = C.CDLL(LIB_PATH)
        L.orc_kdtree_build.restype = C.c_void_p
        L.orc_kdtree_build.argtypes = [_FP, _FP, _FP, C.c_size_t]
        L.orc_kdtree_free.argtypes = [C.c_void_p]
        L.orc_kdtree_nn.argtypes = [C.c_void_p, _FP, _FP, _FP, C.c_size_t, _IP, _FP]
        L.orc_nn_brute.argtypes = [_FP, _FP, _FP, C.c_size_t, _FP, _FP, _FP, C.c_size_t, _IP, _FP]
        L.orc_transform_f32.argtypes = [_DP, _FP, _FP, _FP, C.c_size_t, _FP, _FP, _FP]
        L.orc_match.restype = C.c_size_t
        L.orc_match.argtypes = [_FP, _FP, _FP, C.c_size_t, C.c_void_p, _FP, _FP, _FP, C.c_size_t, _DP, C.c_double,
                                _IP, _FP]
        L.orc_accumulate.argtypes = [_FP, _FP, _FP, _FP, _FP, _FP, _IP, _FP, C.c_size_t, C.POINTER(OParams), _DP,
                                     C.c_int, _DP, _DP, C.POINTER(C.c_uint8), _DP]
        L.orc_horn.restype = C.c_int
        L.orc_horn.argtypes = [_DP, _DP, _DP, _DP]
        L.orc_solve_pairs.restype = C.c_int
        L.orc_solve_pairs.argtypes = [_FP, _FP, _FP, _FP, _FP, _FP, _IP, _FP, C.c_size_t, C.POINTER(OParams), _DP,
                                      _DP, _DP]
        L.orc_se3_log.argtypes = [_DP, _DP]
        L.orc_stall_deltas.argtypes = [_DP, _DP, _DP, _DP]
        L.orc_pose_from_xyzypr.argtypes = [_DP, _DP]
        L.orc_pose_to_xyzypr.argtypes = [_DP, _DP]
        L.orc_align.restype = C.c_int
        L.orc_align.argtypes = [_FP, _FP, _FP, C.c_size_t, _FP, _FP, _FP, C.c_size_t, _DP, C.POINTER(OParams),
                                C.POINTER(OResult), _DP]
        L.orc_sizeof_params.restype = C.c_size_t
        L.orc_sizeof_result.restype = C.c_size_t
        assert L.orc_sizeof_params() == C.sizeof(OParams) and L.orc_sizeof_result() == C.sizeof(OResult)
        _lib = L
    return _lib


def _f(a):
    return a.ctypes.data_as(_FP)


def _d(a):
    return a.ctypes.data_as(_DP)


def _rows(pc):
    a = np.ascontiguousarray(pc, dtype=np.float32)
    assert a.ndim == 2 and a.shape[0] == 3
    return a[0], a[1], a[2], a.shape[1]


def params(max_iterations=40, min_abs_step_trans=5e-5, min_abs_step_rot=1e-5, use_scale_outlier_detector=False,
           scale_outlier_threshold=1.1, use_robust_kernel=False, robust_kernel_param=np.deg2rad(0.1),
           robust_kernel_scale=400.0, matcher_threshold=1.0, run_from_iteration=0, run_up_to_iteration=0,
           quality_threshold=0.10, fixed_iterations=False, use_kdtree=True) -> OParams:
    return OParams(max_iterations, min_abs_step_trans, min_abs_step_rot, int(use_scale_outlier_detector),
                   scale_outlier_threshold, int(use_robust_kernel), robust_kernel_param, robust_kernel_scale,
                   matcher_threshold, run_from_iteration, run_up_to_iteration, quality_threshold,
                   int(fixed_iterations), int(use_kdtree))


def params_from_product(p, use_kdtree=True) -> OParams:
    """Same settings as a product `Parameters` (mola_icp_params)."""
    return params(p.max_iterations, p.min_abs_step_trans, p.min_abs_step_rot, bool(p.use_scale_outlier_detector),
                  p.scale_outlier_threshold, bool(p.use_robust_kernel), p.robust_kernel_param, p.robust_kernel_scale,
                  p.matcher_threshold, p.run_from_iteration, p.run_up_to_iteration, p.quality_threshold,
                  bool(p.fixed_iterations), use_kdtree)


def transform(T, local):
    lx, ly, lz, n = _rows(local)
    out = np.empty((3, n), dtype=np.float32)
    T = np.ascontiguousarray(T, dtype=np.float64).reshape(16)
    lib().orc_transform_f32(_d(T), _f(lx), _f(ly), _f(lz), n, _f(out[0]), _f(out[1]), _f(out[2]))
    return out


def nn_brute(map_pc, q):
    gx, gy, gz, M = _rows(map_pc)
    qx, qy, qz, N = _rows(q)
    idx = np.empty(N, dtype=np.int32)
    d2 = np.empty(N, dtype=np.float32)
    lib().orc_nn_brute(_f(gx), _f(gy), _f(gz), M, _f(qx), _f(qy), _f(qz), N, idx.ctypes.data_as(_IP), _f(d2))
    return idx, d2


class KdTree:
    def __init__(self, map_pc):
        self.gx, self.gy, self.gz, self.M = _rows(map_pc)
        self.h = lib().orc_kdtree_build(_f(self.gx), _f(self.gy), _f(self.gz), self.M)

    def nn(self, q):
        qx, qy, qz, N = _rows(q)
        idx = np.empty(N, dtype=np.int32)
        d2 = np.empty(N, dtype=np.float32)
        lib().orc_kdtree_nn(self.h, _f(qx), _f(qy), _f(qz), N, idx.ctypes.data_as(_IP), _f(d2))
        return idx, d2

    def __del__(self):
        if getattr(self, "h", None):
            lib().orc_kdtree_free(self.h)
            self.h = None


def match(map_pc, local, T, threshold, tree: "KdTree | None" = None):
    """transform + NN + gate: returns (idx with -1 where rejected, d2, n_pairs)."""
    gx, gy, gz, M = _rows(map_pc)
    lx, ly, lz, N = _rows(local)
    idx = np.empty(N, dtype=np.int32)
    d2 = np.empty(N, dtype=np.float32)
    T = np.ascontiguousarray(T, dtype=np.float64).reshape(16)
    n = lib().orc_match(_f(gx), _f(gy), _f(gz), M, tree.h if tree else None, _f(lx), _f(ly), _f(lz), N, _d(T),
                        float(threshold), idx.ctypes.data_as(_IP), _f(d2))
    return idx, d2, int(n)


def accumulate(map_pc, local, idx, d2, p: OParams, Tcur, stage=0, cl=None, cg=None, outlier=None):
    gx, gy, gz, M = _rows(map_pc)
    lx, ly, lz, N = _rows(local)
    acc = np.empty(NACC)
    T = np.ascontiguousarray(Tcur, dtype=np.float64).reshape(16)
    idx = np.ascontiguousarray(idx, dtype=np.int32)
    d2 = np.ascontiguousarray(d2, dtype=np.float32)
    clp = _d(np.ascontiguousarray(cl, dtype=np.float64)) if cl is not None else None
    cgp = _d(np.ascontiguousarray(cg, dtype=np.float64)) if cg is not None else None
    op = outlier.ctypes.data_as(C.POINTER(C.c_uint8)) if outlier is not None else None
    lib().orc_accumulate(_f(lx), _f(ly), _f(lz), _f(gx), _f(gy), _f(gz), idx.ctypes.data_as(_IP), _f(d2), N,
                         C.byref(p), _d(T), stage, clp, cgp, op, _d(acc))
    return acc


def solve_pairs(map_pc, local, idx, d2, p: OParams, Tcur):
    """one solver invocation (weights + accumulation + Horn) -> (T_new, acc)."""
    gx, gy, gz, M = _rows(map_pc)
    lx, ly, lz, N = _rows(local)
    T = np.ascontiguousarray(Tcur, dtype=np.float64).reshape(16)
    Tn = np.empty(16)
    acc = np.empty(NACC)
    idx = np.ascontiguousarray(idx, dtype=np.int32)
    d2 = np.ascontiguousarray(d2, dtype=np.float32)
    rc = lib().orc_solve_pairs(_f(lx), _f(ly), _f(lz), _f(gx), _f(gy), _f(gz), idx.ctypes.data_as(_IP), _f(d2), N,
                               C.byref(p), _d(T), _d(Tn), _d(acc))
    if rc:
        raise ValueError(f"orc_solve_pairs failed: {rc}")
    return Tn.reshape(4, 4), acc


def horn(acc, cl=None, cg=None):
    acc = np.ascontiguousarray(acc, dtype=np.float64)
    T = np.empty(16)
    clp = _d(np.ascontiguousarray(cl, dtype=np.float64)) if cl is not None else None
    cgp = _d(np.ascontiguousarray(cg, dtype=np.float64)) if cg is not None else None
    rc = lib().orc_horn(_d(acc), clp, cgp, _d(T))
    if rc:
        raise ValueError(f"orc_horn failed: {rc}")
    return T.reshape(4, 4)


def se3_log(T):
    T = np.ascontiguousarray(T, dtype=np.float64).reshape(16)
    o = np.empty(6)
    lib().orc_se3_log(_d(T), _d(o))
    return o


def stall_deltas(T, Tprev):
    a = np.ascontiguousarray(T, dtype=np.float64).reshape(16)
    b = np.ascontiguousarray(Tprev, dtype=np.float64).reshape(16)
    x, r = C.c_double(), C.c_double()
    lib().orc_stall_deltas(_d(a), _d(b), C.byref(x), C.byref(r))
    return x.value, r.value


def pose_from_xyzypr(p):
    p = np.ascontiguousarray(p, dtype=np.float64)
    T = np.empty(16)
    lib().orc_pose_from_xyzypr(_d(p), _d(T))
    return T.reshape(4, 4)


def pose_to_xyzypr(T):
    T = np.ascontiguousarray(T, dtype=np.float64).reshape(16)
    p = np.empty(6)
    lib().orc_pose_to_xyzypr(_d(T), _d(p))
    return p


def align(map_pc, local, T_init, p: OParams, trace: bool = False):
    gx, gy, gz, M = _rows(map_pc)
    lx, ly, lz, N = _rows(local)
    T = np.ascontiguousarray(T_init, dtype=np.float64).reshape(16)
    res = OResult()
    tr = np.zeros((max(1, p.max_iterations), 16)) if trace else None
    rc = lib().orc_align(_f(gx), _f(gy), _f(gz), M, _f(lx), _f(ly), _f(lz), N, _d(T), C.byref(p), C.byref(res),
                         _d(tr) if trace else None)
    assert rc == 0
    out = dict(T=np.array(res.T).reshape(4, 4), quality=res.quality, n_iterations=res.n_iterations,
               termination=res.termination, n_pairs=res.n_pairs, rmse=res.rmse, kdtree_build_s=res.kdtree_build_s,
               iter_s=res.iter_s)
    if trace:
        out["trace"] = tr[:res.n_iterations].reshape(-1, 4, 4)
    return out


def pose_error(T, T_ref):
    """(rotation geodesic angle [rad], translation distance [m]) between two poses."""
    T, T_ref = np.asarray(T), np.asarray(T_ref)
    dR = T_ref[:3, :3].T @ T[:3, :3]
    c = np.clip((np.trace(dR) - 1) / 2, -1, 1)
    return float(np.arccos(c)), float(np.linalg.norm(T[:3, 3] - T_ref[:3, 3]))


# ---------------------------------------------------------------- row f3: point-to-plane + Gauss-Newton
def _bind_p2pl():
    L = lib()
    if getattr(L, "_p2pl_bound", False):
        return L
    U8 = C.POINTER(C.c_uint8)
    L.orc_match_point2plane.restype = C.c_size_t
    L.orc_match_point2plane.argtypes = [_FP, _FP, _FP, C.c_size_t, C.c_void_p, _FP, _FP, _FP, C.c_size_t, _DP, C.c_double,
                                        C.c_double, C.c_int, U8, _DP, _DP, _IP]
    L.orc_solve_gauss_newton.restype = C.c_int
    L.orc_solve_gauss_newton.argtypes = [_FP, _FP, _FP, C.c_size_t, U8, _DP, _DP, _DP, C.c_uint32, _DP, _DP,
                                         C.POINTER(C.c_uint32)]
    L.orc_align_p2pl.restype = C.c_int
    L.orc_align_p2pl.argtypes = [_FP, _FP, _FP, C.c_size_t, _FP, _FP, _FP, C.c_size_t, _DP, C.POINTER(OParams), C.c_double,
                                 C.c_int, C.c_uint32, C.POINTER(OResult)]
    L._p2pl_bound = True
    return L


def match_point2plane(map_pc, local, T, threshold, plane_eigen_threshold, knn, tree: "KdTree | None" = None):
    """-> (valid uint8[N], centroid (N,3), normal (N,3), knn_idx (N,knn), n_pairs)"""
    L = _bind_p2pl()
    gx, gy, gz, M = _rows(map_pc)
    lx, ly, lz, N = _rows(local)
    T = np.ascontiguousarray(T, dtype=np.float64).reshape(16)
    valid = np.zeros(max(1, N), np.uint8)
    cen = np.zeros((max(1, N), 3))
    nor = np.zeros((max(1, N), 3))
    kidx = np.full((max(1, N), knn), -1, np.int32)
    n = L.orc_match_point2plane(_f(gx), _f(gy), _f(gz), M, tree.h if tree else None, _f(lx), _f(ly), _f(lz), N, _d(T),
                                float(threshold), float(plane_eigen_threshold), int(knn),
                                valid.ctypes.data_as(C.POINTER(C.c_uint8)), _d(cen), _d(nor), kidx.ctypes.data_as(_IP))
    return valid[:N], cen[:N], nor[:N], kidx[:N], int(n)


def solve_gauss_newton(local, valid, centroid, normal, Tcur, max_iters=20):
    L = _bind_p2pl()
    lx, ly, lz, N = _rows(local)
    T = np.ascontiguousarray(Tcur, dtype=np.float64).reshape(16)
    Tn = np.empty(16)
    cost = C.c_double()
    its = C.c_uint32()
    v = np.ascontiguousarray(valid, dtype=np.uint8)
    c = np.ascontiguousarray(centroid, dtype=np.float64)
    nn = np.ascontiguousarray(normal, dtype=np.float64)
    rc = L.orc_solve_gauss_newton(_f(lx), _f(ly), _f(lz), N, v.ctypes.data_as(C.POINTER(C.c_uint8)), _d(c), _d(nn), _d(T),
                                  int(max_iters), _d(Tn), C.byref(cost), C.byref(its))
    if rc:
        raise ValueError(f"orc_solve_gauss_newton failed: {rc}")
    return Tn.reshape(4, 4), cost.value, its.value


def align_mixed(map_pc, local, T_init, p: OParams, plane_threshold=0.7, plane_eigen_threshold=0.07, knn=6, solver_max_iters=20):
    """both matchers active in every iteration (point-to-point at p.matcher_threshold, point-to-plane at plane_threshold), one
    Gauss-Newton solve over the sum of their costs"""
    L = _bind_p2pl()
    if not getattr(L, "_mixed_bound", False):
        L.orc_align_mixed.restype = C.c_int
        L.orc_align_mixed.argtypes = [_FP, _FP, _FP, C.c_size_t, _FP, _FP, _FP, C.c_size_t, _DP, C.POINTER(OParams), C.c_double,
                                      C.c_double, C.c_int, C.c_uint32, C.POINTER(OResult)]
        L._mixed_bound = True
    gx, gy, gz, M = _rows(map_pc)
    lx, ly, lz, N = _rows(local)
    T = np.ascontiguousarray(T_init, dtype=np.float64).reshape(16)
    res = OResult()
    rc = L.orc_align_mixed(_f(gx), _f(gy), _f(gz), M, _f(lx), _f(ly), _f(lz), N, _d(T), C.byref(p), float(plane_threshold),
                           float(plane_eigen_threshold), int(knn), int(solver_max_iters), C.byref(res))
    assert rc == 0
    return dict(T=np.array(res.T).reshape(4, 4), quality=res.quality, n_iterations=res.n_iterations,
                termination=res.termination, n_pairs=res.n_pairs, rmse=res.rmse, iter_s=res.iter_s)


def align_p2pl(map_pc, local, T_init, p: OParams, plane_eigen_threshold=0.07, knn=6, solver_max_iters=20):
    L = _bind_p2pl()
    gx, gy, gz, M = _rows(map_pc)
    lx, ly, lz, N = _rows(local)
    T = np.ascontiguousarray(T_init, dtype=np.float64).reshape(16)
    res = OResult()
    rc = L.orc_align_p2pl(_f(gx), _f(gy), _f(gz), M, _f(lx), _f(ly), _f(lz), N, _d(T), C.byref(p),
                          float(plane_eigen_threshold), int(knn), int(solver_max_iters), C.byref(res))
    assert rc == 0
    return dict(T=np.array(res.T).reshape(4, 4), quality=res.quality, n_iterations=res.n_iterations,
                termination=res.termination, n_pairs=res.n_pairs, rmse=res.rmse, iter_s=res.iter_s)
