"""Host-side mirror of the `LidarOdometry` front-end logic around the ICP:
`onNewObservation` -> time gate -> constant-velocity guess -> `run_one_icp` -> twist -> keyframe decision
(src/LidarOdometry.cpp:190-514).  Marshalling only: the logic is csrc/lidar_odometry_core.cpp."""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass

import time

import numpy as np

from . import _lib as L
from .icp import ICP, Parameters, Results, _fp, _soa


@dataclass
class Step:
    status: int
    used_with_vel_params: bool
    dt: float
    rel_pose: np.ndarray
    twist: np.ndarray
    dist_since_last_kf: float
    rot_since_last_kf: float
    keyframe_created: bool
    kf_factor: tuple | None          # (from_kf, to_kf, 4x4 pose) when a FactorRelativePose3 would be emitted
    reference_kf: int
    accum_since_last_kf: np.ndarray
    icp: Results | None
    ms_native: float = 0.0           # wall time of the mola_lo_process_scan call itself (this wrapper's marshalling excluded)


class LidarOdometryParams:
    """the scalar front-end parameters + the two ICP parameter sets the odometry path uses
    (include/mola-fe-lidar/LidarOdometry.h:52-102, src/LidarOdometry.cpp:105-128)"""

    def __init__(self):
        self.c = L.CLoParams()
        L.check(L.lib().mola_lo_params_default(C.byref(self.c)))

    @classmethod
    def load_from_file(cls, path: str, mola_dir: str | None = None) -> "LidarOdometryParams":
        p = cls()
        L.check(L.lib().mola_lo_params_from_yaml_file(path.encode(), mola_dir.encode() if mola_dir else None,
                                                      C.byref(p.c)))
        return p

    _SCALARS = ("min_time_between_scans", "min_dist_xyz_between_keyframes", "min_rotation_between_keyframes",
                "min_icp_goodness", "min_icp_goodness_lc", "min_dist_to_matching", "max_dist_to_matching",
                "max_dist_to_loop_closure", "loop_closure_montecarlo_samples", "max_nearby_align_checks",
                "min_topo_dist_to_consider_loopclosure", "max_kfs_local_graph")

    def set_icp(self, with_vel: Parameters, without_vel: Parameters | None = None,
                loop_closure: Parameters | None = None):
        """the three ICP cases of `Parameters::icp` (LidarOdometry.h:96-102)"""
        C.memmove(C.byref(self.c.icp_with_vel), C.byref(with_vel.c), C.sizeof(L.CParams))
        C.memmove(C.byref(self.c.icp_without_vel), C.byref((without_vel or with_vel).c), C.sizeof(L.CParams))
        C.memmove(C.byref(self.c.icp_loop_closure), C.byref((loop_closure or without_vel or with_vel).c),
                  C.sizeof(L.CParams))

    def icp_case(self, name: str) -> Parameters:
        """a copy of one case: "with_vel" (AlignKind::LidarOdometry), "without_vel" (NearbyAlign), "loop_closure"""
        p = Parameters()
        C.memmove(C.byref(p.c), C.byref(getattr(self.c, "icp_" + name)), C.sizeof(L.CParams))
        return p

    def __getattr__(self, name):
        c = object.__getattribute__(self, "c")
        if name in LidarOdometryParams._SCALARS:
            return getattr(c, name)
        raise AttributeError(name)

    def __setattr__(self, name, value):
        if name in LidarOdometryParams._SCALARS:
            setattr(self.c, name, value)
        else:
            object.__setattr__(self, name, value)


def _align_callback(align_fn):
    """wraps `align_fn(from(3,M), to(3,N), T0 4x4, Parameters) -> (T, quality, nIterations, terminationReason)` as a
    mola_lo_align_fn"""
    def _cb(user, fx, fy, fz, M, tx, ty, tz, N, T0, pp, out):
        try:
            f = np.stack([np.ctypeslib.as_array(a, shape=(M,)) for a in (fx, fy, fz)]) if M else np.zeros((3, 0), np.float32)
            t = np.stack([np.ctypeslib.as_array(a, shape=(N,)) for a in (tx, ty, tz)]) if N else np.zeros((3, 0), np.float32)
            p = Parameters()
            C.memmove(C.byref(p.c), pp, C.sizeof(L.CParams))
            T, q, nit, term = align_fn(f, t, np.ctypeslib.as_array(T0, shape=(16,)).reshape(4, 4).copy(), p)
            out[0].T[:] = list(np.asarray(T, dtype=np.float64).reshape(16))
            out[0].quality, out[0].n_iterations, out[0].termination = float(q), int(nit), int(term)
            return 0
        except Exception:
            import traceback
            traceback.print_exc()
            return L.E_INTERNAL
    return L.LO_ALIGN_FN(_cb)


def select_checks(params: LidarOdometryParams, kfs):
    """`checkForNearbyKFs`' selection (src/LidarOdometry.cpp:572-599, 700-729).  kfs: iterable of
    (kf_id, eucl_dist, topo_dist, already_checked).  Returns (nearby ids in sending order, loop-closure id or None)."""
    kfs = list(kfs)
    arr = (L.CLoKfCandidate * max(1, len(kfs)))()
    for i, (kid, d, topo, chk) in enumerate(kfs):
        arr[i].kf_id, arr[i].eucl_dist, arr[i].topo_dist, arr[i].already_checked = int(kid), float(d), int(topo), int(bool(chk))
    ids = (C.c_uint64 * max(1, len(kfs)))()
    n, lc, has = C.c_size_t(0), C.c_uint64(0), C.c_int(0)
    L.check(L.lib().mola_lo_select_checks(C.byref(params.c), arr, len(kfs), ids, len(kfs), C.byref(n), C.byref(lc), C.byref(has)))
    return [int(ids[i]) for i in range(n.value)], (int(lc.value) if has.value else None)


def montecarlo_guesses(init_xyzypr, max_dist_to_loop_closure: float, n_samples: int, seed: int):
    """the loop-closure Monte-Carlo's perturbed guesses (cpp:767-783): (n,6) xyzypr and (n,4,4) poses"""
    g0 = np.ascontiguousarray(init_xyzypr, dtype=np.float64).reshape(6)
    g6 = np.zeros((max(1, n_samples), 6))
    gT = np.zeros((max(1, n_samples), 16))
    L.check(L.lib().mola_lo_montecarlo_guesses(g0.ctypes.data_as(L._DP), float(max_dist_to_loop_closure), int(n_samples),
                                               int(seed), g6.ctypes.data_as(L._DP), gT.ctypes.data_as(L._DP)))
    return g6[:n_samples], gT[:n_samples].reshape(-1, 4, 4)


@dataclass
class CheckResult:
    icp: Results
    best_guess: int
    n_attempts: int
    init_guess_used: np.ndarray
    correction_percent: float
    edge_accepted: bool


def check_nonadjacent(params: LidarOdometryParams, from_pc, to_pc, init_xyzypr, is_loop_closure: bool, seed: int = 0,
                      icp: ICP | None = None, align_fn=None) -> CheckResult:
    """`doCheckForNonAdjacentKFs` (cpp:743-848) without the back-end calls"""
    assert (icp is None) != (align_fn is None)
    cb = _align_callback(align_fn) if align_fn is not None else L.LO_ALIGN_FN()
    fx, fy, fz, M = _soa(from_pc)
    tx, ty, tz, N = _soa(to_pc)
    g0 = np.ascontiguousarray(init_xyzypr, dtype=np.float64).reshape(6)
    out = L.CLoCheckResult()
    L.check(L.lib().mola_lo_check_nonadjacent(icp._h if icp is not None else None, cb, None, C.byref(params.c),
                                              1 if is_loop_closure else 0, _fp(fx), _fp(fy), _fp(fz), M, _fp(tx), _fp(ty),
                                              _fp(tz), N, g0.ctypes.data_as(L._DP), int(seed), C.byref(out)))
    return CheckResult(Results.from_c(out.icp), out.best_guess, out.n_attempts, np.array(out.init_guess_used),
                       out.correction_percent, bool(out.edge_accepted))


class LidarOdometry:
    """`icp`: an `ICP` (GPU) -- or `align_fn(from(3,M), to(3,N), T0 4x4, Parameters) -> (T 4x4, quality, nIterations,
    terminationReason)` to drive the same host logic with another registration (CPU tests)."""

    def __init__(self, params: LidarOdometryParams, icp: ICP | None = None, align_fn=None):
        assert (icp is None) != (align_fn is None)
        self._icp = icp
        self._cb = None
        if align_fn is not None:
            self._cb = _align_callback(align_fn)
        self._h = L._H()
        L.check(L.lib().mola_lo_create(icp._h if icp is not None else None, self._cb or L.LO_ALIGN_FN(), None,
                                       C.byref(params.c), C.byref(self._h)))
        if icp is not None:   # the front-end holds the ICP handle's raw pointer: it must be destroyed first
            import weakref
            icp._dependents.append(weakref.ref(self))

    def close(self):
        if getattr(self, "_h", None) and self._h.value:
            L.lib().mola_lo_destroy(self._h)
            self._h = L._H()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def reset(self):
        L.check(L.lib().mola_lo_reset(self._h))

    def on_new_observation(self, timestamp: float, cloud) -> Step:
        x, y, z, n = _soa(cloud)
        s = L.CLoStep()
        t0 = time.perf_counter()
        rc = L.lib().mola_lo_process_scan(self._h, float(timestamp), _fp(x), _fp(y), _fp(z), n, C.byref(s))
        ms_native = (time.perf_counter() - t0) * 1e3
        L.check(rc)
        fac = (s.kf_factor_from, s.kf_factor_to, np.array(s.kf_factor_pose).reshape(4, 4)) if s.kf_factor_valid else None
        return Step(s.status, bool(s.used_with_vel_params), s.dt, np.array(s.rel_pose).reshape(4, 4), np.array(s.twist),
                    s.dist_since_last_kf, s.rot_since_last_kf, bool(s.keyframe_created), fac, s.reference_kf,
                    np.array(s.accum_since_last_kf).reshape(4, 4),
                    Results.from_c(s.icp) if s.status == L.LO_ICP_RAN else None, ms_native)
