"""Host-side mirror of the reference's ICP surface for the one hot path.

Names follow what `LidarOdometry` touches (src/LidarOdometry.cpp:57-88, 851-895):
`ICP.align(pcs_from, pcs_to, init_guess, params) -> Results`, `Parameters.load_from`,
`Results.{optimal_tf, quality, nIterations, terminationReason}`.  Everything here is
argument marshalling over the C-ABI; the arithmetic lives in the shared library."""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass, field

import numpy as np

from . import _lib as L

IterTermReason = {0: "Undefined", 1: "NoPairings", 2: "SolverError", 3: "MaxIterations", 4: "Stalled"}


def _dp(a: np.ndarray):
    return a.ctypes.data_as(C.POINTER(C.c_double))


def _fp(a: np.ndarray):
    return a.ctypes.data_as(C.POINTER(C.c_float))


def _pose16(T) -> np.ndarray:
    T = np.asarray(T, dtype=np.float64)
    if T.shape == (6,):
        return pose_from_xyzypr(T).reshape(16)
    if T.shape != (4, 4):
        raise ValueError("pose must be 4x4 or (x,y,z,yaw,pitch,roll)")
    return np.ascontiguousarray(T).reshape(16)


def _soa(pc) -> tuple[np.ndarray, np.ndarray, np.ndarray, int]:
    """(3,n) float32 array -> three contiguous fp32 rows."""
    a = np.asarray(pc)
    if a.ndim != 2 or a.shape[0] != 3:
        raise ValueError("a cloud is a (3, n) float32 structure-of-arrays")
    a = np.ascontiguousarray(a, dtype=np.float32)
    return a[0], a[1], a[2], a.shape[1]


# ------------------------------------------------------------------ host math


def pose_from_xyzypr(p) -> np.ndarray:
    p = np.ascontiguousarray(p, dtype=np.float64)
    T = np.empty(16)
    L.check(L.lib().mola_icp_pose_from_xyzypr(_dp(p), _dp(T)))
    return T.reshape(4, 4)


def pose_to_xyzypr(T) -> np.ndarray:
    T = _pose16(T)
    p = np.empty(6)
    L.check(L.lib().mola_icp_pose_to_xyzypr(_dp(T), _dp(p)))
    return p


def se3_log(T) -> np.ndarray:
    T = _pose16(T)
    o = np.empty(6)
    L.check(L.lib().mola_icp_se3_log(_dp(T), _dp(o)))
    return o


def stall_deltas(T, Tprev) -> tuple[float, float]:
    a, b = _pose16(T), _pose16(Tprev)
    dx, dr = C.c_double(), C.c_double()
    L.check(L.lib().mola_icp_stall_deltas(_dp(a), _dp(b), C.byref(dx), C.byref(dr)))
    return dx.value, dr.value


def solve_gauss_newton_planes(acc, T0, max_iterations=20):
    """host Gauss-Newton on the 92-term quadratic form of the point-to-plane cost -> (T, final cost, iterations)"""
    acc = np.ascontiguousarray(acc, dtype=np.float64)
    assert acc.shape == (92,)
    T0 = _pose16(T0)
    T = np.empty(16)
    cost, its = C.c_double(), C.c_uint32()
    L.check(L.lib().mola_icp_solve_gauss_newton_planes(_dp(acc), _dp(T0), int(max_iterations), _dp(T), C.byref(cost),
                                                       C.byref(its)))
    return T.reshape(4, 4), cost.value, its.value


def mixed_form(acc_p2p, T, form) -> np.ndarray:
    """the 92-term quadratic form `form` plus the share of a point-to-point pairing (its 24 sums at pose T)"""
    acc_p2p = np.ascontiguousarray(acc_p2p, dtype=np.float64)
    out = np.array(form, dtype=np.float64)
    assert acc_p2p.shape == (L.NACC,) and out.shape == (92,)
    L.check(L.lib().mola_icp_mixed_form(_dp(acc_p2p), _dp(_pose16(T)), _dp(out)))
    return out


def solve_horn(acc, cl=None, cg=None) -> np.ndarray:
    acc = np.ascontiguousarray(acc, dtype=np.float64)
    assert acc.shape == (L.NACC,)
    T = np.empty(16)
    clp = _dp(np.ascontiguousarray(cl, dtype=np.float64)) if cl is not None else None
    cgp = _dp(np.ascontiguousarray(cg, dtype=np.float64)) if cg is not None else None
    L.check(L.lib().mola_icp_solve_horn(_dp(acc), clp, cgp, _dp(T)))
    return T.reshape(4, 4)


# ------------------------------------------------------------------ parameters / results


@dataclass
class Parameters:
    """mp2p_icp::Parameters + the pipeline settings of one `icp-settings-*.yaml`
    (params/icp-settings-regular.yaml:10-46)."""
    c: L.CParams = field(default_factory=L.CParams)

    def __post_init__(self):
        if self.c.max_iterations == 0 and self.c.matcher_threshold == 0.0:
            L.check(L.lib().mola_icp_params_default(C.byref(self.c)))

    @classmethod
    def load_from(cls, yaml_text: str) -> "Parameters":
        p = cls()
        L.check(L.lib().mola_icp_params_from_yaml(yaml_text.encode(), C.byref(p.c)))
        return p

    @classmethod
    def load_from_file(cls, path: str, mola_dir: str | None = None, key: str | None = None) -> "Parameters":
        p = cls()
        L.check(L.lib().mola_icp_params_from_yaml_file(path.encode(), mola_dir.encode() if mola_dir else None,
                                                       key.encode() if key else None, C.byref(p.c)))
        return p

    def __getattr__(self, name):
        c = object.__getattribute__(self, "c")
        if name in {f[0] for f in L.CParams._fields_}:
            return getattr(c, name)
        raise AttributeError(name)

    def __setattr__(self, name, value):
        if name != "c" and name in {f[0] for f in L.CParams._fields_}:
            setattr(self.c, name, value)
        else:
            object.__setattr__(self, name, value)

    #: the fields of `mp2p_icp::Parameters` proper (cpp:77-78; icpreg:10-21) -- what the reference passes per call
    CALL_FIELDS = ("max_iterations", "min_abs_step_trans", "min_abs_step_rot", "use_scale_outlier_detector",
                   "scale_outlier_threshold", "use_robust_kernel", "robust_kernel_param", "robust_kernel_scale")

    @staticmethod
    def compose(object_settings: "Parameters", call_parameters: "Parameters") -> "Parameters":
        """the ICP object's matcher/solver/quality settings run with ANOTHER case's `mp2p_icp::Parameters`
        (src/LidarOdometry.cpp:287-290 + 869): mola_icp_params_compose()"""
        q = Parameters()
        L.check(L.lib().mola_icp_params_compose(C.byref(object_settings.c), C.byref(call_parameters.c), C.byref(q.c)))
        return q

    def copy(self) -> "Parameters":
        q = Parameters()
        C.memmove(C.byref(q.c), C.byref(self.c), C.sizeof(L.CParams))
        return q


@dataclass
class Results:
    """The fields of mp2p_icp::Results consumed at src/LidarOdometry.cpp:873-888."""
    optimal_tf: np.ndarray          # 4x4 mean (pose of `to` wrt `from`)
    optimal_tf_cov: np.ndarray      # 6x6
    quality: float
    nIterations: int
    terminationReason: int
    n_pairs: int
    rmse: float
    ms_upload: float
    ms_iterations: float
    ms_quality: float
    ms_nn_kernel: float
    n_nn_launches: int
    nn_kernel_used: int
    nn_pairs_evaluated: int

    @classmethod
    def from_c(cls, r: L.CResult) -> "Results":
        return cls(np.array(r.T).reshape(4, 4), np.array(r.cov).reshape(6, 6), r.quality, r.n_iterations,
                   r.termination, r.n_pairs, r.rmse, r.ms_upload, r.ms_iterations, r.ms_quality, r.ms_nn_kernel,
                   r.n_nn_launches, r.nn_kernel_used, r.nn_pairs_evaluated)

    @property
    def termination_name(self) -> str:
        return IterTermReason.get(self.terminationReason, "?")


def _align_batch(fn, handle, pairs, init_guesses, params: "Parameters") -> list["Results"]:
    n = len(pairs)
    keep = []
    FPP = C.POINTER(C.c_float) * n
    arrs = [FPP() for _ in range(6)]
    Ms, Ns = (C.c_size_t * n)(), (C.c_size_t * n)()
    for i, (f, t) in enumerate(pairs):
        fx, fy, fz, M = _soa(f)
        tx, ty, tz, N = _soa(t)
        keep += [fx, fy, fz, tx, ty, tz]
        for a, v in zip(arrs, (fx, fy, fz, tx, ty, tz)):
            a[i] = _fp(v)
        Ms[i], Ns[i] = M, N
    Ts = np.ascontiguousarray(np.stack([_pose16(g) for g in init_guesses]) if n else np.zeros((0, 16)))
    res = (L.CResult * max(1, n))()
    L.check(fn(handle, n, arrs[0], arrs[1], arrs[2], Ms, arrs[3], arrs[4], arrs[5], Ns, _dp(Ts), C.byref(params.c), res))
    return [Results.from_c(res[k]) for k in range(n)]


def pool_assignment(n_pairs: int, n_devices: int) -> list[int]:
    """the device pool's dealing rule (mola_icp_pool_assignment): pair i -> device slot i mod n"""
    out = (C.c_int * max(1, n_pairs))()
    L.check(L.lib().mola_icp_pool_assignment(n_pairs, n_devices, out))
    return list(out[:n_pairs])


class DevicePool:
    """One ICP handle per GPU of the node; independent pairs are dealt round-robin (BASELINE config[3]: the
    nearby-KF / loop-closure batch of src/LidarOdometry.cpp:704-741 over 8 x MI355X, no collective)."""

    def __init__(self, devices=None):
        self._h = L._H()
        if devices:
            arr = (C.c_int * len(devices))(*devices)
            L.check(L.lib().mola_icp_pool_create(arr, len(devices), C.byref(self._h)))
        else:
            L.check(L.lib().mola_icp_pool_create(None, 0, C.byref(self._h)))

    def __len__(self):
        n = C.c_int()
        L.check(L.lib().mola_icp_pool_size(self._h, C.byref(n)))
        return n.value

    def align_batch(self, pairs, init_guesses, params: "Parameters") -> list["Results"]:
        return _align_batch(L.lib().mola_icp_pool_align_batch, self._h, pairs, init_guesses, params)

    def last_shares(self) -> list[int]:
        """pairs each device slot served in the last align_batch call (the slots pull chunks from a shared cursor)"""
        n = len(self)
        arr = (C.c_size_t * n)()
        L.check(L.lib().mola_icp_pool_last_shares(self._h, arr, n))
        return [int(v) for v in arr]

    def close(self):
        if getattr(self, "_h", None) and self._h.value:
            L.lib().mola_icp_pool_destroy(self._h)
            self._h = L._H()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


# ------------------------------------------------------------------ the ICP object


class ICP:
    """One `mp2p_icp::ICP` object of `Parameters::ICP_case`
    (include/mola-fe-lidar/LidarOdometry.h:96-102), backed by one MI355X."""

    def __init__(self, device: int = -1):
        self._h = L._H()
        L.check(L.lib().mola_icp_create(device, C.byref(self._h)))
        self._keep = []       # device tensors / callbacks that must outlive the handle's use of them
        self._ar_cb = None
        self._dependents = []  # weakrefs to objects holding this handle's raw pointer (LidarOdometry): closed first

    def close(self):
        for ref in getattr(self, "_dependents", []):
            d = ref()
            if d is not None:
                d.close()
        self._dependents = []
        if getattr(self, "_h", None) and self._h.value:
            L.lib().mola_icp_destroy(self._h)
            self._h = L._H()
        lc = getattr(self, "_local_comm", None)   # (after the handle that used it)
        if lc is not None:
            if self._local_comm_owned:
                lc.close()
            self._local_comm = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- the hot path, as called at src/LidarOdometry.cpp:869-871
    def align(self, pcs_from, pcs_to, init_guess_to_wrt_from, params: Parameters) -> Results:
        fx, fy, fz, M = _soa(pcs_from)
        tx, ty, tz, N = _soa(pcs_to)
        T = _pose16(init_guess_to_wrt_from)
        r = L.CResult()
        L.check(L.lib().mola_icp_align(self._h, _fp(fx), _fp(fy), _fp(fz), M, _fp(tx), _fp(ty), _fp(tz), N, _dp(T),
                                       C.byref(params.c), C.byref(r)))
        return Results.from_c(r)

    def align_batch(self, pairs, init_guesses, params: Parameters) -> list[Results]:
        """pairs = [(pcs_from, pcs_to), ...]: the nearby-KF / loop-closure batch
        (src/LidarOdometry.cpp:704-741)."""
        return _align_batch(L.lib().mola_icp_align_batch, self._h, pairs, init_guesses, params)

    def align_multi_init(self, pcs_from, pcs_to, init_guesses, params: Parameters):
        """Loop-closure Monte-Carlo (src/LidarOdometry.cpp:767-788): one pair, several initial poses, the
        attempt with the highest goodness wins.  Returns (results, best_index or -1)."""
        fx, fy, fz, M = _soa(pcs_from)
        tx, ty, tz, N = _soa(pcs_to)
        n = len(init_guesses)
        Ts = np.ascontiguousarray(np.stack([_pose16(g) for g in init_guesses]) if n else np.zeros((0, 16)))
        res = (L.CResult * max(1, n))()
        best = L.CResult()
        bi = C.c_int(-1)
        L.check(L.lib().mola_icp_align_multi_init(self._h, _fp(fx), _fp(fy), _fp(fz), M, _fp(tx), _fp(ty), _fp(tz), N, n,
                                                  _dp(Ts), C.byref(params.c), res, C.byref(best), C.byref(bi)))
        return [Results.from_c(res[k]) for k in range(n)], bi.value

    # -- device-resident cloud cache (row f4): keyframe clouds stay prepared in HBM, keyed by id
    def cloud_put(self, cloud_id: int, pc):
        x, y, z, n = _soa(pc)
        L.check(L.lib().mola_icp_cloud_put(self._h, int(cloud_id), _fp(x), _fp(y), _fp(z), n))

    def cloud_drop(self, cloud_id: int):
        L.check(L.lib().mola_icp_cloud_drop(self._h, int(cloud_id)))

    def cloud_count(self) -> tuple[int, int]:
        n, b = C.c_size_t(), C.c_size_t()
        L.check(L.lib().mola_icp_cloud_count(self._h, C.byref(n), C.byref(b)))
        return n.value, b.value

    @staticmethod
    def set_wait_policy(policy: str | int):
        """how host threads wait for a pass's sums: "spin" (default), "yield", "block" (mola_icp_set_wait_policy; process-wide)"""
        code = {"spin": 0, "yield": 1, "block": 2}.get(policy, policy)
        L.check(L.lib().mola_icp_set_wait_policy(int(code)))

    @staticmethod
    def device_pool_trim(device: int = 0, keep_bytes: int = 0) -> int:
        """free parked device blocks of dropped clouds down to `keep_bytes` per device; returns what stays parked on `device`"""
        b = C.c_size_t()
        L.check(L.lib().mola_icp_device_pool_trim(int(device), int(keep_bytes), C.byref(b)))
        return b.value

    def align_cached(self, from_id: int, to_id: int, init_guess_to_wrt_from, params: Parameters) -> Results:
        T = _pose16(init_guess_to_wrt_from)
        r = L.CResult()
        L.check(L.lib().mola_icp_align_cached(self._h, int(from_id), int(to_id), _dp(T), C.byref(params.c), C.byref(r)))
        return Results.from_c(r)

    def align_cached_put(self, from_id: int, to_id: int, to_pc, init_guess_to_wrt_from, params: Parameters) -> Results:
        """`cloud_put(to_id, to_pc)` + `align_cached(from_id, to_id, ...)` in one call, without a host wait between the new cloud's
        prepare chain and the align's first launches (the odometry step); the cloud is cached under `to_id` afterwards"""
        T = _pose16(init_guess_to_wrt_from)
        x, y, z, n = _soa(to_pc)
        r = L.CResult()
        done = C.c_int(0)
        L.check(L.lib().mola_icp_align_cached_put(self._h, int(from_id), int(to_id), _fp(x), _fp(y), _fp(z), n, _dp(T), C.byref(params.c),
                                                  C.byref(r), C.byref(done)))
        return Results.from_c(r)

    def voxel_downsample(self, pc, voxel_size: float) -> np.ndarray:
        """one centroid per occupied voxel (row f4; the decimation step before the ICP) -> (3, n_voxels) float32"""
        x, y, z, n = _soa(pc)
        out = np.empty((3, max(1, n)), dtype=np.float32)
        nv = C.c_size_t()
        L.check(L.lib().mola_icp_voxel_downsample(self._h, _fp(x), _fp(y), _fp(z), n, float(voxel_size), _fp(out[0]),
                                                  _fp(out[1]), _fp(out[2]), n, C.byref(nv)))
        return np.ascontiguousarray(out[:, :nv.value])

    # -- resident clouds (already in HBM): bench + sharded path
    @staticmethod
    def _is_device_tensor(x) -> bool:
        return hasattr(x, "is_cuda") and x.is_cuda

    def set_map(self, pc):
        if self._is_device_tensor(pc):
            import torch
            assert pc.dtype == torch.float32 and pc.dim() == 2 and pc.shape[0] == 3 and pc.is_contiguous()
            # the handle's stream is non-blocking: no implicit ordering against torch's streams (ordering contract of
            # mola_icp_set_*_device in include/mola_icp_amd.h) -- whatever produced `pc` must have finished
            torch.cuda.current_stream(pc.device).synchronize()
            self._keep_map = pc
            L.check(L.lib().mola_icp_set_map_device(self._h, pc[0].data_ptr(), pc[1].data_ptr(), pc[2].data_ptr(),
                                                    pc.shape[1]))
        else:
            x, y, z, n = _soa(pc)
            L.check(L.lib().mola_icp_set_map_host(self._h, _fp(x), _fp(y), _fp(z), n))

    def set_local(self, pc):
        if self._is_device_tensor(pc):
            import torch
            assert pc.dtype == torch.float32 and pc.dim() == 2 and pc.shape[0] == 3 and pc.is_contiguous()
            torch.cuda.current_stream(pc.device).synchronize()   # see set_map
            self._keep_local = pc
            L.check(L.lib().mola_icp_set_local_device(self._h, pc[0].data_ptr(), pc[1].data_ptr(), pc[2].data_ptr(),
                                                      pc.shape[1]))
        else:
            x, y, z, n = _soa(pc)
            L.check(L.lib().mola_icp_set_local_host(self._h, _fp(x), _fp(y), _fp(z), n))

    def set_local_shard(self, pc_full, rank: int, nranks: int) -> int:
        """keeps the slice of rank `rank` of the full scan's Hilbert order (cut on the device) as the local cloud;
        returns the shard's size"""
        n = C.c_size_t(0)
        if self._is_device_tensor(pc_full):
            import torch
            assert pc_full.dtype == torch.float32 and pc_full.dim() == 2 and pc_full.shape[0] == 3 and pc_full.is_contiguous()
            torch.cuda.current_stream(pc_full.device).synchronize()   # see set_map
            L.check(L.lib().mola_icp_set_local_shard_device(self._h, pc_full[0].data_ptr(), pc_full[1].data_ptr(),
                                                            pc_full[2].data_ptr(), pc_full.shape[1], rank, nranks, C.byref(n)))
        else:
            x, y, z, nt = _soa(pc_full)
            L.check(L.lib().mola_icp_set_local_shard_host(self._h, _fp(x), _fp(y), _fp(z), nt, rank, nranks, C.byref(n)))
        self._keep_local = None
        self._shard_n = int(n.value)
        return self._shard_n

    def set_local_shard_range(self, pc_full, lo: int, hi: int) -> int:
        """keeps the slice [lo, hi) of the full scan's Hilbert order as the local cloud (cost-balanced cuts:
        `sharded.balanced_cuts`); returns the shard's size"""
        n = C.c_size_t(0)
        if self._is_device_tensor(pc_full):
            import torch
            assert pc_full.dtype == torch.float32 and pc_full.dim() == 2 and pc_full.shape[0] == 3 and pc_full.is_contiguous()
            torch.cuda.current_stream(pc_full.device).synchronize()   # see set_map
            L.check(L.lib().mola_icp_set_local_shard_range_device(self._h, pc_full[0].data_ptr(), pc_full[1].data_ptr(),
                                                                  pc_full[2].data_ptr(), pc_full.shape[1], int(lo), int(hi), C.byref(n)))
        else:
            x, y, z, nt = _soa(pc_full)
            L.check(L.lib().mola_icp_set_local_shard_range_host(self._h, _fp(x), _fp(y), _fp(z), nt, int(lo), int(hi), C.byref(n)))
        self._keep_local = None
        self._shard_n = int(n.value)
        return self._shard_n

    def local_shard_indices(self) -> np.ndarray:
        """the shard's points as indices into the full scan, in the shard's own order"""
        idx = np.empty(max(1, getattr(self, "_shard_n", 0)), dtype=np.int32)
        L.check(L.lib().mola_icp_local_shard_indices(self._h, idx.ctypes.data_as(C.POINTER(C.c_int32))))
        return idx[:getattr(self, "_shard_n", 0)]

    def shard_reach_box(self, T, margin: float):
        lo, hi = np.zeros(3), np.zeros(3)
        L.check(L.lib().mola_icp_shard_reach_box(self._h, _pose16(T).ctypes.data_as(L._DP), float(margin),
                                                 lo.ctypes.data_as(L._DP), hi.ctypes.data_as(L._DP)))
        return lo, hi

    def set_map_slab(self, pc, lo, hi) -> int:
        """only the map points inside [lo, hi] become this handle's map (original indices are kept); returns how many"""
        lo = np.ascontiguousarray(lo, dtype=np.float64)
        hi = np.ascontiguousarray(hi, dtype=np.float64)
        n = C.c_size_t(0)
        if self._is_device_tensor(pc):
            import torch
            assert pc.dtype == torch.float32 and pc.dim() == 2 and pc.shape[0] == 3 and pc.is_contiguous()
            torch.cuda.current_stream(pc.device).synchronize()
            L.check(L.lib().mola_icp_set_map_slab_device(self._h, pc[0].data_ptr(), pc[1].data_ptr(), pc[2].data_ptr(), pc.shape[1],
                                                         lo.ctypes.data_as(L._DP), hi.ctypes.data_as(L._DP), C.byref(n)))
        else:
            x, y, z, m = _soa(pc)
            L.check(L.lib().mola_icp_set_map_slab_host(self._h, _fp(x), _fp(y), _fp(z), m, lo.ctypes.data_as(L._DP),
                                                       hi.ctypes.data_as(L._DP), C.byref(n)))
        self._keep_map = None
        return int(n.value)

    def set_global_sizes(self, n_local_total: int, n_map_total: int):
        L.check(L.lib().mola_icp_set_global_sizes(self._h, n_local_total, n_map_total))

    def set_stream(self, hip_stream_ptr: int | None):
        L.check(L.lib().mola_icp_set_stream(self._h, C.c_void_p(hip_stream_ptr or 0)))

    def set_profiling(self, on: bool = True):
        """kernel statistics (`ms_nn_kernel`, `nn_pairs_evaluated`) of the following aligns; off by default (the HIP events
        around every matcher launch cost ~8 us per iteration at odometry sizes)"""
        L.check(L.lib().mola_icp_set_profiling(self._h, 1 if on else 0))

    def forget_warm_start(self, schedule: bool = False):
        """drop the pairing / neighbour lists / seeds / plane cache earlier aligns left for the resident clouds: the next align is
        stateless.  The sorted clouds and their work-queue cost order (made once per cloud pair) stay; `schedule=True` drops the
        order too -- the next align costs what the very first one on this pair cost"""
        L.check((L.lib().mola_icp_forget_cloud_schedule if schedule else L.lib().mola_icp_forget_warm_start)(self._h))

    def set_allreduce(self, fn):
        """fn(np.ndarray[float64] of 24) -> None, summing in place across ranks (None = single GPU)."""
        if fn is None:
            self._ar_cb = None
            L.check(L.lib().mola_icp_set_allreduce(self._h, L.ALLREDUCE_FN(), None))
            return

        def _cb(buf, n, device_ptr, user):
            try:
                a = np.ctypeslib.as_array(buf, shape=(n,))
                fn(a)
                return 0
            except Exception:  # never let an exception cross the C boundary
                import traceback
                traceback.print_exc()
                return 1

        self._ar_cb = L.ALLREDUCE_FN(_cb)
        L.check(L.lib().mola_icp_set_allreduce(self._h, self._ar_cb, None))

    def comm_init(self, group=None):
        """Native RCCL communicator for the query-sharded path: rank 0 creates the id, torch.distributed
        carries its 128 bytes to the other ranks, every rank joins (collective call)."""
        import os
        import torch
        import torch.distributed as dist
        rank, world = dist.get_rank(group), dist.get_world_size(group)
        cand = os.path.join(os.path.dirname(torch.__file__), "lib", "librccl.so")
        if os.path.exists(cand):  # the RCCL this process already uses
            L.check(L.lib().mola_icp_comm_set_library(cand.encode()))
        ident = (C.c_uint8 * 128)()
        status, err = 0, None
        if rank == 0:
            try:  # a failure here must still reach the broadcast below, or the other ranks wait for ever
                L.check(L.lib().mola_icp_comm_unique_id(ident))
            except Exception as e:  # noqa: BLE001
                status, err = 1, e
        on_gpu = dist.get_backend(group) == "nccl"
        t = torch.tensor(list(ident) + [status], dtype=torch.uint8)
        if on_gpu:
            t = t.cuda()
        src = dist.get_global_rank(group, 0) if group is not None else 0
        dist.broadcast(t, src=src, group=group)
        got = t.cpu().tolist()
        if got[128] != 0:
            raise err if err is not None else L.IcpError(L.E_COMM,
                                                         "rank 0 could not create the RCCL unique id")
        ident = (C.c_uint8 * 128)(*got[:128])
        L.check(L.lib().mola_icp_comm_init(self._h, ident, world, rank))

    def comm_init_local(self, group=None, comm=None, timeout_s: float = 30.0):
        """Node-local communicator for the query-sharded path (`sharded.LocalComm`: a shared-memory mailbox on the host, where
        the reduced block is consumed -- nothing measurable per step where RCCL's costs 13-19 us): collective over the group's ranks,
        which must run on ONE node.  `comm` = an existing LocalComm to attach instead of creating one."""
        from .sharded import LocalComm
        self._local_comm = comm if comm is not None else LocalComm.from_group(group, timeout_s)
        self._local_comm_owned = comm is None
        L.check(L.lib().mola_icp_comm_attach_local(self._h, self._local_comm.handle))
        return self._local_comm

    def comm_nranks(self) -> int:
        """the size the communicator itself reports: ncclCommCount for RCCL, the ranks that joined for the local one"""
        n = C.c_int(0)
        L.check(L.lib().mola_icp_comm_nranks(self._h, C.byref(n)))
        return int(n.value)

    def comm_destroy(self):
        L.check(L.lib().mola_icp_comm_destroy(self._h))
        lc = getattr(self, "_local_comm", None)
        if lc is not None:
            if self._local_comm_owned:
                lc.close()
            self._local_comm = None

    def align_resident(self, init_guess_to_wrt_from, params: Parameters) -> Results:
        T = _pose16(init_guess_to_wrt_from)
        r = L.CResult()
        L.check(L.lib().mola_icp_align_resident(self._h, _dp(T), C.byref(params.c), C.byref(r)))
        return Results.from_c(r)

    # -- single stages (parity tests)
    def match(self, T, threshold: float, n_local: int, nn_kernel: int = L.NN_AUTO, copy: bool = True):
        T = _pose16(T)
        idx = np.empty(n_local, dtype=np.int32) if copy else None
        d2 = np.empty(n_local, dtype=np.float32) if copy else None
        n = C.c_uint64()
        L.check(L.lib().mola_icp_match(self._h, _dp(T), threshold, nn_kernel,
                                       idx.ctypes.data_as(C.POINTER(C.c_int32)) if copy else None,
                                       _fp(d2) if copy else None, C.byref(n)))
        return idx, d2, n.value

    def match_planes(self, T, params: Parameters, n_local: int):
        """point-to-plane matcher on the resident clouds -> (valid, centroid (N,3), normal (N,3), knn_idx (N,knn), n)"""
        T = _pose16(T)
        valid = np.zeros(max(1, n_local), np.uint8)
        cen = np.zeros((max(1, n_local), 3))
        nor = np.zeros((max(1, n_local), 3))
        kidx = np.full((max(1, n_local), int(params.knn)), -1, np.int32)
        n = C.c_uint64()
        L.check(L.lib().mola_icp_match_planes(self._h, _dp(T), C.byref(params.c), valid.ctypes.data_as(C.POINTER(C.c_uint8)),
                                              _dp(cen), _dp(nor), kidx.ctypes.data_as(C.POINTER(C.c_int32)), C.byref(n)))
        return valid[:n_local], cen[:n_local], nor[:n_local], kidx[:n_local], n.value

    def accumulate_planes(self) -> np.ndarray:
        """the 92-term quadratic form of the stored plane pairing (after match_planes)"""
        acc = np.zeros(92)
        L.check(L.lib().mola_icp_accumulate_planes(self._h, _dp(acc)))
        return acc

    def accumulate(self, params: Parameters, Tcur, stage: int = 0, cl=None, cg=None, reset_outliers: bool = True):
        T = _pose16(Tcur)
        acc = np.empty(L.NACC)
        clp = _dp(np.ascontiguousarray(cl, dtype=np.float64)) if cl is not None else None
        cgp = _dp(np.ascontiguousarray(cg, dtype=np.float64)) if cg is not None else None
        L.check(L.lib().mola_icp_accumulate(self._h, C.byref(params.c), _dp(T), stage, clp, cgp,
                                            1 if reset_outliers else 0, _dp(acc)))
        return acc


def _stage_callbacks(match_fn, accumulate_fn, n_local_total, n_map_total, allreduce_fn=None):
    """ctypes stage callbacks around python functions; returns (CStageCallbacks, objects to keep alive)"""

    def _m(user, Tp, thr, n_out):
        try:
            T = np.ctypeslib.as_array(Tp, shape=(16,)).reshape(4, 4).copy()
            n_out[0] = int(match_fn(T, thr))
            return 0
        except Exception:
            import traceback
            traceback.print_exc()
            return L.E_INTERNAL

    def _a(user, pp, Tp, stage, clp, cgp, reset, acc_out):
        try:
            T = np.ctypeslib.as_array(Tp, shape=(16,)).reshape(4, 4).copy()
            cl = np.ctypeslib.as_array(clp, shape=(3,)).copy() if clp else None
            cg = np.ctypeslib.as_array(cgp, shape=(3,)).copy() if cgp else None
            p = Parameters()
            C.memmove(C.byref(p.c), pp, C.sizeof(L.CParams))
            acc = np.asarray(accumulate_fn(p, T, stage, cl, cg, bool(reset)), dtype=np.float64)
            np.ctypeslib.as_array(acc_out, shape=(L.NACC,))[:] = acc
            return 0
        except Exception:
            import traceback
            traceback.print_exc()
            return L.E_INTERNAL

    def _r(buf, n, device_ptr, user):
        try:
            allreduce_fn(np.ctypeslib.as_array(buf, shape=(n,)))
            return 0
        except Exception:
            import traceback
            traceback.print_exc()
            return 1

    cb = L.CStageCallbacks()
    cb.match = L.MATCH_CB(_m)
    cb.accumulate = L.ACCUM_CB(_a)
    cb.allreduce = L.ALLREDUCE_FN(_r) if allreduce_fn is not None else L.ALLREDUCE_FN()
    cb.user = None
    cb.n_local_total = n_local_total
    cb.n_map_total = n_map_total
    return cb, (cb.match, cb.accumulate, cb.allreduce)


def run_loop_batch(stages, init_guesses, params: Parameters) -> list[Results]:
    """The lockstep loop of the batched aligners (mola_icp_run_loop_batch) over caller-supplied stages:
    `stages` = [(match_fn, accumulate_fn, n_local_total, n_map_total), ...], one entry per problem."""
    n = len(stages)
    cbs = (L.CStageCallbacks * max(1, n))()
    keep = []
    for k, (m, a, nl, nm) in enumerate(stages):
        cb, ka = _stage_callbacks(m, a, nl, nm)
        cbs[k] = cb
        keep.append(ka)
    Ts = np.ascontiguousarray(np.stack([_pose16(g) for g in init_guesses]) if n else np.zeros((0, 16)))
    res = (L.CResult * max(1, n))()
    L.check(L.lib().mola_icp_run_loop_batch(cbs, n, _dp(Ts), C.byref(params.c), res))
    return [Results.from_c(res[k]) for k in range(n)]


def run_loop(match_fn, accumulate_fn, init_guess, params: Parameters, n_local_total: int, n_map_total: int,
             allreduce_fn=None) -> Results:
    """The library's own iteration-control loop over caller-supplied stages
    (mola_icp_run_loop): `match_fn(T4x4, threshold) -> n_pairs`,
    `accumulate_fn(params, T4x4, stage, cl, cg, reset) -> acc[24]`, `allreduce_fn(acc) -> None`."""
    cb, _keep = _stage_callbacks(match_fn, accumulate_fn, n_local_total, n_map_total, allreduce_fn)
    T = _pose16(init_guess)
    r = L.CResult()
    L.check(L.lib().mola_icp_run_loop(C.byref(cb), _dp(T), C.byref(params.c), C.byref(r)))
    return Results.from_c(r)
