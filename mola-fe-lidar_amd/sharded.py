"""Query-sharded multi-GPU ICP: one process per GPU, the map replicated,
the local cloud (`to`, the queries) split into contiguous shards, and ONE all-reduce of the
24-double accumulator block per accumulation pass (RCCL over xGMI when the process group's
backend is "nccl"; gloo on CPU for tests).  Every rank then runs the identical fp64 solve and
stall test, so no pose broadcast is needed.  The reference has no analogue: inside one
`align()` it is serial (src/LidarOdometry.cpp:869-871)."""
from __future__ import annotations

import numpy as np


def shard_bounds(n: int, rank: int, world: int) -> tuple[int, int]:
    """contiguous balanced split of n queries: the first (n % world) ranks get one more."""
    base, rem = divmod(n, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def spatial_order(points: np.ndarray, bits: int = 10) -> np.ndarray:
    """Permutation that puts a 3xN cloud in Morton (Z-curve) order.  Query shards must be spatially compact: the
    tiled matcher works on groups of consecutive sorted queries, and a random 1/W subsample of the scan is W times
    sparser than the map, so every group would sweep W times more map tiles (measured: a random 1/8 shard of the
    1M x 1M job costs as much as the whole job).  Contiguous slices of this order keep the scan's own density."""
    p = np.asarray(points, dtype=np.float64)
    lo = p.min(axis=1, keepdims=True)
    ext = float(max((p.max(axis=1, keepdims=True) - lo).max(), 1e-30))
    q = np.minimum(((p - lo) * ((1 << bits) / ext)).astype(np.uint32), (1 << bits) - 1)

    def spread(v):  # 10 bits -> every third bit
        v = (v | (v << np.uint32(16))) & np.uint32(0x030000FF)
        v = (v | (v << np.uint32(8))) & np.uint32(0x0300F00F)
        v = (v | (v << np.uint32(4))) & np.uint32(0x030C30C3)
        v = (v | (v << np.uint32(2))) & np.uint32(0x09249249)
        return v

    key = spread(q[0]) | (spread(q[1]) << np.uint32(1)) | (spread(q[2]) << np.uint32(2))
    return np.argsort(key, kind="stable")


def slab_margin_for_guess(box_lo, box_hi, gate: float, max_dt: float, max_drot: float) -> float:
    """How far beyond its own box (at the guess) a query shard can reach into the map while the pose stays within
    (max_dt metres, max_drot radians) of the guess: a point p moves by at most max_dt + 2 sin(max_drot / 2) |p| under such a
    correction (rotation about the frame origin), and pairs within `gate` of where it lands.  `box_lo/hi` = the shard's box at the
    guess with zero margin (`ICP.shard_reach_box(guess, 0)`); the farthest corner bounds |p|.  The margin a rank passes to
    `shard_reach_box(guess, margin)` before cutting its map slab -- per rank: a shard near the origin needs less than one 80 m out."""
    lo, hi = np.asarray(box_lo, np.float64), np.asarray(box_hi, np.float64)
    far = float(np.linalg.norm(np.maximum(np.abs(lo), np.abs(hi))))
    return float(gate + max_dt + 2.0 * np.sin(0.5 * max_drot) * far)


def balanced_cuts(cuts, cost, relax: float = 1.0) -> list[int]:
    """Cuts of equal COST.  `cuts` = the W + 1 boundaries of the shards in force (positions in the scan's Hilbert order, cuts[0] = 0,
    cuts[W] = n), `cost` = what each shard's step cost (W values: seconds of a timed iteration, or the matcher's own time) -- taken
    as uniform inside a shard.  Returns W + 1 new boundaries where the cumulated cost reaches k / W of the total.  Every rank
    computes the same cuts from the same all-reduced cost vector.  One or two rounds settle: the cost per query varies smoothly
    along the curve (it follows how far the guess displaces that part of the scan).  `relax` < 1 moves every cut only that fraction
    of the way (the model is crude where a shard holds a short, very expensive stretch: full steps can overshoot)."""
    cuts = [int(c) for c in cuts]
    w = len(cuts) - 1
    cost = np.maximum(np.asarray(cost, np.float64), 1e-12)
    assert w >= 1 and cost.shape == (w,) and all(cuts[k] <= cuts[k + 1] for k in range(w))
    cum = np.concatenate([[0.0], np.cumsum(cost)])
    new = [0]
    for k in range(1, w):
        target = cum[-1] * k / w
        j = int(np.searchsorted(cum, target, side="right") - 1)
        j = min(max(j, 0), w - 1)
        frac = (target - cum[j]) / cost[j]
        target_cut = cuts[j] + frac * (cuts[j + 1] - cuts[j])
        new.append(int(round(cuts[k] + relax * (target_cut - cuts[k]))))
    new.append(cuts[-1])
    # no shard below a quarter of the equal share (a cost vector that says otherwise is a measurement gone wrong -- ranks
    # time-slicing one device, a probe that hit a cold start -- not a property of the scan), monotone whatever the rounding did
    n, lo_share = cuts[-1] - cuts[0], (cuts[-1] - cuts[0]) // (4 * w)
    for k in range(1, w):
        new[k] = max(new[k], new[k - 1] + lo_share)
    for k in range(w - 1, 0, -1):
        new[k] = min(new[k], new[k + 1] - lo_share)
    for k in range(1, w + 1):
        new[k] = max(new[k], new[k - 1])
    return new


def make_allreduce(group=None, device=None):
    """Returns fn(acc: np.ndarray[float64]) that sums `acc` in place over the process group.
    gloo: reduces the host buffer directly.  nccl (= RCCL): stages through a device tensor."""
    import torch
    import torch.distributed as dist

    backend = dist.get_backend(group)
    if backend == "gloo":
        def fn(acc: np.ndarray) -> None:
            dist.all_reduce(torch.from_numpy(acc), op=dist.ReduceOp.SUM, group=group)
        return fn

    dev = device if device is not None else torch.device("cuda", torch.cuda.current_device())
    stages = {}   # one staging tensor per block length: 24 doubles (point-to-point), 92 (the point-to-plane form)

    def fn(acc: np.ndarray) -> None:
        n = acc.shape[0]
        stage = stages.get(n)
        if stage is None:
            stage = stages[n] = torch.zeros(n, dtype=torch.float64, device=dev)
        stage.copy_(torch.from_numpy(acc))
        dist.all_reduce(stage, op=dist.ReduceOp.SUM, group=group)
        acc[:] = stage.cpu().numpy()
    return fn


class LocalComm:
    """The node-local communicator (`mola_icp_local_comm_*`, csrc/local_comm.cpp): the all-reduce of one node's ranks through
    a shared-memory mailbox on the host -- where the reduced block is consumed.  `create` is collective; `name` must be unique
    to the job (`LocalComm.from_group` lets rank 0 pick one and ships it over torch.distributed)."""

    def __init__(self, name: str, nranks: int, rank: int, timeout_s: float = 30.0):
        import ctypes as C
        from . import _lib as L
        self._L = L
        self._c = C.c_void_p()
        L.check(L.lib().mola_icp_local_comm_create(name.encode(), int(nranks), int(rank), float(timeout_s), C.byref(self._c)))
        self.rank, self.world = rank, nranks

    @classmethod
    def from_group(cls, group=None, timeout_s: float = 30.0):
        import os
        import torch.distributed as dist
        rank, world = dist.get_rank(group), dist.get_world_size(group)
        names = [f"mola_icp_{os.getpid()}_{int.from_bytes(os.urandom(6), 'little'):x}" if rank == 0 else None]
        src = dist.get_global_rank(group, 0) if group is not None else 0
        dist.broadcast_object_list(names, src=src, group=group)
        return cls(names[0], world, rank, timeout_s)

    @property
    def handle(self):
        return self._c

    def allreduce(self, acc: np.ndarray) -> None:
        """sum a contiguous float64 array (<= 120 values) in place over the ranks; every rank gets the same bits"""
        import ctypes as C
        assert acc.dtype == np.float64 and acc.flags["C_CONTIGUOUS"]
        self._L.check(self._L.lib().mola_icp_local_comm_allreduce(self._c, acc.ctypes.data_as(C.POINTER(C.c_double)), int(acc.size)))

    def nranks(self) -> int:
        import ctypes as C
        n = C.c_int(0)
        self._L.check(self._L.lib().mola_icp_local_comm_nranks(self._c, C.byref(n)))
        return int(n.value)

    def abort(self) -> None:
        self._L.check(self._L.lib().mola_icp_local_comm_abort(self._c))

    def close(self) -> None:
        if self._c:
            self._L.lib().mola_icp_local_comm_destroy(self._c)
            self._c = None

    def __del__(self):
        try:
            self.close()
        except Exception:  # noqa: BLE001
            pass


class ShardedICP:
    """`ICP` whose local cloud is this rank's shard.  Usage (every rank):
        s = ShardedICP(icp)                 # icp = ICP(device=local_rank)
        s.set_clouds(map_pc, local_pc_full) # or set_shard(map_pc, my_shard, n_local_total)
        res = s.align(init_guess, params)   # identical Results on every rank
    """

    def __init__(self, icp, group=None, collective: str = "auto"):
        """collective: "local" = the node-local shared-memory communicator (`LocalComm`; the ranks must share a node), "rccl" /
        "hook" = the process group's own transport (native RCCL on the device block for the nccl backend, torch.distributed from
        the host hook otherwise), "auto" = local when every rank reports the same host name, else the group's transport."""
        import torch.distributed as dist
        self.icp = icp
        self.group = group
        self.rank = dist.get_rank(group)
        self.world = dist.get_world_size(group)
        self._ar = None
        self.collective = None
        self._cuts = None
        if self.world > 1:
            if collective == "auto":
                import socket
                names = [None] * self.world
                dist.all_gather_object(names, socket.gethostname(), group=group)
                collective = "local" if len(set(names)) == 1 else "group"
            if collective == "local":
                icp.comm_init_local(group)
                self.collective = "local"
            elif dist.get_backend(group) == "nccl" and collective != "hook":
                icp.comm_init(group)  # native RCCL on the device accumulator block
                self.collective = "rccl"
            else:
                self._ar = make_allreduce(group)
                icp.set_allreduce(self._ar)
                self.collective = "hook"

    def set_clouds(self, map_pc, local_pc_full, spatial: bool = True, init_guess=None, slab_margin: float | None = None,
                   guess_uncertainty: tuple[float, float] | None = None, gate: float | None = None):
        """Every rank passes the same full clouds.  It keeps (i) a spatially compact shard of the scan: its slice of the
        scan's Hilbert order, cut on the device (`mola_icp_set_local_shard_*`; no host argsort) and, with a margin, (ii) only
        the part of the map that shard can reach from `init_guess`: its moved bounding box grown by the margin
        (`mola_icp_set_map_slab_*`).  The margin is `slab_margin` [m], or -- `guess_uncertainty` = (max_dt [m], max_drot [rad])
        and `gate` [m] given -- `slab_margin_for_guess` of THIS rank's shard.  It must cover the matcher's gate plus the pose
        correction the align may make; an align that leaves the slab fails loudly (`IcpError`, "outside its map slab") --
        `align` below then cuts a larger slab and runs again."""
        n = local_pc_full.shape[1]
        self._full = (map_pc, local_pc_full, init_guess)
        self._spatial = spatial
        self._guess_uncertainty, self._gate = guess_uncertainty, gate
        self._slab_margin = slab_margin
        self._slab_scale = 1.0
        self._cut_shard()
        self._cut_slab()
        self.icp.set_global_sizes(n, map_pc.shape[1])

    def _cut_shard(self):
        _, local_pc_full, _ = self._full
        n = local_pc_full.shape[1]
        if self._spatial and self._cuts is not None:
            self.icp.set_local_shard_range(local_pc_full, self._cuts[self.rank], self._cuts[self.rank + 1])
        elif self._spatial:
            self.icp.set_local_shard(local_pc_full, self.rank, self.world)
        else:
            lo, hi = shard_bounds(n, self.rank, self.world)
            shard = local_pc_full[:, lo:hi]
            if hasattr(shard, "contiguous"):
                shard = shard.contiguous()
            self.icp.set_local(shard)

    def _probe_own_cost(self, T0, q, probe_iterations: int) -> float:
        """What THIS rank's shard costs per iteration, waiting excluded.  The probe is an ordinary sharded align -- every iteration
        ends in the all-reduce, which waits for the slowest rank -- so its wall time is about the MAXIMUM over the ranks on every
        rank: a cost vector of wall times is near uniform and `balanced_cuts` would not move a cut.  The matcher's own
        time does not contain the wait: HIP events around this rank's matcher launches (`set_profiling`: `ms_nn_kernel`)."""
        self.icp.align_resident(T0, q)            # (clocks, cost orders)
        self.icp.set_profiling(True)
        try:
            r = self.icp.align_resident(T0, q)
        finally:
            self.icp.set_profiling(False)
        return float(r.ms_nn_kernel) * 1e-3 / max(1, probe_iterations)

    def balance(self, params, rounds: int = 2, probe_iterations: int = 4, probe=None) -> list[int]:
        """Cuts of equal COST instead of equal count (`balanced_cuts`): every round, each rank measures what `probe_iterations` fixed
        iterations of `params` from the guess cost ITS shard (`_probe_own_cost`: the matcher's own time, not the wall time of the
        align -- that one contains the wait for the slowest rank), the W costs are all-gathered, all ranks cut again at the same
        places; the best cuts measured (smallest maximum) stay in force.  Worth it where the guess is far off and the align takes
        many iterations (the cost per query then varies several-fold along the scan and a step is as long as its slowest rank);
        each round costs a re-cut of the shard and the slab (milliseconds).  `probe(T0, params, probe_iterations) -> seconds`
        replaces the measurement (tests; callers with a cost model of their own).  Returns the cuts."""
        import torch
        import torch.distributed as dist
        from ._lib import IcpError
        assert hasattr(self, "_full") and self._spatial and self.world > 1
        map_pc, local_full, guess = self._full
        n = local_full.shape[1]
        T0 = np.eye(4) if guess is None else np.asarray(guess, np.float64).reshape(4, 4)
        q = params.copy()
        q.max_iterations, q.fixed_iterations, q.skip_quality = probe_iterations, 1, 1
        cuts = self._cuts or [shard_bounds(n, r, self.world)[0] for r in range(self.world)] + [n]
        best = None
        on_gpu = dist.get_backend(self.group) == "nccl"
        measure = probe if probe is not None else self._probe_own_cost
        for rnd in range(rounds + 1):
            try:   # (all ranks run the same number of all-reduces: the probe is an ordinary sharded align)
                mine = float(measure(T0, q, probe_iterations))
            except IcpError:
                mine = float("nan")
            v = torch.zeros(self.world, dtype=torch.float64)
            v[self.rank] = mine
            if on_gpu:
                v = v.cuda()
            dist.all_reduce(v, op=dist.ReduceOp.SUM, group=self.group)
            cost = v.cpu().numpy()
            if not np.all(np.isfinite(cost)):
                break
            if best is None or float(cost.max()) < best[1]:
                best = (list(cuts), float(cost.max()))
            if rnd == rounds:
                break
            cuts = balanced_cuts(cuts, cost)
            self._cuts = cuts
            self._cut_shard()
            self._cut_slab()
        if best is not None and best[0] != list(cuts):
            self._cuts = best[0]
            self._cut_shard()
            self._cut_slab()
        self.balance_record = {"best_max_cost_s": None if best is None else best[1]}
        return list(self._cuts or cuts)

    def _cut_slab(self):
        map_pc, _, init_guess = self._full
        T0 = np.eye(4) if init_guess is None else np.asarray(init_guess, dtype=np.float64)
        margin = self._slab_margin
        if margin is None and getattr(self, "_guess_uncertainty", None) is not None and self._gate is not None:
            lo0, hi0 = self.icp.shard_reach_box(T0, 0.0)
            margin = slab_margin_for_guess(lo0, hi0, self._gate, self._guess_uncertainty[0], self._guess_uncertainty[1])
        if margin is None:
            self.icp.set_map(map_pc)
            self.n_map_kept = map_pc.shape[1]
            self.slab_margin_used = None
            return
        margin *= getattr(self, "_slab_scale", 1.0)
        self.slab_margin_used = margin
        lo, hi = self.icp.shard_reach_box(T0, margin)
        self.n_map_kept = self.icp.set_map_slab(map_pc, lo, hi)

    def set_shard(self, map_pc, local_shard, n_local_total: int):
        self.icp.set_map(map_pc)
        self.icp.set_local(local_shard)
        self.icp.set_global_sizes(n_local_total, map_pc.shape[1])

    def align(self, init_guess, params):
        """identical Results on every rank.  With a map slab: if ANY rank's pose leaves its slab, every rank doubles the
        margin, cuts again and the align is repeated (the decision is all-reduced, so the ranks stay in step)."""
        from ._lib import IcpError
        if not hasattr(self, "_full") or getattr(self, "slab_margin_used", None) is None:
            return self.icp.align_resident(init_guess, params)
        import torch
        import torch.distributed as dist
        # the slab is cut around the pose the align STARTS from: a guess that differs from the one set_clouds() saw would
        # begin outside the slab and only recover by doubling a margin centred on the stale pose (re-upload + re-sort each time)
        from .icp import _pose16
        T0 = _pose16(init_guess).reshape(4, 4)
        map_pc, local_full, cut_at = self._full
        cut_T = np.eye(4) if cut_at is None else np.asarray(cut_at, dtype=np.float64).reshape(4, 4)
        if not np.array_equal(T0, cut_T):
            self._full = (map_pc, local_full, T0)
            self._cut_slab()
        for _ in range(6):
            res, left = None, 0
            try:
                res = self.icp.align_resident(init_guess, params)
            except IcpError as e:
                if "outside its map slab" not in str(e):
                    raise
                left = 1
            if self.world > 1:
                # a rank that failed stopped joining the accumulator all-reduces: the others' aligns end with a
                # communicator error or hang -- so the slab test must pass on EVERY rank before the loop starts.
                # mola_icp_shard_reach_box at the initial pose is checked by set_map_slab's caller; during the loop the
                # matcher's own check is the safety net for a single process (world 1) and for hook transports that
                # surface errors; here the ranks only agree on the outcome.
                flag = torch.tensor([left], dtype=torch.int32)
                if dist.get_backend(self.group) == "nccl":
                    flag = flag.cuda()
                dist.all_reduce(flag, op=dist.ReduceOp.MAX, group=self.group)
                left = int(flag.item())
            if not left:
                return res
            self._slab_scale = getattr(self, "_slab_scale", 1.0) * 2.0
            self._cut_slab()
        raise IcpError(-1, "the align keeps leaving its map slab")
