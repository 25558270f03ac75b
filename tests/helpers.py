"""Shared test helpers: oracle-backed stages for the product's host loop."""
import numpy as np


class OracleStages:
    """match/accumulate stages computed by the CPU oracle over (a shard of) the local cloud.
    Plugged into the PRODUCT's loop (mola_icp_run_loop) to test its host logic without a GPU."""

    def __init__(self, O, map_pc, local_pc):
        self.O, self.g, self.l = O, map_pc, local_pc
        self.tree = O.KdTree(map_pc) if map_pc.shape[1] else None
        self.idx = self.d2 = self.outlier = None
        self.n_match = 0

    def match(self, T, thr):
        self.n_match += 1
        if self.l.shape[1] == 0 or self.g.shape[1] == 0:
            self.idx = np.full(self.l.shape[1], -1, np.int32)
            self.d2 = np.zeros(self.l.shape[1], np.float32)
            return 0
        self.idx, self.d2, n = self.O.match(self.g, self.l, T, thr, self.tree)
        return n

    def accumulate(self, p, T, stage, cl, cg, reset):
        if reset or self.outlier is None:
            self.outlier = np.zeros(max(1, self.l.shape[1]), np.uint8)
        if self.l.shape[1] == 0:
            return np.zeros(24)
        return self.O.accumulate(self.g, self.l, self.idx, self.d2, self.O.params_from_product(p), T, stage, cl, cg,
                                 self.outlier)


def p2p_params(pkg, **kw):
    p = pkg.Parameters()
    p.max_iterations = 40
    p.min_abs_step_trans = 5e-5
    p.min_abs_step_rot = 1e-5
    p.matcher_threshold = 1.0
    p.quality_threshold = 0.10
    for k, v in kw.items():
        setattr(p, k, v)
    return p


class ReferenceFrontEnd:
    """Independent restatement (numpy) of LidarOdometry::doProcessNewObservation()'s state machine
    (src/LidarOdometry.cpp:190-514, without back-end / world model): what the product's
    csrc/lidar_odometry_core.cpp must reproduce step by step.  `align(from, to, T0, params)` -> (T, quality)."""

    # mp2p_icp::Parameters proper (load_from(cfg["params"]) cpp:77-78; icpreg:10-21): the ONLY thing cpp:287-290 swaps.
    # Matchers / solvers / quality belong to the ICP object (cpp:80-87), which is always the LidarOdometry one (cpp:869).
    CALL_FIELDS = ("max_iterations", "min_abs_step_trans", "min_abs_step_rot", "use_scale_outlier_detector",
                   "scale_outlier_threshold", "use_robust_kernel", "robust_kernel_param", "robust_kernel_scale",
                   "fixed_iterations")

    def __init__(self, min_time, min_dist, min_rot, min_good, p_with, p_without, align, pose_from_xyzypr):
        self.min_time, self.min_dist, self.min_rot, self.min_good = min_time, min_dist, min_rot, min_good
        self.p_with, self.p_without, self.align, self.pose = p_with, p_without, align, pose_from_xyzypr
        self.last_tim = None
        self.last_pts = None
        self.twist = np.zeros(4)
        self.twist_good = False
        self.accum = np.eye(4)
        self.last_kf = None
        self.next_id = 0

    def process(self, t, pts):
        out = dict(status=None, kf=False, factor=None, rel=np.eye(4), quality=None, used_with=None)
        if self.last_tim is not None and (t - self.last_tim) < self.min_time:      # cpp:202-212
            out["status"] = 0
            return self._fin(out)
        last_tim, last_pts = self.last_tim, self.last_pts                          # cpp:229-234
        self.last_tim, self.last_pts = t, pts
        if pts.shape[1] == 0:                                                      # cpp:238-245
            out["status"] = 3
            return self._fin(out)
        if last_pts is None or last_pts.shape[1] == 0:                             # cpp:250-257
            out["status"] = 1
            create = True
        else:
            out["status"] = 2
            dt = t - last_tim                                                      # cpp:268-269
            guess = self.pose(self.twist[0] * dt, self.twist[1] * dt, self.twist[2] * dt, self.twist[3] * dt, 0, 0)
            p = self.p_with.copy()                                                 # the LidarOdometry ICP object, cpp:869
            if not self.twist_good:                                                # cpp:287-290: icpParameters only
                for f in self.CALL_FIELDS:
                    setattr(p, f, getattr(self.p_without, f))
            out["used_with"] = self.twist_good
            T, q = self.align(last_pts, pts, guess, p)                             # cpp:278-279, 299
            out["rel"], out["quality"], out["dt"] = T, q, dt
            yaw = np.arctan2(T[1, 0], T[0, 0])
            self.twist = np.array([T[0, 3] / dt, T[1, 3] / dt, T[2, 3] / dt, yaw / dt])   # cpp:305-311
            self.twist_good = True
            self.accum = self.accum @ T                                            # cpp:321
            dist = np.linalg.norm(self.accum[:3, 3])
            rot = np.arccos(np.clip((np.trace(self.accum[:3, :3]) - 1) / 2, -1, 1))
            out["dist"], out["rot"] = dist, rot
            create = q > self.min_good and (dist > self.min_dist or rot > self.min_rot)  # cpp:333-337
        if create:
            new_id = self.next_id
            self.next_id += 1
            out["kf"] = True
            if self.last_kf is not None:
                out["factor"] = (self.last_kf, new_id, self.accum.copy())          # cpp:436-443
            self.accum = np.eye(4)                                                 # cpp:472-474
            self.last_kf = new_id
        return self._fin(out)

    def _fin(self, out):
        out["accum"] = self.accum.copy()
        out["reference_kf"] = self.last_kf
        out["twist"] = self.twist.copy()
        return out


def compare_front_end_step(step, ref, tol=1e-9):
    """product Step vs ReferenceFrontEnd.process() output"""
    assert step.status == ref["status"]
    assert step.keyframe_created == ref["kf"]
    np.testing.assert_allclose(step.accum_since_last_kf, ref["accum"], atol=tol)
    assert (ref["reference_kf"] is None and not step.keyframe_created and step.reference_kf == 0) or \
        step.reference_kf == ref["reference_kf"]
    if ref["status"] == 2:
        np.testing.assert_allclose(step.rel_pose, ref["rel"], atol=tol)
        np.testing.assert_allclose(step.twist, ref["twist"], rtol=1e-7, atol=tol)
        assert step.used_with_vel_params == ref["used_with"]
        assert abs(step.dist_since_last_kf - ref["dist"]) < tol and abs(step.rot_since_last_kf - ref["rot"]) < 1e-7
        assert abs(step.icp.quality - ref["quality"]) < 1e-12
    if ref["factor"] is None:
        assert step.kf_factor is None
    else:
        assert step.kf_factor[0] == ref["factor"][0] and step.kf_factor[1] == ref["factor"][1]
        np.testing.assert_allclose(step.kf_factor[2], ref["factor"][2], atol=tol)


def drive_scans(synth, n_scans=7, n_rings=16, n_az=360, speed=4.0, yaw_rate=0.05, period=0.1, seed=3):
    """a short synthetic drive down the street canyon: [(timestamp, cloud(3,n) float32)]"""
    out = []
    for k in range(n_scans):
        t = 100.0 + k * period
        pose = synth.pose_from_xyzypr(-20.0 + speed * k * period, 0.3 * np.sin(0.3 * k), 0.0, yaw_rate * k * period, 0, 0)
        out.append((t, synth.lidar_scan(pose, n_rings=n_rings, n_az=n_az, seed=seed + k)))
    return out
