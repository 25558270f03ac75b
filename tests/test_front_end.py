"""the LidarOdometry front-end logic around the ICP
(csrc/lidar_odometry_core.cpp vs an independent restatement of src/LidarOdometry.cpp:190-514)."""
import os

import numpy as np
import pytest

from tests.helpers import ReferenceFrontEnd, compare_front_end_step, drive_scans, p2p_params

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _params(pkg, **kw):
    lp = pkg.LidarOdometryParams()
    lp.min_time_between_scans = 0.05
    lp.min_dist_xyz_between_keyframes = 0.5
    lp.min_rotation_between_keyframes = np.deg2rad(30)
    lp.min_icp_goodness = 0.05
    pw = p2p_params(pkg, max_iterations=30, matcher_threshold=0.8)
    pwo = p2p_params(pkg, max_iterations=30, matcher_threshold=1.5)
    lp.set_icp(pw, pwo)
    for k, v in kw.items():
        setattr(lp, k, v)
    return lp, pw, pwo


def _oracle_align(O):
    def align(f, t, T0, p):
        r = O.align(f, t, T0, O.params_from_product(p))
        return r["T"], r["quality"], r["n_iterations"], r["termination"]
    return align


def _sequence(synth):
    scans = drive_scans(synth)
    seq = list(scans[:3])
    seq.append((scans[2][0] + 0.01, scans[3][1]))                     # too soon: dropped by the time gate
    seq.append((scans[3][0], np.zeros((3, 0), np.float32)))           # empty cloud: ignored, but stored
    seq += scans[4:]                                                  # next scan sees an empty "last" cloud
    return seq


def test_front_end_equals_reference_logic_cpu(pkg, O, synth):
    lp, pw, pwo = _params(pkg)
    al = _oracle_align(O)
    lo = pkg.LidarOdometry(lp, align_fn=al)
    ref = ReferenceFrontEnd(lp.min_time_between_scans, lp.min_dist_xyz_between_keyframes,
                            lp.min_rotation_between_keyframes, lp.min_icp_goodness, pw, pwo,
                            lambda f, t, T0, p: al(f, t, T0, p)[:2], synth.pose_from_xyzypr)
    statuses, kfs, used = [], 0, []
    for t, cloud in _sequence(synth):
        s = lo.on_new_observation(t, cloud)
        r = ref.process(t, cloud)
        compare_front_end_step(s, r)
        statuses.append(s.status)
        kfs += s.keyframe_created
        used.append(s.used_with_vel_params if s.status == 2 else None)
    L = pkg._lib
    assert statuses == [L.LO_FIRST_SCAN, L.LO_ICP_RAN, L.LO_ICP_RAN, L.LO_DROPPED_TOO_SOON, L.LO_EMPTY_CLOUD,
                        L.LO_FIRST_SCAN, L.LO_ICP_RAN, L.LO_ICP_RAN]
    assert used[1] is False and used[2] is True          # the first ICP has no twist yet (cpp:287-290)
    assert kfs == 4   # origin KF, a distance KF (0.57 m > 0.5 m), the KF after the empty cloud, another distance KF
    # reset(): back to the initial state (cpp:160)
    lo.reset()
    s = lo.on_new_observation(0.0, _sequence(synth)[0][1])
    assert s.status == L.LO_FIRST_SCAN and s.keyframe_created and s.reference_kf == 0


def test_front_end_swaps_only_icp_parameters_not_the_icp_object(pkg, O, synth, tmp_path):
    """src/LidarOdometry.cpp:287-290 swaps `icp_in.icp_params` (mp2p_icp::Parameters: maxIterations, minAbsStep_*,
    pairingsWeightParameters) while cpp:869 always runs the AlignKind::LidarOdometry *object* (its matchers, solvers,
    quality evaluators).  With DIFFERENT with/without-vel files the first ICP (no twist yet) must therefore run
    icp_settings_with_vel's pipeline with icp_settings_without_vel's Parameters."""
    reg = open(os.path.join(ROOT, "params", "icp-settings-regular.yaml")).read()
    horn = open(os.path.join(ROOT, "params", "icp-settings-p2p-horn.yaml")).read()
    horn = horn.replace("maxIterations: 100", "maxIterations: 17").replace("minAbsStep_trans: 5e-5", "minAbsStep_trans: 2e-3")
    horn = horn.replace("use_scale_outlier_detector: true", "use_scale_outlier_detector: false")
    horn = horn.replace("threshold: 0.70", "threshold: 1.90").replace("thresholdDistance: 0.10", "thresholdDistance: 0.33")
    (tmp_path / "with.yaml").write_text(reg)
    (tmp_path / "without.yaml").write_text(horn)
    (tmp_path / "fe.yaml").write_text(
        "min_dist_xyz_between_keyframes: 0.5\nmin_time_between_scans: 0.05\nmin_icp_goodness: 0.05\n"
        "icp_settings_with_vel: $include{with.yaml}\nicp_settings_without_vel: $include{without.yaml}\n"
        "icp_settings_loop_closure: $include{with.yaml}\n")
    lp = pkg.LidarOdometryParams.load_from_file(str(tmp_path / "fe.yaml"), ROOT)
    pw, pwo = lp.icp_case("with_vel"), lp.icp_case("without_vel")
    assert pw.matcher_class == pkg._lib.MATCHER_POINT2PLANE and pwo.matcher_class == pkg._lib.MATCHER_POINTS_DISTANCE_THRESHOLD
    seen = []

    def align(f, t, T0, p):
        seen.append(p.copy())
        r = O.align(f, t, T0, O.params_from_product(p))
        return r["T"], r["quality"], r["n_iterations"], r["termination"]
    lo = pkg.LidarOdometry(lp, align_fn=align)
    ref = ReferenceFrontEnd(lp.min_time_between_scans, lp.min_dist_xyz_between_keyframes,
                            lp.min_rotation_between_keyframes, lp.min_icp_goodness, pw, pwo,
                            lambda f, t, T0, p: align(f, t, T0, p)[:2], synth.pose_from_xyzypr)
    for t, cloud in drive_scans(synth, n_scans=3, n_rings=16, n_az=240):
        s = lo.on_new_observation(t, cloud)
        a = seen[-1] if s.status == pkg._lib.LO_ICP_RAN else None
        compare_front_end_step(s, ref.process(t, cloud))
        b = seen[-1] if s.status == pkg._lib.LO_ICP_RAN else None
        if a is not None:   # product call and restatement call received the same flat parameter block
            assert bytes(a.c) == bytes(b.c)
    first, second = seen[0], seen[2]      # (product, restatement) x 2 ICP steps
    # first ICP: no twist -> the LidarOdometry object's pipeline (point-to-plane / GN / quality 0.10) ...
    assert first.matcher_class == pkg._lib.MATCHER_POINT2PLANE and first.solver_class == pkg._lib.SOLVER_GAUSS_NEWTON
    assert first.matcher_threshold == pytest.approx(0.70) and first.knn == 6 and first.quality_threshold == pytest.approx(0.10)
    # ... with the NearbyAlign case's mp2p_icp::Parameters
    assert first.max_iterations == 17 and first.min_abs_step_trans == pytest.approx(2e-3)
    assert first.use_scale_outlier_detector == 0
    # second ICP: twist is good -> the LidarOdometry case's own Parameters
    assert second.max_iterations == 100 and second.min_abs_step_trans == pytest.approx(5e-5)
    assert second.use_scale_outlier_detector == 1 and second.matcher_class == pkg._lib.MATCHER_POINT2PLANE
    # the composition helper of the C-ABI is what the front-end applied
    assert bytes(pkg.Parameters.compose(pw, pwo).c) == bytes(first.c)


def test_front_end_params_from_kitti_yaml(pkg):
    lp = pkg.LidarOdometryParams.load_from_file(os.path.join(ROOT, "params", "kitti-default.yaml"), ROOT)
    assert lp.min_time_between_scans == pytest.approx(0.01) and lp.min_dist_xyz_between_keyframes == pytest.approx(3.0)
    assert lp.min_icp_goodness == pytest.approx(0.50)
    assert lp.min_rotation_between_keyframes == pytest.approx(np.deg2rad(30))      # header default, YAML_LOAD_OPT_DEG
    assert lp.c.icp_with_vel.matcher_threshold == pytest.approx(0.70) and lp.c.icp_with_vel.max_iterations == 100
    # the reference's file selects its shipped pipeline for all three cases (kitti-default.yaml:43,46,50)
    for case in ("with_vel", "without_vel", "loop_closure"):
        c = lp.icp_case(case)
        assert c.matcher_class == pkg._lib.MATCHER_POINT2PLANE and c.solver_class == pkg._lib.SOLVER_GAUSS_NEWTON
        assert c.knn == 6 and c.solver_max_iterations == 20 and c.quality_threshold == pytest.approx(0.10)
    assert lp.min_icp_goodness_lc == pytest.approx(0.70) and lp.loop_closure_montecarlo_samples == 10
    assert (lp.min_dist_to_matching, lp.max_dist_to_matching, lp.max_dist_to_loop_closure) == (5.0, 20.0, 30.0)
    assert lp.max_nearby_align_checks == 5 and lp.min_topo_dist_to_consider_loopclosure == 30
    assert lp.max_kfs_local_graph == 50000                                          # header default (h:90)
    horn = pkg.LidarOdometryParams.load_from_file(os.path.join(ROOT, "params", "kitti-p2p-horn.yaml"), ROOT)
    assert horn.icp_case("with_vel").matcher_class == pkg._lib.MATCHER_POINTS_DISTANCE_THRESHOLD
    with pytest.raises(pkg.IcpError):
        pkg.LidarOdometryParams.load_from_file(os.path.join(ROOT, "params", "icp-settings-p2p-horn.yaml"), ROOT)


@pytest.mark.skipif(not os.path.isdir("/root/reference/params"), reason="reference tree not present on this box")
def test_kitti_default_yaml_equals_the_references(pkg):
    """the repo's params/kitti-default.yaml configures the front-end exactly as the reference's own file does"""
    ours = pkg.LidarOdometryParams.load_from_file(os.path.join(ROOT, "params", "kitti-default.yaml"), ROOT)
    ref = pkg.LidarOdometryParams.load_from_file("/root/reference/params/kitti-default.yaml", "/root/reference")
    assert bytes(ours.c) == bytes(ref.c)


def test_front_end_align_failure_propagates(pkg, synth):
    lp, _, _ = _params(pkg)

    def boom(f, t, T0, p):
        raise RuntimeError("icp exploded")
    lo = pkg.LidarOdometry(lp, align_fn=boom)
    scans = drive_scans(synth, n_scans=2)
    lo.on_new_observation(*scans[0])
    with pytest.raises(pkg.IcpError):
        lo.on_new_observation(*scans[1])


@pytest.mark.gpu
def test_front_end_on_gpu_equals_oracle_driven(pkg, O, synth):
    """the same drive with the MI355X ICP behind the front-end vs the oracle behind the reference logic"""
    lp, pw, pwo = _params(pkg)
    icp = pkg.ICP(device=0)
    lo = pkg.LidarOdometry(lp, icp=icp)
    al = _oracle_align(O)
    ref = ReferenceFrontEnd(lp.min_time_between_scans, lp.min_dist_xyz_between_keyframes,
                            lp.min_rotation_between_keyframes, lp.min_icp_goodness, pw, pwo,
                            lambda f, t, T0, p: al(f, t, T0, p)[:2], synth.pose_from_xyzypr)
    traj = np.eye(4)
    for t, cloud in drive_scans(synth, n_scans=6, n_rings=32, n_az=900):
        s = lo.on_new_observation(t, cloud)
        r = ref.process(t, cloud)
        compare_front_end_step(s, r, tol=1e-7)
        traj = traj @ s.rel_pose
    assert np.linalg.norm(traj[:3, 3]) > 0.5   # the vehicle moved
    lo.close()
    icp.close()
