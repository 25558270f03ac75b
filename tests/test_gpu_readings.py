"""-m gpu: the PRODUCT's reading switches (`mola_icp_params.reading_*`, YAML `readings:`) against the oracle run under the SAME
switch.  Three behaviours recalled from mp2p_icp are not pose- or goodness-neutral (tests/test_readings.py):
the scale-outlier detector's second pass, Point2Plane's ">= 3 vs ALL knn inside the gate" (params/icp-settings-regular.yaml:14-17,
33-39) and PairedRatio's denominator (icpreg:44-46).  If a real mp2p_icp disagrees with a default reading, the drop-in is fixed
by flipping a key, not by a rebuild."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def pair(synth):
    scene = synth.Scene(scene_seed=3, half=12.0, wall_y=5.0, wall_h=4.0, n_boxes=8)
    Tgt = synth.pose_from_xyzypr(0.30, -0.15, 0.04, np.deg2rad(1.5), np.deg2rad(-0.4), np.deg2rad(0.25))
    g, l, _ = synth.make_pair(12000, 10000, seed=11, T_gt=Tgt, scene=scene)     # N > M: the quality denominator matters
    return g, l


def _same(pkg, O, r, ref, tol=1e-7):
    rot, trans = O.pose_error(r.optimal_tf, ref["T"])
    assert r.nIterations == ref["n_iterations"] and r.terminationReason == ref["termination"], (r.nIterations, ref["n_iterations"])
    assert rot < tol and trans < tol, (rot, trans)
    assert r.quality == pytest.approx(ref["quality"], abs=1e-12)


# product field -> the oracle's switch of the same reading
SWITCHES = {"reading_outlier_single_pass": "outlier_single_pass", "reading_p2pl_all_inside_gate": "p2pl_all_inside_gate",
            "reading_quality_denominator_local": "quality_denominator"}


@pytest.mark.parametrize("field", list(SWITCHES) + [None])
def test_point_to_point_under_each_reading(pkg, O, pair, field):
    g, l = pair
    p = pkg.Parameters()
    p.max_iterations, p.matcher_threshold, p.min_abs_step_trans, p.min_abs_step_rot = 100, 1.0, 5e-5, 1e-5
    p.use_scale_outlier_detector, p.scale_outlier_threshold, p.quality_threshold = 1, 1.1, 0.10
    if field:
        setattr(p, field, 1)
    icp = pkg.ICP(device=0)
    try:
        if field:
            O.set_readings(**{SWITCHES[field]: 1})
        ref = O.align(g, l, np.eye(4), O.params_from_product(p))
        base = None
        if field:
            O.set_readings()
            base = O.align(g, l, np.eye(4), O.params_from_product(p))
    finally:
        O.set_readings()
    r = icp.align(g, l, np.eye(4), p)
    _same(pkg, O, r, ref)
    if field == "reading_outlier_single_pass":      # the switch DID change the run (else the test proves nothing)
        assert O.pose_error(ref["T"], base["T"])[1] > 1e-6
    if field == "reading_quality_denominator_local":
        assert abs(ref["quality"] - base["quality"]) > 1e-3
    icp.close()


@pytest.mark.parametrize("field", list(SWITCHES) + [None])
def test_shipped_pipeline_under_each_reading(pkg, O, pair, field):
    """params/icp-settings-regular.yaml (Point2Plane knn 6 + Gauss-Newton) with the optional `readings:` key"""
    g, l = pair
    text = open(os.path.join(ROOT, "params", "icp-settings-regular.yaml")).read()
    if field:
        text += "\nreadings:\n  %s: true\n" % field[len("reading_"):]
    p = pkg.Parameters.load_from(text)
    assert all(getattr(p, f) == (1 if f == field else 0) for f in SWITCHES)
    icp = pkg.ICP(device=0)
    try:
        if field:
            O.set_readings(**{SWITCHES[field]: 1})
        op = O.params_from_product(p)
        ref = O.align_p2pl(g, l, np.eye(4), op, p.plane_eigen_threshold, int(p.knn), int(p.solver_max_iterations))
        base = None
        if field:
            O.set_readings()
            base = O.align_p2pl(g, l, np.eye(4), op, p.plane_eigen_threshold, int(p.knn), int(p.solver_max_iterations))
    finally:
        O.set_readings()
    r = icp.align(g, l, np.eye(4), p)
    _same(pkg, O, r, ref)
    # the same align again on the state the first one left (cached planes decided under the SAME reading), and after the other
    # reading ran on the handle (cached planes of the other reading must not be reused)
    r2 = icp.align(g, l, np.eye(4), p)
    assert np.array_equal(r2.optimal_tf, r.optimal_tf)
    q = p.copy()
    q.reading_p2pl_all_inside_gate = 0 if p.reading_p2pl_all_inside_gate else 1
    icp.set_map(g); icp.set_local(l)
    icp.align_resident(np.eye(4), q)
    r3 = icp.align_resident(np.eye(4), p)
    assert np.array_equal(r3.optimal_tf, r.optimal_tf)
    if field == "reading_p2pl_all_inside_gate":
        assert O.pose_error(ref["T"], base["T"])[1] > 1e-7 or ref["n_pairs"] != base["n_pairs"]
    icp.close()


def test_batched_aligns_follow_the_readings(pkg, O, pair):
    """the lockstep batch (align_batch / align_multi_init) runs the same host logic: each result bit-equal to its stand-alone align"""
    g, l = pair
    text = open(os.path.join(ROOT, "params", "icp-settings-regular.yaml")).read() + "\nreadings:\n  p2pl_all_inside_gate: true\n  quality_denominator_local: true\n"
    p = pkg.Parameters.load_from(text)
    icp = pkg.ICP(device=0)
    alone = icp.align(g, l, np.eye(4), p)
    res = icp.align_batch([(g, l), (g, l)], [np.eye(4)] * 2, p)
    for r in res:
        assert np.array_equal(r.optimal_tf, alone.optimal_tf) and r.quality == alone.quality and r.nIterations == alone.nIterations
    pp = pkg.Parameters()
    pp.max_iterations, pp.matcher_threshold, pp.use_scale_outlier_detector, pp.scale_outlier_threshold = 60, 1.0, 1, 1.1
    pp.reading_outlier_single_pass = 1
    alone = icp.align(g, l, np.eye(4), pp)
    res = icp.align_batch([(g, l), (g, l)], [np.eye(4)] * 2, pp)
    for r in res:
        assert np.array_equal(r.optimal_tf, alone.optimal_tf) and r.nIterations == alone.nIterations
    icp.close()


def test_robust_kernel_flag_on_the_shipped_pipeline(pkg, O, pair):
    """`use_robust_kernel: true` in icp-settings-regular.yaml (its matcher is Matcher_Point2Plane): refused by name by default; under
    `readings: robust_kernel_skips_planes` the align is the flag-less align bit for bit -- the plane pairings keep unit weights --, stand-alone
    and in a lockstep batch, and equals the oracle's"""
    g, l = pair
    text = open(os.path.join(ROOT, "params", "icp-settings-regular.yaml")).read()
    base = pkg.Parameters.load_from(text)
    flagged = text.replace("use_robust_kernel: false", "use_robust_kernel: true")
    icp = pkg.ICP(device=0)
    with pytest.raises(pkg.IcpError) as ex:
        icp.align(g, l, np.eye(4), pkg.Parameters.load_from(flagged))
    assert ex.value.status == pkg._lib.E_UNSUPPORTED and "robust_kernel_skips_planes" in str(ex.value)
    p = pkg.Parameters.load_from(flagged + "\nreadings:\n  robust_kernel_skips_planes: true\n")
    ref = icp.align(g, l, np.eye(4), base)
    r = icp.align(g, l, np.eye(4), p)
    assert np.array_equal(r.optimal_tf, ref.optimal_tf) and r.nIterations == ref.nIterations and r.quality == ref.quality and r.n_pairs == ref.n_pairs
    for rb in icp.align_batch([(g, l), (g, l)], [np.eye(4)] * 2, p):
        assert np.array_equal(rb.optimal_tf, ref.optimal_tf) and rb.nIterations == ref.nIterations
    oref = O.align_p2pl(g, l, np.eye(4), O.params_from_product(p), p.plane_eigen_threshold, int(p.knn), p.solver_max_iterations)
    _same(pkg, O, r, oref)
    icp.close()
