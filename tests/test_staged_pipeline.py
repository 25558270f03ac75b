"""`matchers:` / `solvers:` with more than one entry (params/icp-settings-regular.yaml:28-31: "a sequence of one or more"),
staged by runFromIteration / runUpToIteration (icpreg:38-39): parsing, the rules that are refused by name, and the loop's
per-iteration selection -- on CPU through caller-supplied stages, on the GPU (p2p first, then the shipped point-to-plane
matcher) against the oracle run stage by stage."""
import os

import numpy as np
import pytest

from tests.helpers import OracleStages

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

STAGED = """
icp_class: mp2p_icp::ICP
params:
  maxIterations: 60
  minAbsStep_trans: 5e-5
  minAbsStep_rot: 1e-5
solvers:
  - class: mp2p_icp::Solver_Horn
    params:
      runFromIteration: 0
      runUpToIteration: %(last_p2p)d
  - class: mp2p_icp::Solver_GaussNewton
    params:
      maxIterations: 20
matchers:
  - class: mp2p_icp::Matcher_Points_DistanceThreshold
    params:
      threshold: 1.0
      runFromIteration: 0
      runUpToIteration: %(last_p2p)d
  - class: %(second)s
    params:
      %(second_params)s
      runFromIteration: %(first_second)d
      runUpToIteration: 0
quality:
  - class: mp2p_icp::QualityEvaluator_PairedRatio
    params:
      thresholdDistance: 0.10
"""
P2PL = dict(second="mp2p_icp::Matcher_Point2Plane", second_params="distanceThreshold: 0.70\n      planeEigenThreshold: 0.07\n      knn: 6")
P2P2 = dict(second="mp2p_icp::Matcher_Points_DistanceThreshold", second_params="threshold: 0.4")


def test_sequences_parse_into_entries(pkg):
    p = pkg.Parameters.load_from(STAGED % dict(last_p2p=2, first_second=3, **P2PL))
    assert p.n_extra_matchers == 1 and p.n_extra_solvers == 1
    assert p.matcher_class == pkg._lib.MATCHER_POINTS_DISTANCE_THRESHOLD and p.run_up_to_iteration == 2
    e = p.c.extra_matchers[0]
    assert e.matcher_class == pkg._lib.MATCHER_POINT2PLANE and e.matcher_threshold == pytest.approx(0.70) and e.knn == 6
    assert e.run_from_iteration == 3 and e.run_up_to_iteration == 0
    assert p.solver_class == pkg._lib.SOLVER_HORN and p.solver_run_up_to_iteration == 2
    assert p.c.extra_solvers[0].solver_class == pkg._lib.SOLVER_GAUSS_NEWTON and p.c.extra_solvers[0].solver_max_iterations == 20
    # the single-entry files of the reference still give no extra entries
    q = pkg.Parameters.load_from_file(os.path.join(ROOT, "params", "icp-settings-regular.yaml"))
    assert q.n_extra_matchers == 0 and q.n_extra_solvers == 0
    # more entries than the build holds: refused at load time, by count
    many = STAGED % dict(last_p2p=2, first_second=3, **P2PL)
    block = many[many.index("  - class: mp2p_icp::Matcher_Point2Plane"):many.index("quality:")]
    with pytest.raises(pkg.IcpError) as ex:
        pkg.Parameters.load_from(many.replace("quality:", block + block + block + "quality:"))
    assert ex.value.status == pkg._lib.E_CONFIG and "at most 4" in str(ex.value)


def test_overlapping_matchers_and_impossible_stage_pairs_are_refused(pkg):
    # two matchers active in the same iteration = mixed pairings in one solve: not run, said so
    p = pkg.Parameters.load_from(STAGED % dict(last_p2p=5, first_second=3, **P2PL))
    with pytest.raises(pkg.IcpError) as ex:
        pkg.run_loop(lambda T, thr: 0, lambda *a: np.zeros(24), np.eye(4), p, 10, 10)
    assert ex.value.status == pkg._lib.E_UNSUPPORTED and "active in iteration 3" in str(ex.value)
    # Point2Plane pairings meeting Solver_Horn in some iteration: the same rule as for single entries, found per stage
    txt = (STAGED % dict(last_p2p=2, first_second=3, **P2PL)).replace("runUpToIteration: 2\n  - class: mp2p_icp::Solver_GaussNewton",
                                                                       "runUpToIteration: 9\n  - class: mp2p_icp::Solver_GaussNewton")
    p = pkg.Parameters.load_from(txt)
    assert p.solver_run_up_to_iteration == 9
    with pytest.raises(pkg.IcpError) as ex:
        pkg.run_loop(lambda T, thr: 0, lambda *a: np.zeros(24), np.eye(4), p, 10, 10)
    assert ex.value.status == pkg._lib.E_UNSUPPORTED and "Solver_GaussNewton" in str(ex.value)
    # the lockstep batch runs single-entry pipelines
    ok = pkg.Parameters.load_from(STAGED % dict(last_p2p=2, first_second=3, **P2P2))
    with pytest.raises(pkg.IcpError) as ex:
        pkg.run_loop_batch([(lambda T, thr: 0, lambda *a: np.zeros(24), 1, 1)], [np.eye(4)], ok)
    assert ex.value.status == pkg._lib.E_UNSUPPORTED and "single-entry" in str(ex.value)


def test_loop_switches_matchers_by_iteration(pkg, O, golden):
    """two point-to-point matchers with different gates, iterations 0..2 and 3..: the product's loop over oracle stages sees
    the gate change at iteration 3 and ends where the oracle ends when it is run stage by stage"""
    g, l = golden["A_map"], golden["A_local"]
    p = pkg.Parameters.load_from(STAGED % dict(last_p2p=2, first_second=3, **P2P2))
    st = OracleStages(O, g, l)
    gates = []

    def match(T, thr):
        gates.append(thr)
        return st.match(T, thr)

    r = pkg.run_loop(match, st.accumulate, np.eye(4), p, l.shape[1], g.shape[1])
    assert r.nIterations > 4 and gates[:3] == [1.0, 1.0, 1.0] and all(t == 0.4 for t in gates[3:-1]) and gates[-1] == pytest.approx(0.10)
    a = O.align(g, l, np.eye(4), O.params(max_iterations=3, matcher_threshold=1.0, min_abs_step_trans=5e-5, min_abs_step_rot=1e-5, fixed_iterations=True))
    b = O.align(g, l, a["T"], O.params(max_iterations=57, matcher_threshold=0.4, min_abs_step_trans=5e-5, min_abs_step_rot=1e-5))
    assert r.nIterations == 3 + b["n_iterations"] and r.terminationReason == b["termination"]
    rot, trans = O.pose_error(r.optimal_tf, b["T"])
    assert rot < 1e-9 and trans < 1e-9
    # a gap in the ranges: no matcher active at iteration 3 -> the loop ends there as NoPairings (as a single matcher outside its range)
    gap = pkg.Parameters.load_from(STAGED % dict(last_p2p=2, first_second=4, **P2P2))
    r2 = pkg.run_loop(st.match, st.accumulate, np.eye(4), gap, l.shape[1], g.shape[1])
    assert r2.nIterations == 3 and r2.terminationReason == pkg.TERM_NO_PAIRINGS


@pytest.mark.gpu
def test_p2p_then_point2plane_on_the_gpu(pkg, O, synth):
    """the typical staged pipeline: point-to-point + Horn for iterations 0..2, then Matcher_Point2Plane + Gauss-Newton --
    against the oracle run stage by stage (its second stage starts from the first one's pose, as the loop's Tprev does)"""
    scene = synth.Scene(scene_seed=3, half=12.0, wall_y=5.0, wall_h=4.0, n_boxes=8)
    Tgt = synth.pose_from_xyzypr(0.30, -0.15, 0.04, np.deg2rad(1.5), np.deg2rad(-0.4), np.deg2rad(0.25))
    g, l, _ = synth.make_pair(30000, 26000, seed=11, T_gt=Tgt, scene=scene)
    p = pkg.Parameters.load_from(STAGED % dict(last_p2p=2, first_second=3, **P2PL))
    icp = pkg.ICP(device=0)
    r = icp.align(g, l, np.eye(4), p)
    a = O.align(g, l, np.eye(4), O.params(max_iterations=3, matcher_threshold=1.0, min_abs_step_trans=5e-5, min_abs_step_rot=1e-5, fixed_iterations=True))
    op = O.params(max_iterations=57, matcher_threshold=0.70, quality_threshold=0.10, min_abs_step_trans=5e-5, min_abs_step_rot=1e-5)
    b = O.align_p2pl(g, l, a["T"], op, 0.07, 6, 20)
    assert r.nIterations == 3 + b["n_iterations"] and r.terminationReason == b["termination"]
    rot, trans = O.pose_error(r.optimal_tf, b["T"])
    assert rot <= 1e-4 and trans <= 1e-3 and rot < 1e-7 and trans < 1e-7, (rot, trans)
    assert r.quality == pytest.approx(b["quality"], abs=1e-12) and r.n_pairs == b["n_pairs"]
    # staged pipelines through the batch entry points take the stand-alone path (same results)
    rb = icp.align_batch([(g, l), (g, l)], [np.eye(4), np.eye(4)], p)
    for x in rb:
        assert x.nIterations == r.nIterations and np.array_equal(x.optimal_tf, r.optimal_tf)
    icp.close()


def test_several_quality_evaluators_are_weighed(pkg, O, golden):
    """`quality:` is a sequence too (icpreg:40-46): Results::quality = sum w_i q_i / sum w_i -- one PairedRatio pass per entry"""
    g, l = golden["A_map"], golden["A_local"]
    txt = (STAGED % dict(last_p2p=2, first_second=3, **P2P2)).replace(
        "      thresholdDistance: 0.10\n",
        "      thresholdDistance: 0.10\n    weight: 1.0\n  - class: mp2p_icp::QualityEvaluator_PairedRatio\n    params:\n      thresholdDistance: 0.30\n    weight: 3.0\n")
    p = pkg.Parameters.load_from(txt)
    assert p.n_extra_quality == 1 and p.quality_weight == 1.0
    assert p.c.extra_quality[0].quality_threshold == pytest.approx(0.30) and p.c.extra_quality[0].weight == 3.0
    st = OracleStages(O, g, l)
    gates = []

    def match(T, thr):
        gates.append(thr)
        return st.match(T, thr)

    r = pkg.run_loop(match, st.accumulate, np.eye(4), p, l.shape[1], g.shape[1])
    assert gates[-2:] == [pytest.approx(0.10), pytest.approx(0.30)]        # the two quality passes, in order
    kd = O.KdTree(g)
    denom = min(l.shape[1], g.shape[1])
    q = [O.match(g, l, r.optimal_tf, thr, kd)[2] / denom for thr in (0.10, 0.30)]
    assert r.quality == pytest.approx((1.0 * q[0] + 3.0 * q[1]) / 4.0, abs=1e-12)
    # a single entry keeps its own ratio whatever its weight
    one = pkg.Parameters.load_from((STAGED % dict(last_p2p=2, first_second=3, **P2P2)).replace("      thresholdDistance: 0.10\n", "      thresholdDistance: 0.10\n    weight: 7.0\n"))
    r1 = pkg.run_loop(st.match, st.accumulate, np.eye(4), one, l.shape[1], g.shape[1])
    assert one.n_extra_quality == 0 and r1.quality == pytest.approx(q[0], abs=1e-12)
    # an unknown class in a later entry fails at load time, naming it
    with pytest.raises(pkg.IcpError) as ex:
        pkg.Parameters.load_from(txt.replace("mp2p_icp::QualityEvaluator_PairedRatio\n    params:\n      thresholdDistance: 0.30", "mp2p_icp::QualityEvaluator_Voxels\n    params:\n      thresholdDistance: 0.30"))
    assert ex.value.status == pkg._lib.E_CONFIG and "QualityEvaluator_Voxels" in str(ex.value)
