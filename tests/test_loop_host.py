"""The product's iteration-control loop (csrc/icp_loop.cpp, rows a1/a10/a11/a12) driven over
oracle-computed stages must reproduce the oracle's own align() -- on CPU, no GPU."""
import numpy as np
import pytest

from tests.helpers import OracleStages, p2p_params


def _run(pkg, O, g, l, p, T0=None, allreduce=None):
    st = OracleStages(O, g, l)
    r = pkg.run_loop(st.match, st.accumulate, np.eye(4) if T0 is None else T0, p, l.shape[1], g.shape[1], allreduce)
    return r, st


@pytest.mark.parametrize("kw", [
    dict(),
    dict(use_scale_outlier_detector=1, scale_outlier_threshold=1.1),
    dict(use_robust_kernel=1, robust_kernel_param=np.deg2rad(0.1), robust_kernel_scale=400.0),
    dict(use_scale_outlier_detector=1, scale_outlier_threshold=1.05, use_robust_kernel=1,
         robust_kernel_param=np.deg2rad(0.05), robust_kernel_scale=100.0),
    dict(fixed_iterations=1, max_iterations=12),
    dict(matcher_threshold=0.4, max_iterations=100),
])
def test_loop_equals_oracle_align(pkg, O, golden, kw):
    g, l = golden["A_map"], golden["A_local"]
    p = p2p_params(pkg, **kw)
    r, st = _run(pkg, O, g, l, p)
    ref = O.align(g, l, np.eye(4), O.params_from_product(p))
    assert r.nIterations == ref["n_iterations"]
    assert r.terminationReason == ref["termination"]
    np.testing.assert_allclose(r.optimal_tf, ref["T"], atol=1e-10)
    assert r.quality == pytest.approx(ref["quality"], abs=1e-12)
    assert r.n_pairs == ref["n_pairs"]
    assert r.rmse == pytest.approx(ref["rmse"], rel=1e-9)
    assert st.n_match == r.nIterations + 1  # one matcher pass per iteration + the quality pass


def test_loop_matches_golden(pkg, O, golden):
    g, l = golden["A_map"], golden["A_local"]
    r, _ = _run(pkg, O, g, l, p2p_params(pkg, max_iterations=30))
    assert r.nIterations == int(golden["A_nit"]) and r.terminationReason == int(golden["A_term"])
    np.testing.assert_allclose(r.optimal_tf, golden["A_Tfinal"], atol=1e-9)
    assert r.quality == pytest.approx(float(golden["A_quality"]), abs=1e-12)


def test_termination_reasons(pkg, O, golden):
    g, l = golden["A_map"], golden["A_local"]
    # MaxIterations
    r, _ = _run(pkg, O, g, l, p2p_params(pkg, max_iterations=3))
    assert r.terminationReason == pkg.TERM_MAX_ITERATIONS and r.nIterations == 3
    # Stalled at once when starting from the converged pose with loose thresholds
    r0, _ = _run(pkg, O, g, l, p2p_params(pkg, max_iterations=60))
    r, _ = _run(pkg, O, g, l, p2p_params(pkg, min_abs_step_trans=1e-2, min_abs_step_rot=1e-2), T0=r0.optimal_tf)
    assert r.terminationReason == pkg.TERM_STALLED and r.nIterations == 1
    # NoPairings: matcher disabled at iteration 0 / nothing inside the gate / empty clouds
    r, st = _run(pkg, O, g, l, p2p_params(pkg, run_from_iteration=2))
    assert r.terminationReason == pkg.TERM_NO_PAIRINGS and r.nIterations == 0
    assert np.array_equal(r.optimal_tf, np.eye(4))
    far = (l + np.float32(500)).astype(np.float32)
    r, _ = _run(pkg, O, g, far, p2p_params(pkg))
    assert r.terminationReason == pkg.TERM_NO_PAIRINGS and r.quality == 0 and r.n_pairs == 0
    empty = np.zeros((3, 0), np.float32)
    for gm, lm in ((empty, l), (g, empty)):
        r, st = _run(pkg, O, gm, lm, p2p_params(pkg))
        assert r.terminationReason == pkg.TERM_NO_PAIRINGS and r.quality == 0 and st.n_match == 0
    # max_iterations = 0: no iteration, quality still evaluated at the guess
    r, st = _run(pkg, O, g, l, p2p_params(pkg, max_iterations=0))
    assert r.nIterations == 0 and r.terminationReason == pkg.TERM_MAX_ITERATIONS and st.n_match == 1


def test_run_up_to_iteration(pkg, O, golden):
    g, l = golden["A_map"], golden["A_local"]
    p = p2p_params(pkg, run_up_to_iteration=2, max_iterations=10)
    r, _ = _run(pkg, O, g, l, p)
    ref = O.align(g, l, np.eye(4), O.params_from_product(p))
    assert (r.nIterations, r.terminationReason) == (ref["n_iterations"], ref["termination"]) == (3, pkg.TERM_NO_PAIRINGS)
    np.testing.assert_allclose(r.optimal_tf, ref["T"], atol=1e-10)


def test_covariance_is_sane(pkg, O, golden):
    g, l = golden["A_map"], golden["A_local"]
    r, _ = _run(pkg, O, g, l, p2p_params(pkg))
    C = r.optimal_tf_cov
    np.testing.assert_allclose(C, C.T, rtol=1e-9, atol=1e-18)
    assert (np.linalg.eigvalsh(C) > 0).all()
    # sigma ~ rmse/sqrt(3); translation std ~ sigma/sqrt(n)
    assert np.sqrt(C[0, 0]) == pytest.approx(r.rmse / np.sqrt(3) / np.sqrt(r.n_pairs), rel=0.5)


def test_bad_arguments(pkg, O, golden):
    g, l = golden["A_map"], golden["A_local"]
    for kw in (dict(matcher_threshold=0.0), dict(quality_threshold=-1.0), dict(matcher_class=7), dict(solver_class=9),
               dict(use_scale_outlier_detector=1, scale_outlier_threshold=0.5)):
        with pytest.raises(pkg.IcpError) as e:
            _run(pkg, O, g, l, p2p_params(pkg, **kw))
        assert e.value.status == pkg._lib.E_BADARG
    bad = np.eye(4)
    bad[0, 3] = np.nan
    with pytest.raises(pkg.IcpError):
        _run(pkg, O, g, l, p2p_params(pkg), T0=bad)


def test_stage_failure_propagates(pkg):
    def boom(T, thr):
        raise RuntimeError("stage exploded")
    with pytest.raises(pkg.IcpError):
        pkg.run_loop(boom, lambda *a: np.zeros(24), np.eye(4), p2p_params(pkg), 10, 10)

    def bad_reduce(acc):
        raise RuntimeError("link down")
    with pytest.raises(pkg.IcpError) as e:
        pkg.run_loop(lambda T, thr: 1, lambda *a: np.ones(24), np.eye(4), p2p_params(pkg), 10, 10, bad_reduce)
    assert e.value.status == pkg._lib.E_COMM


@pytest.mark.parametrize("kw", [
    dict(),
    dict(use_scale_outlier_detector=1, scale_outlier_threshold=1.1),
    dict(use_scale_outlier_detector=1, scale_outlier_threshold=1.05, use_robust_kernel=1,
         robust_kernel_param=np.deg2rad(0.05), robust_kernel_scale=100.0),
    dict(fixed_iterations=1, max_iterations=9),
    dict(skip_quality=1, max_iterations=25),
])
def test_lockstep_batch_loop_equals_single_loops(pkg, O, golden, synth, small_scene, kw):
    """mola_icp_run_loop_batch (the loop behind align_multi_init / align_batch: every stage issued for all problems
    still iterating) == one mola_icp_run_loop per problem, bit for bit -- different clouds, different initial poses,
    problems that stop at different iterations, an empty one and one with nothing inside the gate."""
    gA, lA = golden["A_map"], golden["A_local"]
    g2, l2, _ = synth.make_pair(3000, 2500, seed=23, scene=small_scene)
    far = np.ascontiguousarray(l2 + np.float32(400))
    empty = np.zeros((3, 0), np.float32)
    problems = [(gA, lA, np.eye(4)),
                (gA, lA, synth.pose_from_xyzypr(0.2, -0.1, 0.02, 0.01, 0, 0)),
                (g2, l2, np.eye(4)),
                (g2, far, np.eye(4)),          # NoPairings at iteration 0
                (g2, empty, np.eye(4)),        # empty local cloud
                (g2, l2, synth.pose_from_xyzypr(-0.1, 0.15, 0.0, -0.02, 0.003, 0.0))]
    p = p2p_params(pkg, **kw)
    singles = []
    for g, l, T0 in problems:
        r, _ = _run(pkg, O, g, l, p, T0)
        singles.append(r)
    stages = [OracleStages(O, g, l) for g, l, _ in problems]
    res = pkg.run_loop_batch([(st.match, st.accumulate, l.shape[1], g.shape[1]) for st, (g, l, _) in zip(stages, problems)],
                             [T0 for _, _, T0 in problems], p)
    if not kw.get("fixed_iterations"):
        assert len({r.nIterations for r in singles}) >= 3      # they really stop at different iterations
    for r, s, st in zip(res, singles, stages):
        assert r.nIterations == s.nIterations and r.terminationReason == s.terminationReason
        assert np.array_equal(r.optimal_tf, s.optimal_tf) and np.array_equal(r.optimal_tf_cov, s.optimal_tf_cov)
        assert r.quality == s.quality and r.n_pairs == s.n_pairs and r.rmse == s.rmse
    # a finished problem is not matched again: one matcher pass per iteration it ran (+ the quality pass)
    for r, st, (g, l, _) in zip(res, stages, problems):
        if l.shape[1] and r.terminationReason != pkg.TERM_NO_PAIRINGS:
            assert st.n_match == r.nIterations + (0 if kw.get("skip_quality") else 1)
    assert pkg.run_loop_batch([], [], p) == []
    pp = pkg.Parameters.load_from(open(__import__("os").path.join(__import__("os").path.dirname(__import__("os").path.dirname(
        __import__("os").path.abspath(__file__))), "params", "icp-settings-regular.yaml")).read())
    with pytest.raises(pkg.IcpError) as e:                     # the lockstep loop is point-to-point only
        pkg.run_loop_batch([(stages[0].match, stages[0].accumulate, 1, 1)], [np.eye(4)], pp)
    assert e.value.status == pkg._lib.E_UNSUPPORTED


@pytest.mark.parametrize("field,switch", [("reading_outlier_single_pass", "outlier_single_pass"),
                                           ("reading_quality_denominator_local", "quality_denominator")])
def test_loop_follows_the_reading_switches(pkg, O, golden, field, switch):
    """mola_icp_params.reading_* (ABI 4): the product's loop under a switch == the oracle under the same switch -- and the switch
    does change the run (more local points than map points, so that the quality denominator matters)"""
    g, l = np.ascontiguousarray(golden["A_map"][:, ::2]), golden["A_local"]
    assert l.shape[1] > g.shape[1]
    p = p2p_params(pkg, use_scale_outlier_detector=1, scale_outlier_threshold=1.1, max_iterations=60)
    setattr(p, field, 1)
    r, _ = _run(pkg, O, g, l, p)
    try:
        O.set_readings(**{switch: 1})
        ref = O.align(g, l, np.eye(4), O.params_from_product(p))
        O.set_readings()
        base = O.align(g, l, np.eye(4), O.params_from_product(p))
    finally:
        O.set_readings()
    assert (r.nIterations, r.terminationReason) == (ref["n_iterations"], ref["termination"])
    np.testing.assert_allclose(r.optimal_tf, ref["T"], atol=1e-10)
    assert r.quality == pytest.approx(ref["quality"], abs=1e-12)
    assert (not np.allclose(ref["T"], base["T"], atol=1e-9)) or abs(ref["quality"] - base["quality"]) > 1e-6


def test_readings_yaml_key(pkg):
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    text = open(os.path.join(root, "params", "icp-settings-regular.yaml")).read()
    p = pkg.Parameters.load_from(text)
    assert (p.reading_outlier_single_pass, p.reading_p2pl_all_inside_gate, p.reading_quality_denominator_local) == (0, 0, 0)
    p = pkg.Parameters.load_from(text + "\nreadings:\n  outlier_single_pass: true\n  p2pl_all_inside_gate: true\n  quality_denominator_local: false\n")
    assert (p.reading_outlier_single_pass, p.reading_p2pl_all_inside_gate, p.reading_quality_denominator_local) == (1, 1, 0)
    with pytest.raises(pkg.IcpError):
        pkg.Parameters.load_from(text + "\nreadings: 3\n")
    assert p.reading_robust_kernel_skips_planes == 0
    assert pkg.Parameters.load_from(text + "\nreadings:\n  robust_kernel_skips_planes: true\n").reading_robust_kernel_skips_planes == 1


def test_robust_kernel_with_plane_pairings_only_is_refused_by_name_or_read_as_having_nothing_to_act_on(pkg):
    """pairingsWeightParameters.use_robust_kernel flipped to true in the reference's OWN params block (icp-settings-regular.yaml:19; the
    matcher there is Matcher_Point2Plane): refused with a message that names the reading switch, accepted under it (ABI 5) -- the loop
    validates before it touches a cloud, so empty clouds tell the two apart (the last refusal a key of that block
    could trigger)"""
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    text = open(os.path.join(root, "params", "icp-settings-regular.yaml")).read()
    assert "use_robust_kernel: false" in text
    text = text.replace("use_robust_kernel: false", "use_robust_kernel: true")
    p = pkg.Parameters.load_from(text)
    assert p.use_robust_kernel == 1 and p.robust_kernel_scale == 400.0
    with pytest.raises(pkg.IcpError) as ex:
        pkg.run_loop(lambda T, thr: 0, lambda *a: np.zeros(24), np.eye(4), p, 0, 0)
    assert ex.value.status == pkg._lib.E_UNSUPPORTED and "robust_kernel_skips_planes" in str(ex.value)
    q = pkg.Parameters.load_from(text + "\nreadings:\n  robust_kernel_skips_planes: true\n")
    assert pkg.run_loop(lambda T, thr: 0, lambda *a: np.zeros(24), np.eye(4), q, 0, 0).terminationReason == pkg.TERM_NO_PAIRINGS
