"""SURVEY.md §8 row f2, the host policy around the nearby-keyframe / loop-closure ICPs (csrc/nearby_checks.cpp) against
independent restatements of src/LidarOdometry.cpp:572-599, 700-729 (selection), 767-783 (Monte-Carlo guesses) and
751-817 (the check and its accept test)."""
import numpy as np
import pytest


def ref_select(kfs, min_d, max_d, max_lc, max_checks, min_topo):
    """plain-Python restatement of checkForNearbyKFs' selection: a dict keyed by distance (later insertions replace
    earlier ones at the same key), the band, the per-candidate rules, stride thinning, closest loop closure"""
    by = {}
    for kid, d, topo, chk in kfs:
        by[d] = (kid, topo, chk)
    nearby, lcs = [], {}
    for d in sorted(by):
        if d < min_d or d > max(max_lc, max_d):
            continue
        kid, topo, chk = by[d]
        is_lc = topo >= min_topo
        if not is_lc and d > max_d:
            continue
        if chk:
            continue
        if is_lc:
            lcs[d] = kid
        else:
            nearby.append(kid)
    decim = max(1, len(nearby) // max_checks)
    return nearby[::decim], (lcs[min(lcs)] if lcs else None)


def test_select_checks_equals_restatement(pkg):
    rng = np.random.default_rng(3)
    for trial in range(200):
        lp = pkg.LidarOdometryParams()
        lp.min_dist_to_matching = float(rng.uniform(0, 8))
        lp.max_dist_to_matching = float(rng.uniform(5, 25))
        lp.max_dist_to_loop_closure = float(rng.uniform(5, 40))
        lp.max_nearby_align_checks = int(rng.integers(1, 6))
        lp.min_topo_dist_to_consider_loopclosure = int(rng.integers(1, 30))
        n = int(rng.integers(0, 40))
        d = rng.uniform(0, 45, n)
        if n > 4 and trial % 3 == 0:
            d[3] = d[1]                                   # exact distance duplicates collapse (map keyed by distance)
        if n > 0 and trial % 5 == 0:
            d[0] = lp.min_dist_to_matching                # band edges are inclusive on both sides
        if n > 1 and trial % 7 == 0:
            d[1] = max(lp.max_dist_to_loop_closure, lp.max_dist_to_matching)
        kfs = [(100 + i, float(d[i]), int(rng.integers(0, 40)), bool(rng.random() < 0.2)) for i in range(n)]
        got = pkg.select_checks(lp, kfs)
        exp = ref_select(kfs, lp.min_dist_to_matching, lp.max_dist_to_matching, lp.max_dist_to_loop_closure,
                         lp.max_nearby_align_checks, lp.min_topo_dist_to_consider_loopclosure)
        assert got == exp, (trial, kfs)


def test_select_checks_shipped_band(pkg):
    """kitti-default.yaml:35-39 -- 5..20 m nearby, loop closures (topological distance >= 30) up to 30 m, at most 5 checks"""
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    lp = pkg.LidarOdometryParams.load_from_file(os.path.join(root, "params", "kitti-default.yaml"), root)
    kfs = [(1, 3.0, 2, False), (2, 6.0, 4, False), (3, 8.0, 5, False), (4, 19.0, 9, False), (5, 25.0, 12, False),
           (6, 28.0, 31, False), (7, 12.0, 40, False), (8, 29.0, 50, True), (9, 31.0, 60, False)]
    nearby, lc = pkg.select_checks(lp, kfs)
    assert nearby == [2, 3, 4]          # 3 m is too close, 25 m only counts for loop closures
    assert lc == 7                      # the closest of the loop-closure candidates {7, 6}; 8 was checked, 9 is out of the band
    lp.max_nearby_align_checks = 1      # stride 3: only the closest goes out
    assert pkg.select_checks(lp, kfs)[0] == [2]
    lp.max_nearby_align_checks = 2      # stride 1 (3 // 2): all three -- the reference's "maximum" is a stride
    assert pkg.select_checks(lp, kfs)[0] == [2, 3, 4]


def test_montecarlo_guesses(pkg):
    g0 = np.array([3.0, -2.0, 0.5, 0.3, 0.02, -0.01])
    g6, gT = pkg.montecarlo_guesses(g0, 30.0, 20000, seed=11)
    assert g6.shape == (20000, 6) and gT.shape == (20000, 4, 4)
    d = g6 - g0
    assert np.all(d[:, 4:] == 0)                                               # pitch / roll are not perturbed (cpp:777-780)
    assert np.allclose(d[:, :3].std(axis=0), 3.0, rtol=0.03)                   # 0.1 * max_dist_to_loop_closure
    assert np.allclose(d[:, 3].std(), np.deg2rad(2.0), rtol=0.03)
    assert np.all(np.abs(d.mean(axis=0)[:3]) < 0.1) and abs(d[:, 3].mean()) < 1e-3
    assert abs(np.corrcoef(d[:, 0], d[:, 1])[0, 1]) < 0.03
    for i in (0, 7, 19999):
        assert np.allclose(gT[i], pkg.pose_from_xyzypr(g6[i]), atol=1e-15)
    again, _ = pkg.montecarlo_guesses(g0, 30.0, 10, seed=11)
    assert np.array_equal(again, g6[:10])                                      # seeded: reproducible
    other, _ = pkg.montecarlo_guesses(g0, 30.0, 10, seed=12)
    assert not np.array_equal(other, again)


def _lp(pkg, samples=6):
    from tests.helpers import p2p_params
    lp = pkg.LidarOdometryParams()
    near = p2p_params(pkg, max_iterations=30, matcher_threshold=0.8)
    lc = p2p_params(pkg, max_iterations=40, matcher_threshold=1.2)
    lp.set_icp(near, near, lc)
    lp.loop_closure_montecarlo_samples = samples
    lp.max_dist_to_loop_closure = 2.0          # sigma_xyz = 0.2 m
    lp.min_icp_goodness, lp.min_icp_goodness_lc = 0.02, 0.02
    return lp


def test_check_nonadjacent_cpu(pkg, O, synth, small_scene):
    """the check through an injected align function (the CPU oracle): nearby = one ICP with the NearbyAlign case; loop
    closure = K perturbed guesses with the LoopClosure case, first best goodness, the accept test on the LAST guess"""
    g, l, Tgt = synth.make_pair(3000, 3500, seed=21, scene=small_scene)
    calls = []

    def al(f, t, T0, p):
        r = O.align(f, t, T0, O.params_from_product(p))
        calls.append((T0.copy(), p.matcher_threshold, r["quality"]))
        return r["T"], r["quality"], r["n_iterations"], r["termination"]

    lp = _lp(pkg)
    g0 = pkg.pose_to_xyzypr(Tgt) + np.array([0.05, -0.03, 0.0, 0.004, 0, 0])
    r = pkg.check_nonadjacent(lp, g, l, g0, is_loop_closure=False, align_fn=al)
    assert len(calls) == 1 and calls[0][1] == 0.8 and r.n_attempts == 1 and r.best_guess == 0
    assert np.allclose(calls[0][0], pkg.pose_from_xyzypr(g0), atol=1e-15)
    d = np.linalg.inv(pkg.pose_from_xyzypr(g0)) @ r.icp.optimal_tf
    cp = np.linalg.norm(d[:3, 3]) / (np.linalg.norm(g0[:3]) + 0.01)
    assert r.correction_percent == pytest.approx(cp, rel=1e-12)
    assert r.edge_accepted == (r.icp.quality > 0.02 and cp < 0.2)
    assert r.edge_accepted                                                     # a good guess on a good pair

    calls.clear()
    r = pkg.check_nonadjacent(lp, g, l, g0, is_loop_closure=True, seed=5, align_fn=al)
    g6, gT = pkg.montecarlo_guesses(g0, 2.0, 6, seed=5)
    assert len(calls) == 6 and all(c[1] == 1.2 for c in calls) and r.n_attempts == 6
    for c, T in zip(calls, gT):
        assert np.allclose(c[0], T, atol=1e-15)
    q = [c[2] for c in calls]
    assert r.best_guess == int(np.argmax(q)) and r.icp.quality == max(q)      # argmax = the FIRST best (strictly greater)
    assert np.array_equal(r.init_guess_used, g6[-1])
    d = np.linalg.inv(gT[-1]) @ r.icp.optimal_tf
    assert r.correction_percent == pytest.approx(np.linalg.norm(d[:3, 3]) / (np.linalg.norm(g6[-1][:3]) + 0.01), rel=1e-12)
    assert r.edge_accepted == (r.icp.quality > 0.02)                            # loop closures skip the correction test

    # nothing better than goodness 0 -> no attempt is kept, the edge is refused
    far = g0 + np.array([500.0, 0, 0, 0, 0, 0])
    r = pkg.check_nonadjacent(lp, g, l, far, is_loop_closure=True, seed=5, align_fn=al)
    assert r.best_guess == -1 and r.icp.quality == 0.0 and not r.edge_accepted
    assert np.array_equal(r.icp.optimal_tf, np.eye(4))


@pytest.mark.gpu
def test_check_nonadjacent_gpu_batched_montecarlo(pkg, O, synth):
    """the loop-closure check on the GPU: the K guesses are ONE batched device problem; same result as the attempts run
    one after another through the oracle-shaped path"""
    g, l, Tgt = synth.make_pair(20000, 22000, seed=33)
    lp = _lp(pkg, samples=10)
    lp.max_dist_to_loop_closure = 3.0
    icp = pkg.ICP(device=0)
    g0 = pkg.pose_to_xyzypr(Tgt) + np.array([0.1, 0.05, 0.0, 0.01, 0, 0])
    r = pkg.check_nonadjacent(lp, g, l, g0, is_loop_closure=True, seed=9, icp=icp)
    _, gT = pkg.montecarlo_guesses(g0, 3.0, 10, seed=9)
    singles = [icp.align(g, l, T, lp.icp_case("loop_closure")) for T in gT]
    q = [s.quality for s in singles]
    assert r.n_attempts == 10 and r.best_guess == int(np.argmax(q))
    assert np.array_equal(r.icp.optimal_tf, singles[r.best_guess].optimal_tf) and r.icp.quality == max(q)
    ref = O.align(g, l, gT[r.best_guess], O.params_from_product(lp.icp_case("loop_closure")))
    rot, trans = O.pose_error(r.icp.optimal_tf, ref["T"])
    assert rot < 1e-7 and trans < 1e-7 and r.icp.nIterations == ref["n_iterations"]
    near = pkg.check_nonadjacent(lp, g, l, g0, is_loop_closure=False, icp=icp)
    s1 = icp.align(g, l, pkg.pose_from_xyzypr(g0), lp.icp_case("without_vel"))
    assert np.array_equal(near.icp.optimal_tf, s1.optimal_tf) and near.edge_accepted
    icp.close()
