"""-m gpu: the HIP path through the C-ABI against the CPU oracle on the same seeded inputs.
Bars: NN indices and kept d2 bit-exact (integer/index work + the fp32 numeric contract);
fp64 accumulators rel 1e-10; poses within 1e-4 rad / 1e-3 m (north_star), in practice ~1e-9."""
import os
import threading

import numpy as np
import pytest

from tests.helpers import p2p_params

pytestmark = pytest.mark.gpu

ROT_TOL, TRANS_TOL = 1e-4, 1e-3  # BASELINE.json north_star: pose within 1e-4 rad / 1e-3 m of reference


@pytest.fixture(scope="module")
def icp(pkg):
    h = pkg.ICP(device=0)
    yield h
    h.close()


def _kernels(pkg):
    return [pkg.NN_VALU, pkg.NN_MFMA]


def _check_match(pkg, O, icp, g, l, T, thr, kern):
    icp.set_map(g)
    icp.set_local(l)
    idx, d2, n = icp.match(T, thr, l.shape[1], kern)
    oidx, od2, on = O.match(g, l, T, thr, O.KdTree(g) if g.shape[1] else None) if l.shape[1] and g.shape[1] else (
        np.full(l.shape[1], -1, np.int32), np.zeros(l.shape[1], np.float32), 0)
    assert n == on
    assert np.array_equal(idx, oidx), f"{(idx != oidx).sum()} of {len(idx)} NN indices differ"
    k = oidx >= 0
    assert np.array_equal(d2[k], od2[k])
    return idx, d2


@pytest.mark.parametrize("kern", [1, 2, 3])
def test_match_golden_bit_exact(pkg, O, icp, golden, kern):
    g, l = golden["A_map"], golden["A_local"]
    icp.set_map(g)
    icp.set_local(l)
    idx, d2, n = icp.match(np.eye(4), 1.0, l.shape[1], kern)
    assert np.array_equal(idx, golden["A_idx0"])
    k = idx >= 0
    assert np.array_equal(d2[k], golden["A_d20"][k]) and n == int(k.sum())


@pytest.mark.parametrize("kern", [1, 2, 3])
@pytest.mark.parametrize("N,M", [(1, 1), (1, 17), (63, 5), (64, 64), (257, 1023), (1000, 1025), (4099, 3001),
                                 (5000, 16), (30011, 20011)])
def test_match_ragged_sizes(pkg, O, icp, synth, small_scene, kern, N, M):
    g, l, Tgt = synth.make_pair(N, M, seed=100 + N + M, scene=small_scene)
    T = synth.pose_from_xyzypr(0.1, -0.05, 0.02, 0.01, 0.002, -0.003)
    _check_match(pkg, O, icp, g, l, T, 0.7, kern)


@pytest.mark.parametrize("kern", [1, 2, 3])
def test_match_config2_100k(pkg, O, icp, synth, kern):
    """BASELINE config 2: synthetic 100k-vs-100k, single-iteration correctness."""
    g, l, Tgt = synth.make_pair(100000, 100000, seed=42)
    idx, d2 = _check_match(pkg, O, icp, g, l, np.eye(4), 1.0, kern)
    p = p2p_params(pkg)
    acc = icp.accumulate(p, np.eye(4))
    oacc = O.accumulate(g, l, idx, d2, O.params_from_product(p), np.eye(4))
    np.testing.assert_allclose(acc, oacc, rtol=1e-10, atol=1e-6)
    rot, trans = O.pose_error(pkg.solve_horn(acc), O.horn(oacc))
    assert rot < 1e-10 and trans < 1e-9


@pytest.mark.parametrize("kern", [1, 2, 3])
def test_match_ties_lowest_index(pkg, O, icp, kern):
    ax = np.arange(8, dtype=np.float32)
    g = np.stack(np.meshgrid(ax, ax, ax, indexing="ij")).reshape(3, -1)
    g = np.ascontiguousarray(np.concatenate([g, g], axis=1))  # every map point duplicated
    l = np.ascontiguousarray((g[:, :512] + np.float32(0.5)).astype(np.float32))  # 8-way exact ties, twice
    idx, _ = _check_match(pkg, O, icp, g, l, np.eye(4), 2.0, kern)
    assert (idx < 512).all() and (idx >= 0).all()


@pytest.mark.parametrize("qpl", ["1", "2", "1-plain", "1-quads"])
def test_tiled_both_item_sizes(pkg, O, synth, small_scene, qpl, monkeypatch):
    """The tiled matcher's flavours, each forced on ragged sizes, exact ties, a warm-started second launch and a full align: 64-query
    items with the quad sweep (the default: per-quad tile lists, tiles straight into LDS), 128-query items, and 64-query items with the
    pass-by-pass sweep (MOLA_ICP_QUADS=0: what the diagnostic builds run)."""
    if qpl in ("1-plain", "1-quads"):   # (unset: the quad sweep for seeded launches only)
        monkeypatch.setenv("MOLA_ICP_QUADS", "0" if qpl == "1-plain" else "1")
        qpl = "1"
    monkeypatch.setenv("MOLA_ICP_QPL", qpl)
    monkeypatch.setenv("MOLA_ICP_COOP", "0")     # the persistent-wave kernel also at sizes where the cooperative one is the default
    pkg._lib.lib().mola_icp_debug_reload_env()
    icp = pkg.ICP(device=0)
    T = synth.pose_from_xyzypr(0.1, -0.05, 0.02, 0.01, 0.002, -0.003)
    for N, M in ((63, 5), (129, 4000), (9000, 9000), (30011, 20011)):
        g, l, _ = synth.make_pair(N, M, seed=7 + N, scene=small_scene)
        _check_match(pkg, O, icp, g, l, T, 0.7, pkg.NN_TILED)
        idx, d2, n = icp.match(np.eye(4), 0.7, N, pkg.NN_TILED)          # seeded by the launch above
        oidx, od2, on = O.match(g, l, np.eye(4), 0.7, O.KdTree(g))
        assert n == on and np.array_equal(idx, oidx) and np.array_equal(d2[oidx >= 0], od2[oidx >= 0])
    ax = np.arange(8, dtype=np.float32)
    g = np.stack(np.meshgrid(ax, ax, ax, indexing="ij")).reshape(3, -1)
    g = np.ascontiguousarray(np.concatenate([g, g], axis=1))
    l = np.ascontiguousarray((g[:, :512] + np.float32(0.5)).astype(np.float32))
    idx, _ = _check_match(pkg, O, icp, g, l, np.eye(4), 2.0, pkg.NN_TILED)
    assert (idx < 512).all() and (idx >= 0).all()
    g, l, _ = synth.make_pair(20000, 20000, seed=5, scene=small_scene)
    p = p2p_params(pkg, max_iterations=30)
    p.nn_kernel = pkg.NN_TILED
    r = icp.align(g, l, np.eye(4), p)
    ref = O.align(g, l, np.eye(4), O.params_from_product(p))
    assert r.nIterations == ref["n_iterations"] and r.n_pairs == ref["n_pairs"]
    rot, trans = O.pose_error(r.optimal_tf, ref["T"])
    assert rot < 1e-7 and trans < 1e-9
    icp.close()
    monkeypatch.delenv("MOLA_ICP_QPL")
    monkeypatch.delenv("MOLA_ICP_COOP")
    monkeypatch.delenv("MOLA_ICP_QUADS", raising=False)
    pkg._lib.lib().mola_icp_debug_reload_env()


@pytest.mark.parametrize("coop", ["0", "1"])
def test_tiled_cooperative_equals_persistent(pkg, O, synth, small_scene, coop, monkeypatch):
    """k_nn_coop (one workgroup per 128-query item, the default below ~0.4M queries) and k_nn_tiled (persistent waves)
    are the same matcher: both forced here on ragged sizes, exact ties (the exact flavour over the queued items), a
    warm-started second launch at another pose, and a 300k-query cloud (more workgroups than resident slots)."""
    monkeypatch.setenv("MOLA_ICP_COOP", coop)
    pkg._lib.lib().mola_icp_debug_reload_env()
    try:
        icp = pkg.ICP(device=0)
        T = synth.pose_from_xyzypr(0.1, -0.05, 0.02, 0.01, 0.002, -0.003)
        for N, M in ((1, 1), (63, 5), (129, 4000), (9000, 9000), (30011, 20011), (300_007, 150_001)):
            g, l, _ = synth.make_pair(N, M, seed=7 + N, scene=small_scene if N < 100_000 else None)
            tree = O.KdTree(g)
            icp.set_map(g)
            icp.set_local(l)
            for Tq in (T, np.eye(4), T):                   # launches 2 and 3 are seeded by the one before
                idx, d2, n = icp.match(Tq, 0.7, N, pkg.NN_TILED)
                oidx, od2, on = O.match(g, l, Tq, 0.7, tree)
                assert n == on and np.array_equal(idx, oidx), (N, M, int((idx != oidx).sum()))
                assert np.array_equal(d2[oidx >= 0], od2[oidx >= 0])
        ax = np.arange(8, dtype=np.float32)
        g = np.stack(np.meshgrid(ax, ax, ax, indexing="ij")).reshape(3, -1)
        g = np.ascontiguousarray(np.concatenate([g, g], axis=1))        # every map point twice: 16-way exact ties
        l = np.ascontiguousarray((g[:, :512] + np.float32(0.5)).astype(np.float32))
        idx, _ = _check_match(pkg, O, icp, g, l, np.eye(4), 2.0, pkg.NN_TILED)
        assert (idx < 512).all() and (idx >= 0).all()
        idx2, _, _ = icp.match(np.eye(4), 2.0, l.shape[1], pkg.NN_TILED)   # seeded, still the lowest index
        assert np.array_equal(idx, idx2)
        g, l, _ = synth.make_pair(20000, 20000, seed=5, scene=small_scene)
        p = p2p_params(pkg, max_iterations=30)
        p.nn_kernel = pkg.NN_TILED
        icp.set_profiling(True)
        r = icp.align(g, l, np.eye(4), p)
        ref = O.align(g, l, np.eye(4), O.params_from_product(p))
        assert r.nIterations == ref["n_iterations"] and r.n_pairs == ref["n_pairs"]
        rot, trans = O.pose_error(r.optimal_tf, ref["T"])
        assert rot < 1e-7 and trans < 1e-9
        assert r.nn_pairs_evaluated > 0
        icp.close()
    finally:
        monkeypatch.delenv("MOLA_ICP_COOP")
        pkg._lib.lib().mola_icp_debug_reload_env()


@pytest.mark.parametrize("kern", [1, 2, 3])
def test_match_far_from_origin_and_gate_edges(pkg, O, icp, synth, small_scene, kern):
    # clouds 5 km from the origin (fp32 cancellation territory for an expanded-form distance)
    g, l, _ = synth.make_pair(3000, 3000, seed=9, scene=small_scene)
    off = np.array([[5000.0], [-3000.0], [100.0]], dtype=np.float32)
    _check_match(pkg, O, icp, np.ascontiguousarray(g + off), np.ascontiguousarray(l + off), np.eye(4), 0.5, kern)
    # gate exactly at a pair distance: strict '<'
    g1 = np.array([[0.0], [0.0], [0.0]], np.float32)
    l1 = np.array([[0.5, 0.25], [0.0, 0.0], [0.0, 0.0]], np.float32)
    icp.set_map(g1)
    icp.set_local(l1)
    idx, d2, n = icp.match(np.eye(4), 0.5, 2, kern)
    assert list(idx) == [-1, 0] and n == 1


@pytest.mark.parametrize("kw", [dict(), dict(use_scale_outlier_detector=1, scale_outlier_threshold=1.1),
                                dict(use_robust_kernel=1, robust_kernel_param=np.deg2rad(0.1)),
                                dict(use_scale_outlier_detector=1, scale_outlier_threshold=1.05, use_robust_kernel=1,
                                     robust_kernel_param=np.deg2rad(0.05), robust_kernel_scale=100.0)])
def test_accumulate_stages(pkg, O, icp, golden, kw):
    g, l = golden["A_map"], golden["A_local"]
    p = p2p_params(pkg, **kw)
    op = O.params_from_product(p)
    T = golden["A_trace"][1]
    icp.set_map(g)
    icp.set_local(l)
    idx, d2, _ = icp.match(T, 1.0, l.shape[1])
    acc0 = icp.accumulate(p, T, stage=0, reset_outliers=True)
    outl = np.zeros(l.shape[1], np.uint8)
    o0 = O.accumulate(g, l, idx, d2, op, T, 0, None, None, outl)
    np.testing.assert_allclose(acc0, o0, rtol=1e-11, atol=1e-7)
    cl, cg = o0[1:4] / o0[0], o0[4:7] / o0[0]
    acc1 = icp.accumulate(p, T, stage=1, cl=cl, cg=cg, reset_outliers=False)
    o1 = O.accumulate(g, l, idx, d2, op, T, 1, cl, cg, outl)
    np.testing.assert_allclose(acc1, o1, rtol=1e-10, atol=1e-7)
    # the outlier flags persist: a second unit-weight pass skips them
    acc2 = icp.accumulate(p, T, stage=0, reset_outliers=False)
    o2 = O.accumulate(g, l, idx, d2, op, T, 0, None, None, outl)
    np.testing.assert_allclose(acc2, o2, rtol=1e-11, atol=1e-7)
    assert acc2[16] == o2[16] <= acc0[16]


def test_accumulate_deterministic(pkg, icp, golden):
    g, l = golden["A_map"], golden["A_local"]
    icp.set_map(g)
    icp.set_local(l)
    icp.match(np.eye(4), 1.0, l.shape[1], copy=False)
    p = p2p_params(pkg)
    a = icp.accumulate(p, np.eye(4))
    for _ in range(5):
        assert np.array_equal(a, icp.accumulate(p, np.eye(4)))  # bitwise run-to-run


@pytest.mark.parametrize("kw", [dict(max_iterations=30), dict(max_iterations=60, matcher_threshold=0.5),
                                dict(max_iterations=30, use_scale_outlier_detector=1, scale_outlier_threshold=1.1),
                                dict(max_iterations=10, fixed_iterations=1),
                                dict(max_iterations=30, use_robust_kernel=1, robust_kernel_param=np.deg2rad(0.1))])
def test_align_equals_oracle(pkg, O, icp, golden, kw):
    g, l = golden["A_map"], golden["A_local"]
    p = p2p_params(pkg, **kw)
    icp.set_profiling(True)          # per-launch HIP events + executed-pair counters (off by default)
    r = icp.align(g, l, np.eye(4), p)
    ref = O.align(g, l, np.eye(4), O.params_from_product(p))
    assert r.nIterations == ref["n_iterations"] and r.terminationReason == ref["termination"]
    rot, trans = O.pose_error(r.optimal_tf, ref["T"])
    assert rot <= ROT_TOL and trans <= TRANS_TOL
    assert rot < 1e-8 and trans < 1e-8  # what the exact-NN design actually delivers
    assert r.quality == pytest.approx(ref["quality"], abs=1e-12)
    assert r.n_pairs == ref["n_pairs"] and r.rmse == pytest.approx(ref["rmse"], rel=1e-9)
    assert r.n_nn_launches == r.nIterations + 1 and r.ms_nn_kernel > 0
    icp.set_profiling(False)
    r0 = icp.align(g, l, np.eye(4), p)
    assert r0.n_nn_launches == r.n_nn_launches and r0.ms_nn_kernel == 0 and np.array_equal(r0.optimal_tf, r.optimal_tf)


def test_align_golden_result(pkg, icp, golden):
    r = icp.align(golden["A_map"], golden["A_local"], np.eye(4), p2p_params(pkg, max_iterations=30))
    assert r.nIterations == int(golden["A_nit"]) and r.terminationReason == int(golden["A_term"])
    np.testing.assert_allclose(r.optimal_tf, golden["A_Tfinal"], atol=1e-8)
    assert r.quality == pytest.approx(float(golden["A_quality"]), abs=1e-12)


@pytest.mark.parametrize("name", ["identity", "trans", "yaw", "pitch", "roll", "se3"])
def test_known_answer_transforms(pkg, O, icp, golden, name):
    g, l = golden["B_map"], golden[f"B_{name}_local"]
    r = icp.align(g, l, np.eye(4), p2p_params(pkg, max_iterations=100, matcher_threshold=0.5))
    rot, trans = O.pose_error(r.optimal_tf, golden[f"B_{name}_T"])
    assert rot <= ROT_TOL and trans <= TRANS_TOL, (rot, trans)
    assert r.quality > 0.99


def test_pose_convention_to_wrt_from(pkg, icp, synth, small_scene):
    # g ~= T (+) l : include/mola-fe-lidar/LidarOdometry.h:122,131
    Tgt = synth.pose_from_xyzypr(0.15, 0.1, -0.02, 0.015, 0.0, 0.0)
    g, l, _ = synth.make_pair(20000, 20000, seed=5, T_gt=Tgt, scene=small_scene, noise_sigma=0.0)
    r = icp.align(g, l, pkg.pose_to_xyzypr(np.eye(4)), p2p_params(pkg, max_iterations=100, matcher_threshold=0.5))
    assert np.abs(r.optimal_tf[:3, 3] - Tgt[:3, 3]).max() < 0.02
    assert abs(pkg.pose_to_xyzypr(r.optimal_tf)[3] - 0.015) < 2e-3


def test_edge_cases(pkg, icp, golden):
    g, l = golden["A_map"], golden["A_local"]
    empty = np.zeros((3, 0), np.float32)
    for gm, lm in ((empty, l), (g, empty), (empty, empty)):
        r = icp.align(gm, lm, np.eye(4), p2p_params(pkg))
        assert r.terminationReason == pkg.TERM_NO_PAIRINGS and r.nIterations == 0 and r.quality == 0
        assert np.array_equal(r.optimal_tf, np.eye(4))
    far = np.ascontiguousarray(l + np.float32(1000))
    r = icp.align(g, far, np.eye(4), p2p_params(pkg, matcher_threshold=0.5))
    assert r.terminationReason == pkg.TERM_NO_PAIRINGS and r.n_pairs == 0 and r.quality == 0
    with pytest.raises(pkg.IcpError):
        icp.align(g, l, np.full((4, 4), np.nan), p2p_params(pkg))
    bad = p2p_params(pkg)
    bad.matcher_class = pkg._lib.MATCHER_POINT2PLANE   # plane pairings with Solver_Horn: rejected by name
    with pytest.raises(pkg.IcpError) as e:
        icp.align(g, l, np.eye(4), bad)
    assert e.value.status == pkg._lib.E_UNSUPPORTED and "Solver_GaussNewton" in str(e.value)
    # the handle survives errors
    assert icp.align(g, l, np.eye(4), p2p_params(pkg, max_iterations=2)).nIterations == 2


def test_dropped_clouds_park_their_device_blocks(pkg, synth, small_scene):
    """A stream of put / align / drop (an odometry drive drops one cloud per scan): the dropped clouds' device blocks are parked and
    handed to the next cloud -- results are those of fresh allocations, the parked amount stays bounded, and the trim call frees it."""
    pkg.ICP.device_pool_trim(0, 0)
    icp = pkg.ICP(device=0)
    p = p2p_params(pkg, max_iterations=10, matcher_threshold=0.6)
    ref = pkg.ICP(device=0)
    clouds = [synth.make_pair(9000 + 37 * k, 9000 + 37 * k, seed=500 + k, scene=small_scene)[1] for k in range(8)]
    parked = []
    icp.cloud_put(0, clouds[0])
    for k in range(1, 8):
        icp.cloud_put(k, clouds[k])
        r = icp.align_cached(k - 1, k, np.eye(4), p)
        s = ref.align(clouds[k - 1], clouds[k], np.eye(4), p)
        assert r.nIterations == s.nIterations and np.array_equal(r.optimal_tf, s.optimal_tf) and r.quality == s.quality, k
        icp.cloud_drop(k - 1)
        parked.append(pkg.ICP.device_pool_trim(0, 1 << 40))    # (a limit nothing reaches: just reads the parked amount)
    # (a dropped cloud lives until the workspace that used it last lets go of it: one align later)
    assert parked[-1] > 0                         # dropped clouds' blocks are parked ...
    assert max(parked) <= 2 * parked[1] + (1 << 20)   # ... and later clouds take them instead of piling up new ones (one cloud's worth, padded, stays)
    assert pkg.ICP.device_pool_trim(0, 0) == 0
    icp.cloud_put(99, clouds[0])                  # (after a trim: plain allocations again)
    assert icp.align_cached(99, 7, np.eye(4), p).nIterations >= 1


def test_non_finite_coordinates_are_refused(pkg, golden):
    """The bounding box of a new cloud is no longer waited for (its sort reads it on the device); the host checks its copy at the
    next wait it makes anyway: a NaN / inf coordinate must still fail the call it came with, through every way a cloud gets in."""
    import os
    g, l = golden["A_map"], golden["A_local"]
    shipped = pkg.Parameters.load_from_file(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "params", "icp-settings-regular.yaml"))
    # the tiled matcher needs >= 8192 points a side: tile the golden clouds
    reps = 8192 // min(g.shape[1], l.shape[1]) + 1
    big_g = np.ascontiguousarray(np.concatenate([g + np.float32(0.001 * k) for k in range(reps)], axis=1))
    big_l = np.ascontiguousarray(np.concatenate([l + np.float32(0.001 * k) for k in range(reps)], axis=1))
    for params in (p2p_params(pkg), shipped):
        for which in ("map", "local"):
            for bad_value in (np.nan, np.inf):
                icp = pkg.ICP(device=0)
                a, b = big_g.copy(), big_l.copy()
                (a if which == "map" else b)[1, 77] = bad_value
                with pytest.raises(pkg.IcpError) as e:
                    icp.align(a, b, np.eye(4), params)
                assert e.value.status == pkg._lib.E_BADARG and "non-finite" in str(e.value), (which, bad_value)
                try:                                 # ... and nothing is left on the device that a resident align could run on
                    rr = icp.align_resident(np.eye(4), params)
                    assert rr.nIterations == 0 and rr.n_pairs == 0
                except pkg.IcpError:
                    pass
                # ... and the handle is usable afterwards
                r = icp.align(big_g, big_l, np.eye(4), params)
                assert r.nIterations >= 1
    icp = pkg.ICP(device=0)
    a = big_g.copy()
    a[2, 5] = np.nan
    with pytest.raises(pkg.IcpError) as e:
        icp.cloud_put(1, a)
    assert e.value.status == pkg._lib.E_BADARG
    icp.cloud_put(1, big_g)
    icp.cloud_put(2, big_l)
    assert icp.align_cached(1, 2, np.eye(4), shipped).nIterations >= 1


def test_align_is_reentrant_per_handle(pkg, O, icp, golden, synth, small_scene):
    """>=4 threads on ONE handle with DIFFERENT per-call params, as the reference's pool does
    (src/LidarOdometry.cpp:94-96, 287-290, 869)."""
    g, l = golden["A_map"], golden["A_local"]
    g2, l2, _ = synth.make_pair(7000, 9000, seed=77, scene=small_scene)
    jobs = [(g, l, p2p_params(pkg, max_iterations=30)), (g2, l2, p2p_params(pkg, max_iterations=25, matcher_threshold=0.6)),
            (g, l, p2p_params(pkg, max_iterations=12, fixed_iterations=1)),
            (g2, l2, p2p_params(pkg, max_iterations=30, use_scale_outlier_detector=1, scale_outlier_threshold=1.1)),
            (g, l, p2p_params(pkg, max_iterations=30)), (g2, l2, p2p_params(pkg, max_iterations=8))]
    refs = [O.align(a, b, np.eye(4), O.params_from_product(p)) for a, b, p in jobs]
    out = [None] * len(jobs)

    def run(i):
        for _ in range(3):
            out[i] = icp.align(jobs[i][0], jobs[i][1], np.eye(4), jobs[i][2])
    th = [threading.Thread(target=run, args=(i,)) for i in range(len(jobs))]
    [t.start() for t in th]
    [t.join() for t in th]
    for r, ref in zip(out, refs):
        assert r is not None and r.nIterations == ref["n_iterations"]
        rot, trans = O.pose_error(r.optimal_tf, ref["T"])
        assert rot < 1e-8 and trans < 1e-8


def test_align_batch(pkg, O, icp, synth, small_scene):
    """config 4's shape: independent pairs, stream-per-pair, no collective."""
    pairs, refs = [], []
    p = p2p_params(pkg, max_iterations=40, matcher_threshold=0.6)
    for s in range(10):
        g, l, _ = synth.make_pair(3000 + 100 * s, 2500 + 50 * s, seed=200 + s, scene=small_scene)
        pairs.append((g, l))
        refs.append(O.align(g, l, np.eye(4), O.params_from_product(p)))
    res = icp.align_batch(pairs, [np.eye(4)] * len(pairs), p)
    for r, ref in zip(res, refs):
        assert r.nIterations == ref["n_iterations"] and r.terminationReason == ref["termination"]
        rot, trans = O.pose_error(r.optimal_tf, ref["T"])
        assert rot < 1e-8 and trans < 1e-8
    assert icp.align_batch([], [], p) == []


def test_sequential_shards_equal_full(pkg, icp, synth):
    """1/2/4/8-way query sharding run sequentially on one GPU: summed shard accumulators ==
    full accumulators -- the reduction the 8-GPU path performs over RCCL."""
    import importlib
    sharded = importlib.import_module("mola-fe-lidar_amd.sharded")
    g, l, _ = synth.make_pair(50000, 40000, seed=8)
    p = p2p_params(pkg)
    icp.set_map(g)
    icp.set_local(l)
    full_idx, _, _ = icp.match(np.eye(4), 1.0, l.shape[1])
    full = icp.accumulate(p, np.eye(4))
    for world in (2, 4, 8):
        tot = np.zeros(24)
        idxs = []
        for r in range(world):
            lo, hi = sharded.shard_bounds(l.shape[1], r, world)
            icp.set_local(np.ascontiguousarray(l[:, lo:hi]))
            i, _, _ = icp.match(np.eye(4), 1.0, hi - lo)
            idxs.append(i)
            tot += icp.accumulate(p, np.eye(4))
        assert np.array_equal(np.concatenate(idxs), full_idx)
        np.testing.assert_allclose(tot, full, rtol=1e-12, atol=1e-7)
        assert tot[16] == full[16]


class _HipBuf:
    """device buffer through the process's HIP runtime (ctypes), so the GPU suite needs no torch"""
    _hip = None

    @classmethod
    def hip(cls):
        if cls._hip is None:
            import ctypes
            cls._hip = ctypes.CDLL("libamdhip64.so.7")  # the instance the product library already uses (same SONAME)
            cls._hip.hipMalloc.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_size_t]
            cls._hip.hipMemcpy.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int]
            cls._hip.hipFree.argtypes = [ctypes.c_void_p]
        return cls._hip

    def __init__(self, host):
        import ctypes
        self.host = np.ascontiguousarray(host, dtype=np.float32)
        self.ptr = ctypes.c_void_p()
        assert self.hip().hipMalloc(ctypes.byref(self.ptr), self.host.nbytes) == 0
        assert self.hip().hipMemcpy(self.ptr, self.host.ctypes.data, self.host.nbytes, 1) == 0  # hipMemcpyHostToDevice

    def row(self, k):
        return self.ptr.value + k * self.host.shape[1] * 4

    def free(self):
        self.hip().hipFree(self.ptr)


def test_resident_device_pointers(pkg, O, synth, small_scene):
    """clouds already in HBM (caller-owned device pointers): mola_icp_set_*_device + align_resident"""
    import ctypes
    g, l, _ = synth.make_pair(8000, 6000, seed=31, scene=small_scene)
    icp = pkg.ICP(device=0)
    dg, dl = _HipBuf(g), _HipBuf(l)
    L = pkg._lib
    L.check(L.lib().mola_icp_set_map_device(icp._h, dg.row(0), dg.row(1), dg.row(2), g.shape[1]))
    L.check(L.lib().mola_icp_set_local_device(icp._h, dl.row(0), dl.row(1), dl.row(2), l.shape[1]))
    p = p2p_params(pkg, max_iterations=30)
    for kern in (pkg.NN_VALU, pkg.NN_MFMA, pkg.NN_TILED):
        p.nn_kernel = kern
        r = icp.align_resident(np.eye(4), p)
        ref = O.align(g, l, np.eye(4), O.params_from_product(p))
        rot, trans = O.pose_error(r.optimal_tf, ref["T"])
        assert r.nIterations == ref["n_iterations"] and rot < 1e-8 and trans < 1e-8
        assert r.nn_kernel_used == kern
    icp.set_stream(None)  # own stream again
    assert icp.align_resident(np.eye(4), p).nIterations == ref["n_iterations"]
    icp.close()
    dg.free()
    dl.free()


def test_loop_closure_montecarlo_multi_init(pkg, O, icp, synth, small_scene):
    """f2 (src/LidarOdometry.cpp:767-788): several perturbed guesses on one uploaded pair; the winner is the
    first attempt with the highest goodness; every attempt equals a stand-alone align bit for bit."""
    g, l, Tgt = synth.make_pair(12000, 10000, seed=61, scene=small_scene, noise_sigma=0.005)
    p = p2p_params(pkg, max_iterations=40, matcher_threshold=0.6)
    rng = np.random.default_rng(5)
    guesses = [np.eye(4)]
    for _ in range(5):
        d = rng.normal(0, 1, 4) * np.array([0.3, 0.3, 0.3, np.deg2rad(2.0)])
        guesses.append(synth.pose_from_xyzypr(d[0], d[1], d[2], d[3], 0, 0))
    res, best = icp.align_multi_init(g, l, guesses, p)
    singles = [icp.align(g, l, T0, p) for T0 in guesses]
    for r, s in zip(res, singles):
        assert r.nIterations == s.nIterations and r.terminationReason == s.terminationReason
        assert np.array_equal(r.optimal_tf, s.optimal_tf) and r.quality == s.quality
    q = [r.quality for r in res]
    assert best == int(np.argmax(q)) and q[best] == max(q)
    ref = O.align(g, l, guesses[best], O.params_from_product(p))
    rot, trans = O.pose_error(res[best].optimal_tf, ref["T"])
    assert rot < 1e-8 and trans < 1e-8
    # nothing inside the gate for any guess -> no winner
    far = np.ascontiguousarray(l + np.float32(500))
    res, best = icp.align_multi_init(g, far, guesses[:2], p)
    assert best == -1 and all(r.quality == 0 for r in res)


def test_cloud_cache_align_cached(pkg, O, synth, small_scene):
    """row f4: clouds kept prepared in HBM by id; align between cached ids == align from host buffers, in every role"""
    import threading
    icp = pkg.ICP(device=0)
    clouds = {}
    for k in range(4):
        g, l, _ = synth.make_pair(9000 + 500 * k, 9000 + 300 * k, seed=300 + k, scene=small_scene)
        clouds[2 * k], clouds[2 * k + 1] = g, l
        icp.cloud_put(2 * k, g)
        icp.cloud_put(2 * k + 1, l)
    n, nbytes = icp.cloud_count()
    assert n == 8 and nbytes > 8 * 9000 * 28
    p = p2p_params(pkg, max_iterations=30, matcher_threshold=0.6)
    for kern in (pkg.NN_AUTO, pkg.NN_MFMA, pkg.NN_VALU):
        p.nn_kernel = kern
        for a, b in ((0, 1), (3, 2), (4, 5), (1, 0)):      # every cloud serves as map and as local cloud
            r = icp.align_cached(a, b, np.eye(4), p)
            s = icp.align(clouds[a], clouds[b], np.eye(4), p)
            assert r.nIterations == s.nIterations and r.terminationReason == s.terminationReason
            assert np.array_equal(r.optimal_tf, s.optimal_tf) and r.quality == s.quality
            assert r.ms_upload == 0.0
    # the shipped point-to-plane pipeline on cached clouds
    pp = pkg.Parameters.load_from_file(os.path.join(os.path.dirname(os.path.dirname(__file__)), "params",
                                                    "icp-settings-regular.yaml"))
    r = icp.align_cached(0, 1, np.eye(4), pp)
    s = icp.align(clouds[0], clouds[1], np.eye(4), pp)
    assert r.nIterations == s.nIterations and np.array_equal(r.optimal_tf, s.optimal_tf)
    # concurrent aligns on shared cached clouds (the reference's pool threads read the same KF clouds)
    out = [None] * 6
    p.nn_kernel = pkg.NN_AUTO

    def run(i):
        out[i] = icp.align_cached(2 * (i % 4), 2 * (i % 4) + 1, np.eye(4), p)
    th = [threading.Thread(target=run, args=(i,)) for i in range(6)]
    [t.start() for t in th]
    [t.join() for t in th]
    for i, r in enumerate(out):
        ref = O.align(clouds[2 * (i % 4)], clouds[2 * (i % 4) + 1], np.eye(4), O.params_from_product(p))
        assert r.nIterations == ref["n_iterations"]
        rot, trans = O.pose_error(r.optimal_tf, ref["T"])
        assert rot < 1e-7 and trans < 1e-9
    # replace, drop, errors
    icp.cloud_put(1, clouds[3])
    assert np.array_equal(icp.align_cached(0, 1, np.eye(4), p).optimal_tf, icp.align(clouds[0], clouds[3], np.eye(4), p).optimal_tf)
    icp.cloud_drop(7)
    assert icp.cloud_count()[0] == 7
    for bad in (lambda: icp.cloud_drop(7), lambda: icp.align_cached(0, 7, np.eye(4), p), lambda: icp.align_cached(99, 1, np.eye(4), p)):
        with pytest.raises(pkg.IcpError) as e:
            bad()
        assert e.value.status == pkg._lib.E_BADARG
    # an empty cached cloud behaves like an empty host cloud
    icp.cloud_put(50, np.zeros((3, 0), np.float32))
    assert icp.align_cached(0, 50, np.eye(4), p).terminationReason == pkg.TERM_NO_PAIRINGS
    assert icp.align_cached(50, 1, np.eye(4), p).terminationReason == pkg.TERM_NO_PAIRINGS
    icp.close()


def test_voxel_downsample(pkg, icp, synth, small_scene):
    """row f4: centroid per occupied voxel vs a numpy restatement (same fp32 key arithmetic)"""
    g, _, _ = synth.make_pair(10, 60000, seed=17, scene=small_scene)
    for voxel in (0.25, 1.0, 7.5):
        out = icp.voxel_downsample(g, voxel)
        o = g.min(1)
        inv = np.float32(1.0) / np.float32(voxel)
        ijk = np.floor(((g - o[:, None]).astype(np.float32) * inv).astype(np.float32)).astype(np.int64)
        key = (ijk[0] << 42) | (ijk[1] << 21) | ijk[2]
        order = np.argsort(key, kind="stable")
        ks = key[order]
        heads = np.nonzero(np.r_[True, ks[1:] != ks[:-1]])[0]
        cnt = np.diff(np.r_[heads, len(ks)])
        ref = np.stack([np.add.reduceat(g[a][order].astype(np.float64), heads) / cnt for a in range(3)]).astype(np.float32)
        assert out.shape == ref.shape
        np.testing.assert_allclose(out, ref, rtol=0, atol=2e-6)
        assert out.shape[1] < g.shape[1]
    assert icp.voxel_downsample(np.zeros((3, 0), np.float32), 1.0).shape == (3, 0)
    one = icp.voxel_downsample(np.array([[1.0], [2.0], [3.0]], np.float32), 0.5)
    assert np.array_equal(one, np.array([[1.0], [2.0], [3.0]], np.float32))
    with pytest.raises(pkg.IcpError):
        icp.voxel_downsample(g, 0.0)
    with pytest.raises(pkg.IcpError):
        icp.voxel_downsample(g, 1e-9)   # more than 2^21 voxels per axis


def test_native_rccl_single_rank(pkg, icp, golden):
    """the RCCL transport of the query-sharded path with a 1-rank communicator: ncclAllReduce runs in place on the
    device accumulator block between k_reduce_partials and the read-back; results must equal the plain path."""
    import ctypes
    import importlib.util
    L = pkg._lib
    spec = importlib.util.find_spec("torch")
    if spec is not None and spec.origin:   # the RCCL that matches the HIP runtime this process loaded
        cand = os.path.join(os.path.dirname(spec.origin), "lib", "librccl.so")
        if os.path.exists(cand):
            L.check(L.lib().mola_icp_comm_set_library(cand.encode()))
    g, l = golden["A_map"], golden["A_local"]
    h = pkg.ICP(device=0)
    h.set_map(g)
    h.set_local(l)
    p = p2p_params(pkg, max_iterations=30)
    plain = h.align_resident(np.eye(4), p)
    with pytest.raises(pkg.IcpError, match="no communicator"):   # nothing to report before mola_icp_comm_init
        h.comm_nranks()
    ident = (ctypes.c_uint8 * 128)()
    L.check(L.lib().mola_icp_comm_unique_id(ident))
    L.check(L.lib().mola_icp_comm_init(h._h, ident, 1, 0))
    assert h.comm_nranks() == 1                # what RCCL itself reports (ncclCommCount)
    with pytest.raises(pkg.IcpError):      # a second communicator on the same handle is refused
        L.check(L.lib().mola_icp_comm_init(h._h, ident, 1, 0))
    r = h.align_resident(np.eye(4), p)
    assert r.nIterations == plain.nIterations and np.array_equal(r.optimal_tf, plain.optimal_tf)
    assert r.quality == plain.quality
    pp = pkg.Parameters.load_from_file(os.path.join(os.path.dirname(os.path.dirname(__file__)), "params",
                                                    "icp-settings-regular.yaml"))
    a = h.align_resident(np.eye(4), pp)    # the 92-double plane form goes through the same collective
    h.comm_destroy()
    b = h.align_resident(np.eye(4), pp)
    assert a.nIterations == b.nIterations and np.array_equal(a.optimal_tf, b.optimal_tf)
    # (under a native communicator the PairedRatio count goes through the collective every rank issues -- the lists' own
    # count, which stays on the rank, stands back; with one rank both give the same number, which is all one device can show)
    assert a.quality == b.quality and a.n_pairs == b.n_pairs
    h.close()


@pytest.mark.gpu
def test_align_cached_put_is_put_then_align(pkg, synth, small_scene):
    """mola_icp_align_cached_put (the odometry step: the new scan is prepared, aligned and cached in one call, no host wait between
    its prepare chain and the align's first launches) = cloud_put + align_cached, bit for bit, for both pipelines; the cloud is in
    the cache afterwards and serves as `from`; a cloud with a NaN is refused and NOT cached, an unknown `from` is an error"""
    icp = pkg.ICP(device=0)
    ref = pkg.ICP(device=0)
    shipped = pkg.Parameters.load_from_file(os.path.join(os.path.dirname(os.path.dirname(__file__)), "params", "icp-settings-regular.yaml"))
    p2p = p2p_params(pkg, max_iterations=30, matcher_threshold=0.6)
    clouds = [synth.make_pair(20_000 + 700 * k, 10, seed=400 + k, scene=small_scene)[1] for k in range(5)]
    icp.cloud_put(0, clouds[0])
    ref.cloud_put(0, clouds[0])
    for k in range(1, 5):
        p = shipped if k % 2 else p2p
        r = icp.align_cached_put(k - 1, k, clouds[k], np.eye(4), p)
        ref.cloud_put(k, clouds[k])
        s = ref.align_cached(k - 1, k, np.eye(4), p)
        assert r.nIterations == s.nIterations and r.terminationReason == s.terminationReason
        assert np.array_equal(r.optimal_tf, s.optimal_tf) and r.quality == s.quality and r.n_pairs == s.n_pairs
    assert icp.cloud_count()[0] == 5
    bad = clouds[2].copy()
    bad[1, 1234] = np.nan
    with pytest.raises(pkg.IcpError, match="non-finite"):
        icp.align_cached_put(4, 77, bad, np.eye(4), shipped)
    assert icp.cloud_count()[0] == 5
    with pytest.raises(pkg.IcpError):
        icp.align_cached(4, 77, np.eye(4), shipped)
    with pytest.raises(pkg.IcpError, match="no cached cloud"):
        icp.align_cached_put(1234, 78, clouds[1], np.eye(4), shipped)
    assert icp.cloud_count()[0] == 5
    # an empty scan: cached like cloud_put caches it, the align ends without pairings
    r = icp.align_cached_put(4, 80, np.zeros((3, 0), np.float32), np.eye(4), shipped)
    assert r.nIterations == 0 and r.quality == 0.0 and icp.cloud_count()[0] == 6
    # ... and the handle is fine afterwards
    r = icp.align_cached_put(4, 79, clouds[1], np.eye(4), shipped)
    assert r.nIterations >= 1 and icp.cloud_count()[0] == 7
    icp.close()
    ref.close()


@pytest.mark.gpu
def test_side_stream_prepare_changes_nothing(pkg, synth, small_scene, monkeypatch):
    """a call that brings both clouds prepares the queries on a second stream beside the map (HipWorkspace::prepare_both): the same
    result, bit for bit, as with both chains on one stream (MOLA_ICP_NO_SIDE_PREPARE=1) -- both pipelines, sizes on both sides of
    the tiled matcher's threshold, repeated calls on one handle (the side stream's scratch is reused)"""
    shipped = pkg.Parameters.load_from_file(os.path.join(os.path.dirname(os.path.dirname(__file__)), "params", "icp-settings-regular.yaml"))
    p2p = p2p_params(pkg, max_iterations=30, matcher_threshold=0.6)
    pairs = [synth.make_pair(n, m, seed=500 + k, scene=small_scene)[:2] for k, (n, m) in enumerate(((9000, 12000), (30000, 20000), (2000, 3000), (45000, 45000)))]
    res = {}
    try:
        for mode in ("side", "one stream"):
            if mode == "one stream":
                monkeypatch.setenv("MOLA_ICP_NO_SIDE_PREPARE", "1")
            else:
                monkeypatch.delenv("MOLA_ICP_NO_SIDE_PREPARE", raising=False)
            pkg._lib.lib().mola_icp_debug_reload_env()
            icp = pkg.ICP(device=0)
            res[mode] = [icp.align(g, l, np.eye(4), p) for (g, l) in pairs for p in (p2p, shipped)]
            icp.close()
    finally:
        monkeypatch.delenv("MOLA_ICP_NO_SIDE_PREPARE", raising=False)
        pkg._lib.lib().mola_icp_debug_reload_env()
    for a, b in zip(res["side"], res["one stream"]):
        assert np.array_equal(a.optimal_tf, b.optimal_tf) and a.nIterations == b.nIterations and a.quality == b.quality and a.n_pairs == b.n_pairs


@pytest.mark.gpu
def test_align_cached_put_from_four_threads_on_one_handle(pkg, synth, small_scene):
    """the reference calls align() from the odometry thread and from pool threads on one ICP object (src/LidarOdometry.cpp:94-96,
    869): four threads each drive their own chain of scans through mola_icp_align_cached_put on ONE handle (pool workspaces, the
    shared cloud cache, deferred builds in flight side by side) -- every result equals the serial run's, bit for bit"""
    import threading
    shipped = pkg.Parameters.load_from_file(os.path.join(os.path.dirname(os.path.dirname(__file__)), "params", "icp-settings-regular.yaml"))
    chains = [[synth.make_pair(15_000 + 900 * t + 300 * k, 10, seed=700 + 10 * t + k, scene=small_scene)[1] for k in range(5)] for t in range(4)]

    def run_chain(icp, t, out):
        base = 1000 * (t + 1)
        icp.cloud_put(base, chains[t][0])
        for k in range(1, 5):
            out.append(icp.align_cached_put(base + k - 1, base + k, chains[t][k], np.eye(4), shipped))
            icp.cloud_drop(base + k - 1)

    serial = pkg.ICP(device=0)
    want = [[] for _ in range(4)]
    for t in range(4):
        run_chain(serial, t, want[t])
    serial.close()
    icp = pkg.ICP(device=0)
    got, errs = [[] for _ in range(4)], []

    def worker(t):
        try:
            for _ in range(3):      # (three times over: the pool's workspaces change hands between the threads)
                got[t].clear()
                run_chain(icp, t, got[t])
                icp.cloud_drop(1000 * (t + 1) + 4)
        except Exception as e:  # noqa: BLE001
            errs.append((t, repr(e)))

    th = [threading.Thread(target=worker, args=(t,)) for t in range(4)]
    [x.start() for x in th]
    [x.join(300) for x in th]
    assert not errs, errs
    for t in range(4):
        assert len(got[t]) == 4
        for a, b in zip(got[t], want[t]):
            assert np.array_equal(a.optimal_tf, b.optimal_tf) and a.nIterations == b.nIterations and a.quality == b.quality
    assert icp.cloud_count()[0] == 0
    icp.close()
