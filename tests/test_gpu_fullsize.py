"""-m gpu: BASELINE.json's full sizes, checked through size-independent properties (the oracle
cannot finish brute force there; its kd-tree still checks a sample of queries exactly)."""
import numpy as np
import pytest

from tests.helpers import p2p_params

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def big(synth):
    return synth.make_pair(1_000_000, 1_000_000, seed=42)


@pytest.mark.parametrize("kern", [1, 2, 3])
def test_1m_x_1m_properties(pkg, O, synth, big, kern):
    g, l, Tgt = big
    icp = pkg.ICP(device=0)
    icp.set_map(g)
    # (1) map against itself at identity: every point is its own nearest neighbour at d2 == 0
    #     (lowest-index rule also covers exact duplicate points)
    icp.set_local(g)
    idx, d2, n = icp.match(np.eye(4), 0.5, g.shape[1], kern)
    assert n == g.shape[1] and (d2 == 0).all()
    same = idx == np.arange(g.shape[1])
    if not same.all():  # duplicates: the reported index must hold identical coordinates and be lower
        bad = np.nonzero(~same)[0]
        assert (idx[bad] < bad).all() and np.array_equal(g[:, idx[bad]], g[:, bad])
    # (2) the real pair: idempotent, and a 20k-query sample equals the oracle's exact kd-tree bit for bit
    icp.set_local(l)
    idx, d2, n = icp.match(np.eye(4), 1.0, l.shape[1], kern)
    idx2, d22, n2 = icp.match(np.eye(4), 1.0, l.shape[1], kern)
    assert n == n2 and np.array_equal(idx, idx2) and np.array_equal(d2, d22)
    sel = np.arange(0, l.shape[1], 50)
    oidx, od2, _ = O.match(g, np.ascontiguousarray(l[:, sel]), np.eye(4), 1.0, O.KdTree(g))
    assert np.array_equal(idx[sel], oidx)
    k = oidx >= 0
    assert np.array_equal(d2[sel][k], od2[k])
    # (3) reported d2 is the distance to the reported index, and < gate^2; count matches
    q = O.transform(np.eye(4), l)
    kk = idx >= 0
    dd = q[:, kk].astype(np.float64) - g[:, idx[kk]].astype(np.float64)
    np.testing.assert_allclose((dd * dd).sum(0), d2[kk], rtol=1e-5, atol=1e-9)
    assert (d2[kk] < np.float32(1.0)).all() and int(kk.sum()) == n
    # (4) linearity of the accumulators: the sums over two halves of the scan (each its own set_local + match +
    #     accumulate) add up to the whole scan's -- same pairings, fp64 sums in another order: 1e-12
    p = p2p_params(pkg)
    whole = icp.accumulate(p, np.eye(4))
    assert whole[16] == n
    half = l.shape[1] // 2 + 12345          # (an uneven cut: neither half is a whole number of items)
    parts = np.zeros(24)
    for lo_, hi_ in ((0, half), (half, l.shape[1])):
        icp.set_local(np.ascontiguousarray(l[:, lo_:hi_]))
        i_h, d_h, n_h = icp.match(np.eye(4), 1.0, hi_ - lo_, kern)
        assert np.array_equal(i_h, idx[lo_:hi_]) and np.array_equal(d_h, d2[lo_:hi_])
        parts += icp.accumulate(p, np.eye(4))
    np.testing.assert_allclose(parts, whole, rtol=1e-12, atol=1e-7)
    assert parts[16] == whole[16]
    icp.close()


def test_1m_x_1m_config3_run(pkg, O, big):
    """config 3: 1M scan vs 1M map, 40 fixed iterations.  Point-to-point ICP on dense planar
    clouds converges slowly (the same in the oracle), so the checks are: every iteration ran, the
    pose moved monotonically closer to the seeded T_gt, the first 3 iterations equal the
    oracle's (exact kd-tree, fp64 sums) to ~1e-9 -- and so does the pose after all 40."""
    g, l, Tgt = big
    icp = pkg.ICP(device=0)
    r = icp.align(g, l, np.eye(4), p2p_params(pkg, max_iterations=40, fixed_iterations=1))
    assert r.nIterations == 40 and r.terminationReason == pkg.TERM_MAX_ITERATIONS
    assert r.n_nn_launches == 41
    rot40, trans40 = O.pose_error(r.optimal_tf, Tgt)
    r10 = icp.align(g, l, np.eye(4), p2p_params(pkg, max_iterations=10, fixed_iterations=1))
    rot10, trans10 = O.pose_error(r10.optimal_tf, Tgt)
    rot0, trans0 = O.pose_error(np.eye(4), Tgt)
    assert trans40 < trans10 < trans0 and rot40 < rot10 < rot0
    assert 0 < r.quality <= 1
    p3 = p2p_params(pkg, max_iterations=3, fixed_iterations=1)
    r3 = icp.align(g, l, np.eye(4), p3)
    ref = O.align(g, l, np.eye(4), O.params_from_product(p3))
    rot, trans = O.pose_error(r3.optimal_tf, ref["T"])
    assert rot <= 1e-4 and trans <= 1e-3 and rot < 1e-8 and trans < 1e-8, (rot, trans)
    assert r3.n_pairs == ref["n_pairs"] and r3.quality == pytest.approx(ref["quality"], abs=1e-12)
    # ... and the WHOLE workload: the pose after all 40 iterations against the oracle's 40 (BASELINE.json's tolerance
    # 1e-4 rad / 1e-3 m; the bit-identical pairings leave fp64 summation order only: ~1e-9)
    p40 = p2p_params(pkg, max_iterations=40, fixed_iterations=1)
    ref40 = O.align(g, l, np.eye(4), O.params_from_product(p40))
    rot, trans = O.pose_error(r.optimal_tf, ref40["T"])
    assert ref40["n_iterations"] == 40 == r.nIterations and ref40["termination"] == r.terminationReason
    assert rot <= 1e-4 and trans <= 1e-3 and rot < 1e-7 and trans < 1e-7, (rot, trans)
    assert r.n_pairs == ref40["n_pairs"] and r.quality == pytest.approx(ref40["quality"], abs=1e-12)
    icp.close()


def test_500k_queries_use_64_query_items(pkg, O, synth):
    """~0.4-0.8M queries: the matcher runs with 64-query items (several per persistent wave).  Idempotence,
    seeded == unseeded, and a sample against the oracle's kd-tree."""
    g, l, _ = synth.make_pair(500_000, 700_000, seed=3)
    icp = pkg.ICP(device=0)
    icp.set_map(g)
    icp.set_local(l)
    T = synth.pose_from_xyzypr(0.05, 0.02, -0.01, 0.003, 0.0, 0.001)
    idx0, d20, n0 = icp.match(T, 1.0, l.shape[1], pkg.NN_TILED)        # unseeded
    icp.match(np.eye(4), 1.0, l.shape[1], pkg.NN_TILED)
    idx1, d21, n1 = icp.match(T, 1.0, l.shape[1], pkg.NN_TILED)        # seeded from another pose
    assert n0 == n1 and np.array_equal(idx0, idx1) and np.array_equal(d20, d21)
    sel = np.arange(0, l.shape[1], 100)
    oidx, od2, _ = O.match(g, np.ascontiguousarray(l[:, sel]), T, 1.0, O.KdTree(g))
    assert np.array_equal(idx1[sel], oidx)
    k = oidx >= 0
    assert np.array_equal(d21[sel][k], od2[k])
    icp.close()


def test_1m_x_1m_shipped_point2plane(pkg, O, big):
    """The shipped Point2Plane + Gauss-Newton pipeline at config-3 size (waves take several items each, the warm
    start is active from the second iteration): the warm start must not change a single bit, and a sample of
    the plane pairings equals the oracle's exact kd-tree kNN."""
    import os
    g, l, Tgt = big
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    p = pkg.Parameters.load_from_file(os.path.join(root, "params", "icp-settings-regular.yaml"))
    p.fixed_iterations, p.skip_quality, p.max_iterations = 1, 1, 4
    icp = pkg.ICP(device=0)
    icp.set_map(g)
    icp.set_local(l)
    r = icp.align_resident(np.eye(4), p)
    os.environ["MOLA_ICP_NO_KNN_SEED"] = "1"
    pkg._lib.lib().mola_icp_debug_reload_env()
    try:
        r0 = icp.align_resident(np.eye(4), p)
        v0, c0, n0, k0, cnt0 = icp.match_planes(r.optimal_tf, p, l.shape[1])
    finally:
        del os.environ["MOLA_ICP_NO_KNN_SEED"]
        pkg._lib.lib().mola_icp_debug_reload_env()
    assert r.nIterations == r0.nIterations == 4 and np.array_equal(r.optimal_tf, r0.optimal_tf)
    rot, trans = O.pose_error(r.optimal_tf, Tgt)
    rot_i, trans_i = O.pose_error(np.eye(4), Tgt)
    assert rot < 0.1 * rot_i and trans < 0.1 * trans_i        # point-to-plane converges fast on this scene
    # seeded (neighbours of the previous launch at another pose) == unseeded, bit for bit
    icp.match_planes(np.eye(4), p, l.shape[1])
    v1, c1, n1, k1, cnt1 = icp.match_planes(r.optimal_tf, p, l.shape[1])
    assert cnt1 == cnt0 and np.array_equal(k1, k0) and np.array_equal(v1, v0)
    assert np.array_equal(c1, c0) and np.array_equal(n1, n0)
    # a 4k-query sample against the oracle
    sel = np.arange(0, l.shape[1], 250)
    ov, oc, on, ok, _ = O.match_point2plane(g, np.ascontiguousarray(l[:, sel]), r.optimal_tf, p.matcher_threshold,
                                            p.plane_eigen_threshold, p.knn, O.KdTree(g))
    assert np.array_equal(k1[sel], ok) and np.array_equal(v1[sel], ov)
    kk = ov.astype(bool)
    np.testing.assert_allclose(c1[sel][kk], oc[kk], atol=1e-12)
    icp.close()


def test_config1_kitti_like_pair_through_front_end(pkg, O, synth):
    """BASELINE configs[0]: params/kitti-default.yaml + one KITTI-like 64-ring scan pair (~120k points each; no
    real KITTI data exists in the image) through the front-end, GPU ICP vs the CPU oracle on the same pair."""
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    lp = pkg.LidarOdometryParams.load_from_file(os.path.join(root, "params", "kitti-default.yaml"), root)
    s0 = synth.lidar_scan(synth.pose_from_xyzypr(-10.0, 0.2, 0, 0.00, 0, 0), seed=11)
    s1 = synth.lidar_scan(synth.pose_from_xyzypr(-9.0, 0.25, 0, 0.01, 0, 0), seed=12)   # 1 m forward, 0.1 s later
    assert 100_000 <= s0.shape[1] <= 130_000
    icp = pkg.ICP(device=0)
    lo = pkg.LidarOdometry(lp, icp=icp)
    a = lo.on_new_observation(10.0, s0)
    b = lo.on_new_observation(10.1, s1)
    assert a.status == pkg._lib.LO_FIRST_SCAN and a.keyframe_created
    assert b.status == pkg._lib.LO_ICP_RAN and not b.used_with_vel_params
    # no twist yet: the LidarOdometry ICP object with the NearbyAlign case's mp2p_icp::Parameters (cpp:287-290, 869)
    p = pkg.Parameters.compose(lp.icp_case("with_vel"), lp.icp_case("without_vel"))
    assert p.matcher_class == pkg._lib.MATCHER_POINT2PLANE     # the reference's shipped pipeline (kitti-default.yaml:43,46)
    ref = O.align_p2pl(s0, s1, np.eye(4), O.params_from_product(p), p.plane_eigen_threshold, p.knn, p.solver_max_iterations)
    assert b.icp.nIterations == ref["n_iterations"] and b.icp.terminationReason == ref["termination"]
    rot, trans = O.pose_error(b.rel_pose, ref["T"])
    assert rot <= 1e-4 and trans <= 1e-3 and rot < 1e-7 and trans < 1e-7, (rot, trans)
    assert b.icp.quality == pytest.approx(ref["quality"], abs=1e-12)
    assert b.icp.nn_kernel_used == pkg.NN_TILED          # auto picks the tiled matcher at this size
    lo.close()
    icp.close()


def test_config5_10m_map_sequential_shards(pkg, O, synth):
    """BASELINE configs[4]: 10M-point map vs 1M queries, 8 query shards of 125k run one after another on one
    GPU: the summed shard accumulators equal the un-sharded ones (the reduction RCCL performs on 8 GPUs), and a
    sample of the pairing equals the oracle's exact kd-tree bit for bit."""
    import importlib
    sharded = importlib.import_module("mola-fe-lidar_amd.sharded")
    g, l, _ = synth.make_pair(1_000_000, 10_000_000, seed=7)
    p = p2p_params(pkg)
    icp = pkg.ICP(device=0)
    icp.set_map(g)
    icp.set_local(l)
    idx, d2, n = icp.match(np.eye(4), 1.0, l.shape[1])
    full = icp.accumulate(p, np.eye(4))
    assert full[16] == n
    sel = np.arange(0, l.shape[1], 200)
    oidx, od2, _ = O.match(g, np.ascontiguousarray(l[:, sel]), np.eye(4), 1.0, O.KdTree(g))
    assert np.array_equal(idx[sel], oidx)
    k = oidx >= 0
    assert np.array_equal(d2[sel][k], od2[k])
    tot = np.zeros(24)
    for r in range(8):
        lo_, hi_ = sharded.shard_bounds(l.shape[1], r, 8)
        icp.set_local(np.ascontiguousarray(l[:, lo_:hi_]))
        i_r, _, _ = icp.match(np.eye(4), 1.0, hi_ - lo_)
        assert np.array_equal(i_r, idx[lo_:hi_])
        tot += icp.accumulate(p, np.eye(4))
    np.testing.assert_allclose(tot, full, rtol=1e-12, atol=1e-6)
    icp.close()


@pytest.mark.parametrize("quads", ["0", "1"])
def test_dense_map_both_sweeps_of_the_tiled_matcher(pkg, O, synth, quads, monkeypatch):
    """300k queries against a 3M-point map -- ten map points per query, the upper box levels read from global memory --, a launch WITHOUT
    seeds under the whole gate (the quad sweep then fills its per-quad tile lists to the brim and drains them in the middle of a list) and
    two seeded ones, with k_nn_tiled's quad sweep forced for every launch and with the pass-by-pass sweep: a sample of every pairing
    against the oracle's exact kd-tree, bit for bit, and the pair counts against each other."""
    monkeypatch.setenv("MOLA_ICP_QUADS", quads)
    monkeypatch.setenv("MOLA_ICP_COOP", "0")
    pkg._lib.lib().mola_icp_debug_reload_env()
    g, l, _ = synth.make_pair(300_000, 3_000_000, seed=11)
    kd = O.KdTree(g)
    sel = np.arange(3, l.shape[1], 50)
    ls = np.ascontiguousarray(l[:, sel])
    icp = pkg.ICP(device=0)
    icp.set_map(g)
    icp.set_local(l)
    for T in (np.eye(4), synth.pose_from_xyzypr(0.05, -0.02, 0.01, 0.004, 0.001, -0.001), synth.pose_from_xyzypr(0.3, 0.1, 0.0, 0.02, 0.0, 0.0)):
        idx, d2, n = icp.match(T, 1.0, l.shape[1], pkg.NN_TILED)
        oidx, od2, _ = O.match(g, ls, T, 1.0, kd)
        assert np.array_equal(idx[sel], oidx)
        k = oidx >= 0
        assert np.array_equal(d2[sel][k], od2[k])
        assert n == int((idx >= 0).sum())
    icp.close()
    monkeypatch.delenv("MOLA_ICP_QUADS")
    monkeypatch.delenv("MOLA_ICP_COOP")
    pkg._lib.lib().mola_icp_debug_reload_env()


def test_config5_sharded_align_10_iterations(pkg, O, synth):
    """BASELINE configs[4] as an ALIGN: 10M-point map, 1M queries, 8 query shards (the device's own Hilbert cut, as the
    8 ranks would hold them) run one after another on this one GPU inside every iteration -- each shard's matcher at
    the current pose, its accumulators summed over the shards (the sum RCCL performs over xGMI), fed back through the
    product's own loop (mola_icp_run_loop: Horn, stall test, quality).  The pose after 10 iterations equals the
    un-sharded align's to fp64 summation order."""
    g, l, _ = synth.make_pair(1_000_000, 10_000_000, seed=7)
    W = 8
    p = p2p_params(pkg, max_iterations=10, fixed_iterations=1)
    whole = pkg.ICP(device=0)
    whole.set_map(g)
    whole.set_local(l)
    ref = whole.align_resident(np.eye(4), p)
    whole.close()
    assert ref.nIterations == 10
    ranks = []
    seen = np.zeros(l.shape[1], dtype=np.int32)
    for r in range(W):                       # every "rank": the whole map resident, its shard of the scan
        icp = pkg.ICP(device=0)
        icp.set_map(g)
        n_r = icp.set_local_shard(l, r, W)
        seen[icp.local_shard_indices()] += 1
        ranks.append((icp, n_r))
    assert np.all(seen == 1) and sum(n for _, n in ranks) == l.shape[1]
    state = {}

    def match_fn(T, thr):
        state["T"], state["thr"] = T, thr
        tot = 0
        for icp, n_r in ranks:
            tot += icp.match(T, thr, n_r, copy=False)[2]
        return tot

    def accumulate_fn(pp, T, stage, cl, cg, reset):
        acc = np.zeros(24)
        for icp, _ in ranks:                 # the all-reduce, as a loop over the shards
            acc += icp.accumulate(pp, T, stage, cl, cg, reset)
        return acc

    res = pkg.run_loop(match_fn, accumulate_fn, np.eye(4), p, l.shape[1], g.shape[1])
    assert res.nIterations == 10 and res.terminationReason == ref.terminationReason
    rot, trans = O.pose_error(res.optimal_tf, ref.optimal_tf)
    assert rot < 1e-10 and trans < 1e-10, (rot, trans)
    assert res.n_pairs == ref.n_pairs and res.quality == pytest.approx(ref.quality, abs=1e-12)
    for icp, _ in ranks:
        icp.close()


def test_config4_64_pairs_of_100k_batch_and_device_pool(pkg, O, synth):
    """BASELINE configs[3] at its workload: 64 independent 100k x 100k scan pairs (seeds 100..163), <= 100 iterations with
    the stall test.  mola_icp_align_batch advances them a dozen at a time (one launch per stage over the pairs still
    iterating); the device pool deals them round-robin to one handle per device (two handles on this one GPU here,
    eight GPUs on a node).  Every result is bit-equal to the pair's stand-alone align; four sampled pairs also equal
    the CPU oracle."""
    import time
    made = [synth.make_pair(100_000, 100_000, seed=100 + s) for s in range(64)]
    pairs = [m[:2] for m in made]
    p = p2p_params(pkg, max_iterations=100)
    icp = pkg.ICP(device=0)
    icp.align(pairs[0][0], pairs[0][1], np.eye(4), p)           # warm-up (first launches, pools)
    t0 = time.perf_counter()
    res = icp.align_batch(pairs, [np.eye(4)] * 64, p)
    dt = time.perf_counter() - t0
    t0 = time.perf_counter()
    singles = [icp.align(g, l, np.eye(4), p) for g, l in pairs]
    dt1 = time.perf_counter() - t0
    print(f"\n[config 4] align_batch: 64 pairs in {dt*1e3:.1f} ms = {64/dt:.0f} pairs/s "
          f"({sum(r.nIterations for r in res)/dt:.0f} iterations/s); one after another: {64/dt1:.0f} pairs/s")
    for r, s in zip(res, singles):
        assert r.nIterations == s.nIterations and r.terminationReason == s.terminationReason
        assert np.array_equal(r.optimal_tf, s.optimal_tf) and r.quality == s.quality
        assert r.n_pairs == s.n_pairs and r.rmse == s.rmse and np.array_equal(r.optimal_tf_cov, s.optimal_tf_cov)
        assert r.nn_kernel_used == pkg.NN_TILED
    # pairs that stop at different iterations drop out of the lockstep launches one by one: every other pair starts at
    # its ground-truth pose (stalls within a few iterations), the rest at the identity
    inits = [made[k][2] if k % 2 else np.eye(4) for k in range(12)]
    mixed = icp.align_batch(pairs[:12], inits, p)
    assert len({r.nIterations for r in mixed}) > 1
    for k, r in enumerate(mixed):
        s1 = icp.align(pairs[k][0], pairs[k][1], inits[k], p)
        assert r.nIterations == s1.nIterations and r.terminationReason == s1.terminationReason
        assert np.array_equal(r.optimal_tf, s1.optimal_tf) and r.quality == s1.quality
    for k in (0, 21, 42, 63):
        ref = O.align(pairs[k][0], pairs[k][1], np.eye(4), O.params_from_product(p))
        assert res[k].nIterations == ref["n_iterations"] and res[k].terminationReason == ref["termination"]
        rot, trans = O.pose_error(res[k].optimal_tf, ref["T"])
        assert rot <= 1e-4 and trans <= 1e-3 and rot < 1e-7 and trans < 1e-8, (k, rot, trans)
        assert res[k].quality == pytest.approx(ref["quality"], abs=1e-12)
    # the device pool: one handle per visible device ...
    pool = pkg.DevicePool()
    assert len(pool) >= 1
    rp = pool.align_batch(pairs[:16], [np.eye(4)] * 16, p)
    for r, s in zip(rp, singles[:16]):
        assert r.nIterations == s.nIterations and np.array_equal(r.optimal_tf, s.optimal_tf) and r.quality == s.quality
    pool.close()
    # ... and the multi-handle dispatch exercised on this one GPU: two device slots pulling chunks of the call's pairs
    pool2 = pkg.DevicePool([0, 0])
    assert len(pool2) == 2
    rp = pool2.align_batch(pairs[:25], [np.eye(4)] * 25, p)
    for r, s in zip(rp, singles[:25]):
        assert r.nIterations == s.nIterations and np.array_equal(r.optimal_tf, s.optimal_tf) and r.quality == s.quality
    assert sum(pool2.last_shares()) == 25 and all(n > 0 for n in pool2.last_shares())   # both slots pulled chunks from the shared cursor
    assert pool2.align_batch([], [], p) == []
    # a MIXED batch (pairs that stop after a few iterations next to pairs that run the full hundred): the slots pull chunks until the
    # cursor runs dry -- every result still its stand-alone align's, and no slot is left without work
    inits25 = [made[k][2] if k % 3 == 0 else np.eye(4) for k in range(25)]
    rp = pool2.align_batch(pairs[:25], inits25, p)
    assert len({r.nIterations for r in rp}) > 1 and sum(pool2.last_shares()) == 25
    for k in (0, 1, 3, 12, 24):
        s1 = icp.align(pairs[k][0], pairs[k][1], inits25[k], p)
        assert rp[k].nIterations == s1.nIterations and np.array_equal(rp[k].optimal_tf, s1.optimal_tf) and rp[k].quality == s1.quality
    # the pool is reusable from several host threads (calls are serialised per pool)
    import threading
    outs = [None, None]

    def call(i):
        outs[i] = pool2.align_batch(pairs[:6], [np.eye(4)] * 6, p)
    th = [threading.Thread(target=call, args=(i,)) for i in range(2)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    for o in outs:
        for r, s in zip(o, singles[:6]):
            assert np.array_equal(r.optimal_tf, s.optimal_tf)
    pool2.close()
    icp.close()


def test_loop_closure_montecarlo_10_guesses_at_100k(pkg, O, synth):
    """row f2 (src/LidarOdometry.cpp:767-788): 10 perturbed guesses on ONE 100k x 100k pair as a batch dimension on the
    device -- each attempt bit-equal to a stand-alone align from that guess, first-best selection, and the wall time
    of the batch against the stand-alone aligns one after another."""
    import time
    g, l, Tgt = synth.make_pair(100_000, 100_000, seed=42)
    p = p2p_params(pkg, max_iterations=100)
    rng = np.random.default_rng(7)
    guesses = []
    for _ in range(10):
        d = rng.normal(0, 1, 4) * np.array([0.3, 0.3, 0.3, np.deg2rad(2.0)])
        guesses.append(synth.pose_from_xyzypr(d[0], d[1], d[2], d[3], 0, 0))
    icp = pkg.ICP(device=0)
    icp.align_multi_init(g, l, guesses, p)                       # warm-up (all ten: per-guess buffers are allocated once)
    dt = 1e9
    for _ in range(3):   # (the best of three: a timing comparison must not fail on one hiccup of the box)
        t0 = time.perf_counter()
        res, best = icp.align_multi_init(g, l, guesses, p)
        dt = min(dt, time.perf_counter() - t0)
    t0 = time.perf_counter()
    singles = [icp.align(g, l, T0, p) for T0 in guesses]
    dt1 = time.perf_counter() - t0
    print(f"\n[f2] 10 guesses on 100k x 100k: batched {dt*1e3:.1f} ms, stand-alone aligns one after another {dt1*1e3:.1f} ms "
          f"(x{dt1/dt:.1f}); iterations {[r.nIterations for r in res]}")
    for r, s in zip(res, singles):
        assert r.nIterations == s.nIterations and r.terminationReason == s.terminationReason
        assert np.array_equal(r.optimal_tf, s.optimal_tf) and r.quality == s.quality and r.n_pairs == s.n_pairs
    q = [r.quality for r in res]
    assert best == int(np.argmax(q))                              # first best (strictly greater wins, cpp:785)
    assert dt < dt1                                               # (the measured ratio is printed above and in DESIGN.md)
    ref = O.align(g, l, guesses[best], O.params_from_product(p))
    rot, trans = O.pose_error(res[best].optimal_tf, ref["T"])
    assert res[best].nIterations == ref["n_iterations"] and rot < 1e-7 and trans < 1e-8
    icp.close()


def test_cooperative_matcher_on_a_map_whose_box_levels_exceed_lds(pkg, O, synth):
    """a small scan against a 4M-point map: one workgroup per item (k_nn_coop), the upper box levels read from global memory
    (they do not fit the 40 KB LDS budget above ~3.4M points); unseeded and seeded launches, fused row sums"""
    g, l, _ = synth.make_pair(20_011, 4_000_000, seed=77)
    icp = pkg.ICP(device=0)
    icp.set_map(g)
    icp.set_local(l)
    T = synth.pose_from_xyzypr(0.2, -0.1, 0.03, 0.01, 0.0, 0.002)
    tree = O.KdTree(g)
    p = p2p_params(pkg)
    for Tk in (np.eye(4), T, T):        # unseeded, seeded after a pose step, seeded at the same pose
        idx, d2, n = icp.match(Tk, 1.0, l.shape[1], pkg.NN_TILED)
        oidx, od2, on = O.match(g, l, Tk, 1.0, tree)
        assert n == on and np.array_equal(idx, oidx)
        assert np.array_equal(d2[oidx >= 0], od2[oidx >= 0])
        acc = icp.accumulate(p, Tk)
        oacc = O.accumulate(g, l, oidx, od2, O.params_from_product(p), Tk)
        np.testing.assert_allclose(acc, oacc, rtol=1e-10, atol=1e-6)
    icp.close()
