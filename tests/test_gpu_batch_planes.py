"""-m gpu: the reference's own nearby-keyframe / loop-closure settings (params/icp-settings-loop-closure.yaml:23-39 =
Matcher_Point2Plane + Solver_GaussNewton; src/LidarOdometry.cpp:704-741, 767-788) through the LOCKSTEP batched path:
K guesses / K pairs as blockIdx.y of k_knn_coop, the plane form of every problem in one launch, K host Gauss-Newton solves
per step.  Every batched result must be bit-equal to its stand-alone align; the winner also equals the CPU oracle."""
import os
import time

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _same(r, s):
    assert r.nIterations == s.nIterations and r.terminationReason == s.terminationReason, (r.nIterations, s.nIterations)
    assert np.array_equal(r.optimal_tf, s.optimal_tf) and r.quality == s.quality
    assert r.n_pairs == s.n_pairs and r.rmse == s.rmse and np.array_equal(r.optimal_tf_cov, s.optimal_tf_cov)


def test_loop_closure_montecarlo_shipped_settings_is_a_device_batch(pkg, O, synth):
    g, l, Tgt = synth.make_pair(100_000, 100_000, seed=42)
    p = pkg.Parameters.load_from_file(os.path.join(ROOT, "params", "icp-settings-loop-closure.yaml"))
    assert p.matcher_class == pkg._lib.MATCHER_POINT2PLANE
    rng = np.random.default_rng(7)
    guesses = []
    for _ in range(10):   # the sampler's spread (kitti-default.yaml:35-39 scale)
        d = rng.normal(0, 1, 4) * np.array([0.3, 0.3, 0.3, np.deg2rad(2.0)])
        guesses.append(synth.pose_from_xyzypr(d[0], d[1], d[2], d[3], 0, 0))
    icp = pkg.ICP(device=0)
    icp.align_multi_init(g, l, guesses, p)                       # warm-up (all ten: the per-guess plane buffers are allocated once)
    dt = 1e9
    for _ in range(3):   # (the best of three: a timing comparison must not fail on one hiccup of the box)
        t0 = time.perf_counter()
        res, best = icp.align_multi_init(g, l, guesses, p)
        dt = min(dt, time.perf_counter() - t0)
    t0 = time.perf_counter()
    singles = [icp.align(g, l, T0, p) for T0 in guesses]
    dt1 = time.perf_counter() - t0
    print(f"\n[f2, shipped settings] 10 guesses on 100k x 100k: batched {dt*1e3:.2f} ms, stand-alone aligns one after another "
          f"{dt1*1e3:.2f} ms (x{dt1/dt:.1f}); iterations {[r.nIterations for r in res]}")
    for r, s in zip(res, singles):
        _same(r, s)
    assert len({r.nIterations for r in res}) > 1                  # problems leave the lockstep launches one by one
    q = [r.quality for r in res]
    assert best == int(np.argmax(q))
    assert dt < dt1
    op = O.params_from_product(p)
    ref = O.align_p2pl(g, l, guesses[best], op, p.plane_eigen_threshold, int(p.knn), int(p.solver_max_iterations))
    rot, trans = O.pose_error(res[best].optimal_tf, ref["T"])
    assert res[best].nIterations == ref["n_iterations"] and rot < 1e-7 and trans < 1e-7, (rot, trans)
    assert res[best].quality == pytest.approx(ref["quality"], abs=1e-12)
    icp.close()


def test_nearby_keyframe_pairs_shipped_settings_batch(pkg, O, synth):
    """K different pairs of different sizes -- one beyond the cooperative kernel's stand-alone range (its stand-alone align
    runs the persistent k_knn_planes; the lists are exact either way), one tiny, one without any overlap -- each equal to
    its stand-alone align, bit for bit; more pairs than one lockstep chunk holds."""
    p = pkg.Parameters.load_from_file(os.path.join(ROOT, "params", "icp-settings-loop-closure.yaml"))
    sizes = [(40_000, 50_000), (100_000, 100_000), (150_000, 160_000), (700, 900), (30_011, 64_000), (64_000, 30_011)] + [(20_000 + 1000 * k, 25_000) for k in range(9)]
    made = [synth.make_pair(n, m, seed=300 + i) for i, (n, m) in enumerate(sizes)]
    pairs = [mm[:2] for mm in made]
    far = pairs[4][1].copy()
    far[0] += 500.0                                               # a pair that does not overlap: NoPairings at iteration 0
    pairs[4] = (pairs[4][0], far)
    inits = [np.eye(4) if k % 3 else made[k][2] for k in range(len(pairs))]   # every third starts at its ground truth
    icp = pkg.ICP(device=0)
    res = icp.align_batch(pairs, inits, p)
    assert len(res) == len(pairs)
    for k, r in enumerate(res):
        s = icp.align(pairs[k][0], pairs[k][1], inits[k], p)
        _same(r, s)
    assert res[4].terminationReason == pkg.TERM_NO_PAIRINGS and res[4].nIterations == 0
    assert len({r.nIterations for r in res}) > 2
    # two sampled pairs against the oracle
    op = O.params_from_product(p)
    for k in (0, 3):
        ref = O.align_p2pl(pairs[k][0], pairs[k][1], inits[k], op, p.plane_eigen_threshold, int(p.knn), int(p.solver_max_iterations))
        rot, trans = O.pose_error(res[k].optimal_tf, ref["T"])
        assert res[k].nIterations == ref["n_iterations"] and rot < 1e-6 and trans < 1e-6, (k, rot, trans)
    # the same through the knn 3 ... 8 instantiations (two pairs each)
    for knn in (3, 8):
        q = p.copy()
        q.knn = knn
        rb = icp.align_batch(pairs[:2], inits[:2], q)
        for k in range(2):
            _same(rb[k], icp.align(pairs[k][0], pairs[k][1], inits[k], q))
    icp.close()


def test_concurrent_batch_calls_on_one_handle_finish(pkg, synth):
    """mola_icp_align_batch used to run its second lockstep loop as a job on the handle's fixed-size worker pool, where
    it blocked on prepare-ahead tasks queued to the same pool -- several concurrent calls with 4+ chunks each could fill every
    worker with a waiting loop.  Six threads, 40 pairs each (4 chunks), one handle: all must return, each result equal to the
    single-threaded batch's."""
    import threading
    pairs = [synth.make_pair(9_000 + 300 * k, 10_000, seed=500 + k)[:2] for k in range(40)]
    p = pkg.Parameters()
    p.matcher_threshold, p.max_iterations, p.min_abs_step_trans, p.min_abs_step_rot = 1.0, 30, 5e-5, 1e-5
    icp = pkg.ICP(device=0)
    ref = icp.align_batch(pairs, [np.eye(4)] * len(pairs), p)
    out, errs = [None] * 6, []

    def work(i):
        try:
            out[i] = icp.align_batch(pairs, [np.eye(4)] * len(pairs), p)
        except Exception as e:  # noqa: BLE001
            errs.append(e)

    th = [threading.Thread(target=work, args=(i,)) for i in range(6)]
    for t in th:
        t.start()
    for t in th:
        t.join(timeout=300)
    assert not any(t.is_alive() for t in th), "a concurrent align_batch call never returned"
    assert not errs, errs
    for res in out:
        for r, s in zip(res, ref):
            assert r.nIterations == s.nIterations and np.array_equal(r.optimal_tf, s.optimal_tf)
    icp.close()
