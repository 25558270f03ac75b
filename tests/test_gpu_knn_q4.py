"""-m gpu: k_knn_q4 (the plane matcher with four, two or one lane(s) per query, csrc/kernels_knn_q4.hpp): every instantiated list length against the
oracle at a first launch and at seeded / certified launches behind it, ragged tails, a map too large for the LDS copy of the box
levels, exact distance ties, the switch (MOLA_ICP_KNN_Q4=0|1: the same lists, planes and aligns from the kernels it replaces, also when
the kernel changes between the launches of one align), and the lockstep batches."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
REGULAR = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "params", "icp-settings-regular.yaml")


def _reload(pkg, **env):
    for k, v in env.items():
        if v is None:
            os.environ.pop(k, None)
        else:
            os.environ[k] = v
    pkg._lib.lib().mola_icp_debug_reload_env()


def _against_oracle(O, g, l, T, p, out, kd, step=1):
    valid, cen, nor, kidx, n = out
    sel = np.arange(0, l.shape[1], step)
    ov, oc, on, ok, onum = O.match_point2plane(g, np.ascontiguousarray(l[:, sel]), T, p.matcher_threshold, p.plane_eigen_threshold, int(p.knn), kd)
    assert np.array_equal(kidx[sel], ok), f"{(kidx[sel] != ok).any(1).sum()} of {len(ok)} neighbour lists differ"
    assert np.array_equal(valid[sel], ov)
    if step == 1:
        assert n == onum
    k = ov.astype(bool)
    np.testing.assert_allclose(cen[sel][k], oc[k], atol=1e-12)
    np.testing.assert_allclose(np.abs((nor[sel][k] * on[k]).sum(1)), 1.0, atol=1e-9)   # normals up to sign


@pytest.mark.parametrize("knn,n,lpq", [(3, 9000, 4), (4, 8207, 2), (5, 12_345, 4), (6, 20_011, 2), (6, 20_011, 4), (7, 9001, 2), (8, 8192, 2), (8, 8192, 4),
                                       (9, 10_000, 4), (3, 777, 2), (6, 33, 2), (6, 20_011, 1), (9, 10_000, 1), (3, 9001, 1), (6, 33, 1)])
def test_every_list_length_first_seeded_and_certified_launches(pkg, O, synth, knn, n, lpq):
    """list lengths 4 .. 10 at four lanes per query and at one, 4 .. 9 at two; N not a multiple of 16, 32 or 64 (the last wave / workgroup partly padding,
    a cloud smaller than one wave's share): a first launch (key-bootstrapped
    seeds), a launch 1 cm away (seeded, few certificates), one 0.1 mm further (nearly every query certified: most waves skip the sweep)"""
    g, l, _ = synth.make_pair(n, 30_000, seed=3 + knn)
    p = pkg.Parameters.load_from_file(REGULAR)
    p.knn = knn
    kd = O.KdTree(g)
    icp = pkg.ICP(device=0)
    icp.set_map(g)
    icp.set_local(l)
    try:
        _reload(pkg, MOLA_ICP_KNN_Q4="1", MOLA_ICP_KNN_Q4_LPQ=str(lpq))   # (lanes per query: four = 16 queries per wave, two = 32)
        for x in ([0.05, -0.02, 0.01, 0.004, 0.001, -0.002], [0.04, -0.02, 0.01, 0.004, 0.001, -0.002], [0.0401, -0.02, 0.01, 0.004, 0.001, -0.002]):
            T = pkg.pose_from_xyzypr(x)
            _against_oracle(O, g, l, T, p, icp.match_planes(T, p, n), kd)
    finally:
        _reload(pkg, MOLA_ICP_KNN_Q4=None, MOLA_ICP_KNN_Q4_LPQ=None)
    icp.close()


@pytest.mark.parametrize("lpq", [4, 2, 1])
def test_large_map_box_levels_from_global_memory(pkg, O, synth, lpq):
    """30 011 queries against a 2M-point map: sixteen top boxes, 977 super-tiles -- 24 KB of box levels, more than k_knn_q4 keeps in LDS
    beside its rings; a far launch (lists overflow and resume) and a near one"""
    g, l, _ = synth.make_pair(30_011, 2_000_000, seed=17)
    p = pkg.Parameters.load_from_file(REGULAR)
    kd = O.KdTree(g)
    icp = pkg.ICP(device=0)
    icp.set_map(g)
    icp.set_local(l)
    try:
        _reload(pkg, MOLA_ICP_KNN_Q4="1", MOLA_ICP_KNN_Q4_LPQ=str(lpq))
        for x in ([0.5, 0.2, 0.05, 0.03, 0.01, 0.005], [0.05, -0.02, 0.01, 0.004, 0.001, -0.002], [0.0502, -0.02, 0.01, 0.004, 0.001, -0.002]):
            T = pkg.pose_from_xyzypr(x)
            _against_oracle(O, g, l, T, p, icp.match_planes(T, p, l.shape[1]), kd, step=5)
    finally:
        _reload(pkg, MOLA_ICP_KNN_Q4=None, MOLA_ICP_KNN_Q4_LPQ=None)
    icp.close()


@pytest.mark.parametrize("lpq", [4, 2, 1])
def test_exact_ties_resolve_by_original_index(pkg, O, lpq):
    """a lattice map with every point duplicated, queries at cell centres: sixteen points at the same distance compete for six places --
    the packed (d2, original index) keys of the four sub-lanes' lists and of their merge must order them as the oracle does"""
    ax = np.arange(22, dtype=np.float32) * np.float32(0.25)
    cell = np.stack(np.meshgrid(ax, ax, ax, indexing="ij")).reshape(3, -1)
    g = np.ascontiguousarray(np.concatenate([cell, cell], axis=1))
    l = np.ascontiguousarray((cell[:, :9000] + np.float32(0.125)).astype(np.float32))
    p = pkg.Parameters.load_from_file(REGULAR)
    icp = pkg.ICP(device=0)
    icp.set_map(g)
    icp.set_local(l)
    kd = O.KdTree(g)
    try:
        _reload(pkg, MOLA_ICP_KNN_Q4="1", MOLA_ICP_KNN_Q4_LPQ=str(lpq))
        out = icp.match_planes(np.eye(4), p, l.shape[1])
        out2 = icp.match_planes(np.eye(4), p, l.shape[1])      # seeded by itself: the ties are met again
    finally:
        _reload(pkg, MOLA_ICP_KNN_Q4=None, MOLA_ICP_KNN_Q4_LPQ=None)
    ov, oc, on, ok, onum = O.match_point2plane(g, l, np.eye(4), p.matcher_threshold, p.plane_eigen_threshold, int(p.knn), kd)
    assert np.array_equal(out[3], ok) and np.array_equal(out2[3], ok)
    assert np.array_equal(out[0], ov) and np.array_equal(out2[0], ov) and out[4] == onum
    icp.close()


def test_the_switch_changes_nothing(pkg, synth):
    """the shipped pipeline (Point2Plane + Gauss-Newton) through k_knn_q4 everywhere, nowhere, and by the default rule (far launches on
    k_knn_coop, near ones on k_knn_q4 -- the kernel changes INSIDE the align, and with it who wrote the seeds the next launch reads):
    the same pose, iteration count, quality and pairing, bit for bit; sizes beyond k_knn_coop's range too (k_knn_q4 with two lanes per query,
    with one / the persistent kernel), and a lockstep batch"""
    p = pkg.Parameters.load_from_file(REGULAR)
    for n, m in ((40_000, 60_000), (150_000, 150_000), (300_000, 280_000)):   # (by the default rule: four+two / two / one lane(s) per query)
        g, l, _ = synth.make_pair(n, m, seed=23)
        outs = []
        try:
            for q, lpq in (("1", "4"), ("0", None), (None, None), ("1", "2"), ("1", "1")):
                _reload(pkg, MOLA_ICP_KNN_Q4=q, MOLA_ICP_KNN_Q4_LPQ=lpq)
                icp = pkg.ICP(device=0)
                a = icp.align(g, l, np.eye(4), p)
                icp.set_map(g)
                icp.set_local(l)
                pl = icp.match_planes(a.optimal_tf, p, n)
                b = icp.align_batch([(g, l), (g[:, : m // 2], l[:, : n // 2])], [np.eye(4)] * 2, p) if n < 100_000 else []
                outs.append((a.optimal_tf, a.nIterations, a.quality, a.n_pairs, pl[0], pl[1], pl[2], pl[3], pl[4]) + tuple(x.optimal_tf for x in b) + tuple(x.nIterations for x in b))
                icp.close()
        finally:
            _reload(pkg, MOLA_ICP_KNN_Q4=None, MOLA_ICP_KNN_Q4_LPQ=None)
        for other in outs[1:]:
            for x, y in zip(outs[0], other):
                assert np.array_equal(x, y)


def test_loop_closure_guesses_in_one_launch(pkg, synth):
    """align_multi_init (the loop-closure Monte-Carlo: src/LidarOdometry.cpp:767-788): ten poses on one pair in lockstep, blockIdx.y =
    problem -- through k_knn_q4 and through k_knn_coop the same results, and each the stand-alone align's"""
    g, l, _ = synth.make_pair(20_000, 25_000, seed=31)
    p = pkg.Parameters.load_from_file(REGULAR)
    rng = np.random.default_rng(5)
    guesses = [synth.pose_from_xyzypr(*(rng.normal(0, 1, 3) * 0.2), rng.normal(0, 1) * 0.02, 0, 0) for _ in range(10)]
    outs = []
    try:
        for q, lpq in (("1", "4"), ("0", None), ("1", "2"), ("1", "1")):
            _reload(pkg, MOLA_ICP_KNN_Q4=q, MOLA_ICP_KNN_Q4_LPQ=lpq)
            icp = pkg.ICP(device=0)
            res, best = icp.align_multi_init(g, l, guesses, p)
            alone = [icp.align(g, l, T, p) for T in guesses[:3]]
            for r, a in zip(res, alone):
                assert r.nIterations == a.nIterations and np.array_equal(r.optimal_tf, a.optimal_tf) and r.quality == a.quality
            outs.append(([r.optimal_tf for r in res], [r.nIterations for r in res], best))
            icp.close()
    finally:
        _reload(pkg, MOLA_ICP_KNN_Q4=None, MOLA_ICP_KNN_Q4_LPQ=None)
    for other in outs[1:]:
        assert outs[0][1] == other[1] and outs[0][2] == other[2]
        for x, y in zip(outs[0][0], other[0]):
            assert np.array_equal(x, y)
