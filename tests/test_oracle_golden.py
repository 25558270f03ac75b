"""The oracle (oracle/icp_oracle.c) against the independent numpy/scipy golden vectors
(tests/golden/make_golden.py).  The reference itself holds no vectors for this path
(parity unpinned) -- this is the strongest pin available."""
import numpy as np
import pytest


def test_transform_bit_exact(O, golden):
    q = O.transform(np.eye(4), golden["A_local"])
    assert np.array_equal(q, golden["A_q0"])
    # a non-trivial pose too: compare against the trace pose
    T = golden["A_trace"][0]
    from tests.golden import make_golden as G
    assert np.array_equal(O.transform(T, golden["A_local"][:, :500]), G.transform32(T, golden["A_local"][:, :500]))


@pytest.mark.parametrize("use_tree", [False, True])
def test_match_indices_bit_exact(O, golden, use_tree):
    g, l = golden["A_map"], golden["A_local"]
    tree = O.KdTree(g) if use_tree else None
    idx, d2, n = O.match(g, l, np.eye(4), 1.0, tree)
    assert np.array_equal(idx, golden["A_idx0"])
    keep = idx >= 0
    assert np.array_equal(d2[keep], golden["A_d20"][keep])
    assert n == int(keep.sum())


def test_kdtree_equals_brute_with_ties(O):
    # lattice points give many exact ties: the lowest index must win in both searches
    ax = np.arange(8, dtype=np.float32)
    g = np.stack(np.meshgrid(ax, ax, ax, indexing="ij")).reshape(3, -1)
    g = np.concatenate([g, g], axis=1)  # every point duplicated -> ties on every query
    q = (g[:, :512] + np.float32(0.5)).astype(np.float32)
    i1, d1 = O.nn_brute(g, q)
    i2, d2 = O.KdTree(g).nn(q)
    assert np.array_equal(i1, i2) and np.array_equal(d1, d2)
    assert (i1 < 512).all()


def test_accumulators(O, golden):
    g, l = golden["A_map"], golden["A_local"]
    acc = O.accumulate(g, l, golden["A_idx0"], golden["A_d20"], O.params(), np.eye(4))
    np.testing.assert_allclose(acc, golden["A_acc0"], rtol=1e-12, atol=1e-9)


def test_horn_matches_eigh_and_svd(O, golden):
    T = O.horn(golden["A_acc0"])
    np.testing.assert_allclose(T, golden["A_T1"], atol=1e-11)


def test_align_trace_and_result(O, golden):
    g, l = golden["A_map"], golden["A_local"]
    for kd in (False, True):
        r = O.align(g, l, np.eye(4), O.params(max_iterations=30, matcher_threshold=1.0, use_kdtree=kd), trace=True)
        assert r["n_iterations"] == int(golden["A_nit"])
        assert r["termination"] == int(golden["A_term"])
        np.testing.assert_allclose(r["trace"][:5], golden["A_trace"], atol=1e-10)
        np.testing.assert_allclose(r["T"], golden["A_Tfinal"], atol=1e-9)
        assert r["quality"] == pytest.approx(float(golden["A_quality"]), abs=1e-12)


@pytest.mark.parametrize("name", ["identity", "trans", "yaw", "pitch", "roll", "se3"])
def test_known_answer_transforms(O, golden, name):
    g, l = golden["B_map"], golden[f"B_{name}_local"]
    idx, d2, _ = O.match(g, l, np.eye(4), 0.5)
    assert np.array_equal(idx, golden[f"B_{name}_idx0"])
    acc = O.accumulate(g, l, idx, d2, O.params(), np.eye(4))
    np.testing.assert_allclose(O.horn(acc), golden[f"B_{name}_T1"], atol=1e-10)
    # ICP on a noise-free moved copy recovers the motion to the north-star tolerance
    r = O.align(g, l, np.eye(4), O.params(max_iterations=100, matcher_threshold=0.5))
    rot, trans = O.pose_error(r["T"], golden[f"B_{name}_T"])
    assert rot <= 1e-4 and trans <= 1e-3, (rot, trans)


def test_se3_log_and_pose_conversions(O, golden):
    for p, T, lg in zip(golden["C_xyzypr"], golden["C_T"], golden["C_log"]):
        np.testing.assert_allclose(O.pose_from_xyzypr(p), T, atol=1e-14)
        np.testing.assert_allclose(O.pose_from_xyzypr(O.pose_to_xyzypr(T)), T, atol=1e-12)
        np.testing.assert_allclose(O.se3_log(T), lg, atol=1e-9)


def test_edge_cases(O, golden):
    g, l = golden["A_map"], golden["A_local"]
    empty = np.zeros((3, 0), dtype=np.float32)
    for gm, lm in ((empty, l), (g, empty), (empty, empty)):
        r = O.align(gm, lm, np.eye(4), O.params())
        assert r["termination"] == 1 and r["n_iterations"] == 0 and r["quality"] == 0
        assert np.array_equal(r["T"], np.eye(4))
    # everything beyond the gate -> NoPairings, pose untouched
    far = (l + np.float32(1000.0)).astype(np.float32)
    r = O.align(g, far, np.eye(4), O.params(matcher_threshold=0.5))
    assert r["termination"] == 1 and r["n_pairs"] == 0
