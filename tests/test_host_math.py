"""Product host-side math (mola_icp_solve_horn / se3_log / pose conversions / stall test)
against the golden vectors and the oracle.  No GPU needed."""
import numpy as np
import pytest


def test_pose_conversions_and_log(pkg, golden):
    for p, T, lg in zip(golden["C_xyzypr"], golden["C_T"], golden["C_log"]):
        np.testing.assert_allclose(pkg.pose_from_xyzypr(p), T, atol=1e-14)
        np.testing.assert_allclose(pkg.pose_from_xyzypr(pkg.pose_to_xyzypr(T)), T, atol=1e-12)
        np.testing.assert_allclose(pkg.se3_log(T), lg, atol=1e-9)


def test_log_near_identity_and_near_pi(pkg, O):
    for p in ([0, 0, 0, 0, 0, 0], [1e-9, 0, 0, 1e-10, 0, 0], [0.3, 0.1, -0.2, np.pi - 1e-7, 0, 0],
              [1, 2, 3, 0, np.pi / 2 - 1e-9, 0], [0, 0, 0, 0.7, -1.1, 2.9]):
        T = pkg.pose_from_xyzypr(p)
        a, b = pkg.se3_log(T), O.se3_log(T)
        # compare through the rotation angle (axis sign is ambiguous at pi)
        assert np.linalg.norm(a[3:]) == pytest.approx(np.linalg.norm(b[3:]), abs=1e-6)
        if np.linalg.norm(a[3:]) < 3.0:
            np.testing.assert_allclose(a, b, atol=1e-8)


def test_horn_on_golden_accumulators(pkg, O, golden):
    T = pkg.solve_horn(golden["A_acc0"])
    np.testing.assert_allclose(T, golden["A_T1"], atol=1e-11)
    np.testing.assert_allclose(T, O.horn(golden["A_acc0"]), atol=1e-12)
    R = T[:3, :3]
    np.testing.assert_allclose(R @ R.T, np.eye(3), atol=1e-13)
    assert np.linalg.det(R) == pytest.approx(1.0, abs=1e-13)


def test_horn_explicit_centroids(pkg, O, golden):
    acc = golden["A_acc0"]
    cl, cg = acc[1:4] / acc[0] + 0.01, acc[4:7] / acc[0] - 0.02
    np.testing.assert_allclose(pkg.solve_horn(acc, cl, cg), O.horn(acc, cl, cg), atol=1e-12)


@pytest.mark.parametrize("name", ["identity", "trans", "yaw", "pitch", "roll", "se3"])
def test_horn_known_answers(pkg, golden, name):
    # accumulators of the TRUE correspondences -> the known transform
    l, g = golden[f"B_{name}_local"].astype(np.float64), golden["B_map"].astype(np.float64)
    acc = np.zeros(24)
    acc[0] = acc[16] = l.shape[1]
    acc[1:4], acc[4:7], acc[7:16] = l.sum(1), g.sum(1), (l @ g.T).reshape(9)
    T = pkg.solve_horn(acc)
    np.testing.assert_allclose(T, golden[f"B_{name}_Ttrue"], atol=1e-10)
    assert np.abs(T - golden[f"B_{name}_T"]).max() < 2e-5


def _acc_of(l, g):
    acc = np.zeros(24)
    acc[0] = acc[16] = l.shape[1]
    acc[1:4], acc[4:7], acc[7:16] = l.sum(1), g.sum(1), (l @ g.T).reshape(9)
    return acc


def test_horn_degenerate(pkg, O):
    """csrc/se3_math.hpp, kSingularRel: a solve whose pairings do not determine the pose is REFUSED (the loop ends with
    SolverError) -- no pairings, fewer than three, a line (the rotation about it is free), all queries in one point -- and the
    checker refuses the same inputs; a plane of pairings is fine for Horn"""
    with pytest.raises(pkg.IcpError):
        pkg.solve_horn(np.zeros(24))
    t = np.linspace(-1, 1, 50)
    line = np.stack([t, 0.3 * t, -0.2 * t])
    shift = np.array([[0.1], [0.0], [0.05]])
    rng = np.random.default_rng(0)
    cases = {
        "two pairings": rng.normal(size=(3, 2)),
        "a line": line,
        "one point fifty times": np.repeat(rng.normal(size=(3, 1)), 50, axis=1),
    }
    for name, l in cases.items():
        acc = _acc_of(l, l + shift)
        with pytest.raises(pkg.IcpError):
            pkg.solve_horn(acc)
        with pytest.raises(ValueError):
            O.horn(acc)
    # coplanar pairings determine the pose
    plane = np.stack([rng.uniform(-1, 1, 40), rng.uniform(-1, 1, 40), np.zeros(40)])
    Tgt = pkg.pose_from_xyzypr([0.1, -0.2, 0.05, 0.02, -0.01, 0.03])
    g = Tgt[:3, :3] @ plane + Tgt[:3, 3:4]
    T = pkg.solve_horn(_acc_of(plane, g))
    np.testing.assert_allclose(T, Tgt, atol=1e-12)
    np.testing.assert_allclose(O.horn(_acc_of(plane, g)), Tgt, atol=1e-12)


def test_stall_deltas(pkg, O):
    Tp = pkg.pose_from_xyzypr([1, 2, 3, 0.1, 0.2, 0.3])
    d = pkg.pose_from_xyzypr([3e-5, -2e-5, 1e-5, 4e-6, -3e-6, 2e-6])
    T = Tp @ d
    dx, dr = pkg.stall_deltas(T, Tp)
    ox, orr = O.stall_deltas(T, Tp)
    assert dx == pytest.approx(ox, rel=1e-9) and dr == pytest.approx(orr, rel=1e-6)
    assert dx == pytest.approx(np.linalg.norm([3e-5, -2e-5, 1e-5]), rel=1e-4)
