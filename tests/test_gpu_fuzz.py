"""-m gpu: randomised parity -- random cloud sizes (1 ... 40k, the awkward ones over-represented), gates, 5-launch pose
sequences, queries without neighbours, duplicated map points, and a random kernel flavour per case (cooperative / persistent
matchers, 64- / 128-query items, quad / pass-by-pass sweep, queue knobs, blocks per CU): every launch of the NN matcher and of the plane matcher against the
CPU oracle, bit for bit.  80 cases here (seconds); MOLA_ICP_FUZZ_CASES / MOLA_ICP_FUZZ_SEED run more (860 cases were run on the
round's final kernels)."""
import os

import numpy as np
import pytest

pytestmark = [pytest.mark.gpu, pytest.mark.timeout(1200)]

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KNOBS = ("MOLA_ICP_COOP", "MOLA_ICP_KNN_COOP", "MOLA_ICP_QPL", "MOLA_ICP_EARLY_POP", "MOLA_ICP_NO_LPT", "MOLA_ICP_BLOCKS_PER_CU", "MOLA_ICP_QUADS",
         "MOLA_ICP_Q4", "MOLA_ICP_KNN_Q4", "MOLA_ICP_KNN_Q4_LPQ")


def test_random_cases_equal_the_oracle(pkg, O, synth):
    n_cases = int(os.environ.get("MOLA_ICP_FUZZ_CASES", "80"))
    max_n = int(os.environ.get("MOLA_ICP_FUZZ_MAXN", "40000"))   # (larger: several entries per persistent wave -- the work queue at work)
    rng = np.random.default_rng(int(os.environ.get("MOLA_ICP_FUZZ_SEED", "1")))
    scene = synth.Scene(scene_seed=3, half=12.0, wall_y=5.0, wall_h=4.0, n_boxes=8)
    p2pl = pkg.Parameters.load_from_file(os.path.join(ROOT, "params", "icp-settings-regular.yaml"))
    saved = {k: os.environ.get(k) for k in KNOBS}
    failures = []
    try:
        for case in range(n_cases):
            for k in KNOBS:
                os.environ.pop(k, None)
            env = {"MOLA_ICP_COOP": str(rng.integers(0, 2)), "MOLA_ICP_KNN_COOP": str(rng.integers(0, 2)), "MOLA_ICP_QPL": str(rng.integers(1, 3))}
            if rng.random() < 0.2:
                env["MOLA_ICP_EARLY_POP"] = "1"
            if rng.random() < 0.2:
                env["MOLA_ICP_NO_LPT"] = "1"
            if rng.random() < 0.3:
                env["MOLA_ICP_BLOCKS_PER_CU"] = str(rng.integers(1, 5))
            u = rng.random()   # (unset: the quad sweep for seeded launches, pass by pass without seeds; "0" / "1": one of them always)
            if u < 0.3:
                env["MOLA_ICP_QUADS"] = "0"
            elif u < 0.5:
                env["MOLA_ICP_QUADS"] = "1"
            # (round 6: the four-lanes-per-query kernels -- unset = the product's rule, "0" never, "1" at every launch, whatever the size)
            u = rng.random()
            if u < 0.5:
                env["MOLA_ICP_Q4"] = "1" if u < 0.35 else "0"
            u = rng.random()
            if u < 0.5:
                env["MOLA_ICP_KNN_Q4"] = "1" if u < 0.35 else "0"
            if rng.random() < 0.5:
                env["MOLA_ICP_KNN_Q4_LPQ"] = str(rng.choice([1, 2, 4]))
            os.environ.update(env)
            pkg._lib.lib().mola_icp_debug_reload_env()
            N = int(rng.choice([1, 63, 64, 65, 127, 129, 1000, 4097, 9000, 20000, 33333]) if rng.random() < 0.5 else rng.integers(1, max_n))
            M = int(rng.choice([1, 17, 64, 2047, 2049, 10000, 30000]) if rng.random() < 0.5 else rng.integers(1, max_n))
            g, l, _ = synth.make_pair(N, M, seed=int(rng.integers(1, 10**6)), scene=scene if max(N, M) <= 60000 else None)
            if rng.random() < 0.3 and N > 10:
                l = l.copy()
                l[2, : N // 7] += 40.0          # queries without neighbours
            if rng.random() < 0.2 and M > 10:
                g = np.ascontiguousarray(np.concatenate([g, g[:, : M // 3]], axis=1))   # duplicated map points: exact ties
            kd = O.KdTree(g)
            icp = pkg.ICP(device=0)
            icp.set_map(g)
            icp.set_local(l)
            x = np.zeros(6)
            knn = int(rng.integers(3, 9))
            p2pl.knn = knn
            for launch in range(5):
                x = x + rng.normal(0, 1, 6) * np.array([0.2, 0.2, 0.05, 0.02, 0.005, 0.005]) * (0.3 ** launch if rng.random() < 0.7 else 1.0)
                T = synth.pose_from_xyzypr(*x)
                thr = float(rng.choice([0.3, 0.5, 0.7, 1.0]))
                idx, d2, n = icp.match(T, thr, N, pkg.NN_TILED)
                oidx, od2, on = O.match(g, l, T, thr, kd)
                if n != on or not np.array_equal(idx, oidx) or not np.array_equal(d2[oidx >= 0], od2[oidx >= 0]):
                    failures.append(f"case {case} launch {launch}: NN mismatch N={N} M={g.shape[1]} thr={thr} env={env}")
                if g.shape[1] >= 3:
                    p2pl.matcher_threshold = thr
                    valid, cen, nor, kidx, npl = icp.match_planes(T, p2pl, N)
                    ov, oc, onn, okn, onum = O.match_point2plane(g, l, T, thr, p2pl.plane_eigen_threshold, knn, kd)
                    if npl != onum or not np.array_equal(kidx, okn) or not np.array_equal(valid, ov):
                        failures.append(f"case {case} launch {launch}: plane mismatch N={N} M={g.shape[1]} knn={knn} thr={thr} env={env}")
            icp.close()
    finally:
        for k, v in saved.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
        pkg._lib.lib().mola_icp_debug_reload_env()
    assert not failures, failures[:5]


def test_large_random_cases_equal_the_oracle_on_a_sample(pkg, O, synth):
    """300k ... 900k points: every persistent wave takes SEVERAL entries from the work queue (segments, steals, dry-set bits,
    heavy-first orders, redo of tied items) -- the sizes where the hand-rolled queue matters.  Random sizes, gates, 4-launch
    pose sequences and queue knobs; every launch of the NN matcher and of the plane matcher against the oracle's exact
    kd-tree on every 20th query, bit for bit; the reported pair counts against the whole returned pairing."""
    n_cases = int(os.environ.get("MOLA_ICP_FUZZ_LARGE_CASES", "6"))
    rng = np.random.default_rng(int(os.environ.get("MOLA_ICP_FUZZ_SEED", "1")) + 77)
    p2pl = pkg.Parameters.load_from_file(os.path.join(ROOT, "params", "icp-settings-regular.yaml"))
    saved = {k: os.environ.get(k) for k in KNOBS}
    failures = []
    try:
        for case in range(n_cases):
            for k in KNOBS:
                os.environ.pop(k, None)
            env = {"MOLA_ICP_QPL": str(rng.integers(1, 3))}
            if rng.random() < 0.3:
                env["MOLA_ICP_EARLY_POP"] = "1"
            if rng.random() < 0.3:
                env["MOLA_ICP_NO_LPT"] = "1"
            if rng.random() < 0.5:
                env["MOLA_ICP_BLOCKS_PER_CU"] = str(rng.integers(1, 5))
            u = rng.random()
            if u < 0.3:
                env["MOLA_ICP_QUADS"] = "0"
            elif u < 0.5:
                env["MOLA_ICP_QUADS"] = "1"
            u = rng.random()   # (300k .. 900k queries: k_knn_q4's own range ends at 0.56M; "1" runs it -- and k_nn_q4 -- beyond)
            if u < 0.5:
                env["MOLA_ICP_KNN_Q4"] = "1" if u < 0.3 else "0"
            if rng.random() < 0.3:
                env["MOLA_ICP_Q4"] = "1"
            if rng.random() < 0.4:
                env["MOLA_ICP_KNN_Q4_LPQ"] = str(rng.choice([1, 2, 4]))
            os.environ.update(env)
            pkg._lib.lib().mola_icp_debug_reload_env()
            N = int(rng.integers(300_000, 900_001))
            M = int(rng.integers(300_000, 900_001))
            g, l, _ = synth.make_pair(N, M, seed=int(rng.integers(1, 10**6)))
            if rng.random() < 0.5:
                l = l.copy()
                l[2, : N // 9] += 40.0          # queries without neighbours
            if rng.random() < 0.5:
                g = np.ascontiguousarray(np.concatenate([g, g[:, : M // 5]], axis=1))   # duplicated map points: exact ties -> redone items
            kd = O.KdTree(g)
            sel = np.arange(int(rng.integers(0, 20)), N, 20)
            ls = np.ascontiguousarray(l[:, sel])
            icp = pkg.ICP(device=0)
            icp.set_map(g)
            icp.set_local(l)
            x = np.zeros(6)
            knn = int(rng.integers(3, 9))
            p2pl.knn = knn
            for launch in range(4):
                x = x + rng.normal(0, 1, 6) * np.array([0.2, 0.2, 0.05, 0.02, 0.005, 0.005]) * (0.3 ** launch if rng.random() < 0.7 else 1.0)
                T = synth.pose_from_xyzypr(*x)
                thr = float(rng.choice([0.5, 0.7, 1.0]))
                idx, d2, n = icp.match(T, thr, N, pkg.NN_TILED)
                oidx, od2, _ = O.match(g, ls, T, thr, kd)
                if not np.array_equal(idx[sel], oidx) or not np.array_equal(d2[sel][oidx >= 0], od2[oidx >= 0]) or n != int((idx >= 0).sum()):
                    failures.append(f"case {case} launch {launch}: NN mismatch N={N} M={g.shape[1]} thr={thr} env={env}")
                p2pl.matcher_threshold = thr
                valid, cen, nor, kidx, npl = icp.match_planes(T, p2pl, N)
                ov, oc, onn, okn, _ = O.match_point2plane(g, ls, T, thr, p2pl.plane_eigen_threshold, knn, kd)
                if not np.array_equal(kidx[sel], okn) or not np.array_equal(valid[sel], ov) or npl != int(valid.sum()):
                    failures.append(f"case {case} launch {launch}: plane mismatch N={N} M={g.shape[1]} knn={knn} thr={thr} env={env}")
            icp.close()
    finally:
        for k, v in saved.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
        pkg._lib.lib().mola_icp_debug_reload_env()
    assert not failures, failures[:5]
