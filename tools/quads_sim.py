"""What per-16-lane tile lists inside a 64-query item would evaluate (CPU estimate on the bench's 1M x 1M pair)."""
import importlib, os, sys, time
import numpy as np
from scipy.spatial import cKDTree
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
synth = importlib.import_module("mola-fe-lidar_amd.synth")
from test_gpu_prepare import hilbert_keys
N = 1_000_000; M = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000   # (10000000: configs[4])
g, l, Tgt = synth.make_pair(N, M, seed=42)
def order(pc):
    k = hilbert_keys(pc)
    return np.argsort(k, kind="stable")
t0 = time.time()
go = order(g); G = g[:, go].T.astype(np.float64)            # sorted map
# tile boxes (32 points)
nt = M // 32
tb_lo = G[:nt * 32].reshape(nt, 32, 3).min(1); tb_hi = G[:nt * 32].reshape(nt, 32, 3).max(1)
tree = cKDTree(G)
tile_tree = cKDTree((tb_lo + tb_hi) / 2)
tile_rad = np.linalg.norm((tb_hi - tb_lo) / 2, axis=1)
print("prep %.1fs, max tile radius %.3f median %.3f" % (time.time() - t0, tile_rad.max(), np.median(tile_rad)))
for name, T, slack in (("near convergence (ground-truth pose, bound = 1.1 x NN distance)", Tgt, 1.1), ("creeping regime (pose 3 cm / 0.1 deg off, bound = previous NN)", None, 1.0)):
    if T is None:
        d = synth.pose_from_xyzypr(0.03, 0.0, 0.0, np.deg2rad(0.1), 0, 0)
        T = d @ Tgt
        Tprev = Tgt
    else:
        Tprev = None
    lo_ = order(l)                                          # (queries are sorted in their own frame)
    Q = (l[:, lo_].T.astype(np.float64) @ T[:3, :3].T + T[:3, 3])
    rng = np.random.default_rng(1)
    items = rng.choice(N // 64, 400, replace=False)
    it_tiles, q_tiles, quad_max, quad_mean, per_query = [], [], [], [], []
    for it in items:
        q = Q[it * 64:(it + 1) * 64]
        if Tprev is not None:   # bound: distance (at the new pose) to the neighbour found at the previous pose
            qp = (l[:, lo_].T[it * 64:(it + 1) * 64].astype(np.float64) @ Tprev[:3, :3].T + Tprev[:3, 3])
            _, j = tree.query(qp)
            bound = np.linalg.norm(q - G[j], axis=1)
        else:
            dnn, _ = tree.query(q)
            bound = dnn * slack
        bound = np.minimum(bound, 1.0)
        sets = []
        for k in range(64):
            cand = tile_tree.query_ball_point(q[k], bound[k] + tile_rad.max())
            cand = np.asarray(cand, int)
            gap = np.maximum(np.maximum(tb_lo[cand] - q[k], q[k] - tb_hi[cand]), 0.0)
            sets.append(set(cand[(gap ** 2).sum(1) <= bound[k] ** 2].tolist()))
        u = set().union(*sets)
        it_tiles.append(len(u))
        per_query.append(np.mean([len(s) for s in sets]))
        qs = [len(set().union(*sets[16 * a:16 * a + 16])) for a in range(4)]
        quad_max.append(max(qs)); quad_mean.append(np.mean(qs))
    it_tiles, quad_max, quad_mean, per_query = map(np.asarray, (it_tiles, quad_max, quad_mean, per_query))
    print(name)
    print("  tiles a single query needs: mean %.1f" % per_query.mean())
    print("  64-query item: tiles in the union mean %.1f (pairs/query %.0f)   passes of two tiles: %.1f" % (it_tiles.mean(), 32 * it_tiles.mean(), np.ceil(it_tiles / 2).mean()))
    print("  16-query quads: mean %.1f tiles (pairs/query %.0f), the fullest quad of an item %.1f -> passes %.1f (= %.0f %% of the item's)" % (
        quad_mean.mean(), 32 * quad_mean.mean(), quad_max.mean(), np.ceil(quad_max / 2).mean(), 100 * np.ceil(quad_max / 2).mean() / np.ceil(it_tiles / 2).mean()))
