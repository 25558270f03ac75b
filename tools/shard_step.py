#!/usr/bin/env python3
"""Step time of ONE rank's share of a query-sharded job, alone on the GPU (no collective): what a rank of an N-GPU run spends per
iteration before the all-reduce -- the projection DESIGN.md section 6 quotes while no multi-GPU node is at hand.
  --config c3 : configs[2]'s 1M x 1M pair (the headline), whole map on every rank: Hilbert shards against index-order shards
  --config c5 : configs[4], a 10M-point map x 1M queries: the device-cut shard of every rank in turn + its map slab (margin from
                the guess: sharded.slab_margin_for_guess), the slowest rank of each world size = that world's step
One JSON line per world size on stdout."""
import argparse
import importlib
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ap = argparse.ArgumentParser()
ap.add_argument("--config", choices=["c3", "c5"], default="c3")
ap.add_argument("--iters", type=int, default=20)
ap.add_argument("--worlds", default="1,2,4,8")
ap.add_argument("--n-map", type=int, default=0)
ap.add_argument("--balance-rounds", type=int, default=2)
ap.add_argument("--relax", type=float, default=1.0)
args = ap.parse_args()
pkg = importlib.import_module("mola-fe-lidar_amd")
synth = importlib.import_module("mola-fe-lidar_amd.synth")
sharded = importlib.import_module("mola-fe-lidar_amd.sharded")
GATE = 1.0
p = pkg.Parameters()
p.matcher_threshold, p.fixed_iterations, p.skip_quality, p.max_iterations = GATE, 1, 1, args.iters
worlds = [int(w) for w in args.worlds.split(",")]


def timed(icp):
    icp.align_resident(np.eye(4), p)
    icp.align_resident(np.eye(4), p)
    ts = []
    for _ in range(3):   # (the median of three aligns: one stall of a few hundred microseconds in a 20-iteration align is +50 % on its step)
        t0 = time.perf_counter()
        icp.align_resident(np.eye(4), p)
        ts.append((time.perf_counter() - t0) / args.iters)
    dt = sorted(ts)[1]
    icp.set_profiling(True)
    r = icp.align_resident(np.eye(4), p)
    icp.set_profiling(False)
    return dt * 1e3, r.ms_nn_kernel / max(1, r.n_nn_launches)


if args.config == "c3":
    g, l, _ = synth.make_pair(1_000_000, args.n_map or 1_000_000, seed=42)
    order = sharded.spatial_order(l)
    icp = pkg.ICP(device=0)
    icp.set_map(g)
    for world in worlds:
        lo, hi = sharded.shard_bounds(l.shape[1], 0, world)
        for name, sh in (("z-order", l[:, order[lo:hi]]), ("index-order", l[:, lo:hi])):
            icp.set_local(np.ascontiguousarray(sh))
            icp.set_global_sizes(l.shape[1], g.shape[1])
            ms, k_ms = timed(icp)
            print(json.dumps({"config": "c3", "world": world, "rank": 0, "shard": name, "queries": hi - lo, "ms_per_iteration": ms,
                              "matcher_ms_per_launch": k_ms}), flush=True)
else:
    import torch
    M = args.n_map or 10_000_000
    g, l, _ = synth.make_pair(1_000_000, M, seed=42)
    tg, tl = torch.from_numpy(g).cuda(), torch.from_numpy(np.ascontiguousarray(l)).cuda()
    icp = pkg.ICP(device=0)
    base = None
    def measure(world, cuts):
        rows = []
        for rank in range(world):
            if world == 1:
                icp.set_map(tg)
                icp.set_local(tl)
                n_shard, kept, margin = l.shape[1], M, 0.0
            else:
                n_shard = icp.set_local_shard_range(tl, cuts[rank], cuts[rank + 1])
                blo0, bhi0 = icp.shard_reach_box(np.eye(4), 0.0)
                margin = sharded.slab_margin_for_guess(blo0, bhi0, GATE, 1.0, np.deg2rad(3.0))
                blo, bhi = icp.shard_reach_box(np.eye(4), margin)
                kept = icp.set_map_slab(tg, blo, bhi)
            icp.set_global_sizes(l.shape[1], M)
            ms, k_ms = timed(icp)
            rows.append({"rank": rank, "queries": n_shard, "map_points_kept": kept, "margin_m": margin, "ms_per_iteration": ms,
                         "matcher_ms_per_launch": k_ms})
        return rows

    for world in worlds:
        n = l.shape[1]
        cuts = [sharded.shard_bounds(n, r, world)[0] for r in range(world)] + [n]
        for rnd in range(1 if world == 1 else 1 + args.balance_rounds):
            rows = measure(world, cuts)
            worst = max(rows, key=lambda r: r["ms_per_iteration"])
            base = base or worst["ms_per_iteration"]
            print(json.dumps({"config": "c5", "n_map": M, "world": world, "cuts": "equal counts" if rnd == 0 else f"cost-balanced, round {rnd}",
                              "step_ms_slowest_rank": worst["ms_per_iteration"],
                              "projected_speedup_before_collective": base / worst["ms_per_iteration"],
                              "step_ms_mean_rank": float(np.mean([r["ms_per_iteration"] for r in rows])), "ranks": rows}), flush=True)
            cuts = sharded.balanced_cuts(cuts, [r["ms_per_iteration"] for r in rows], relax=args.relax)
