#!/usr/bin/env python3
"""BASELINE configs[3] on ONE GPU: independent 100k x 100k scan pairs through mola_icp_align_batch (stream-per-pair,
host buffers, <=100 iterations with the stall test) and through the cloud cache.  Prints pairs/s and aggregate it/s."""
import argparse
import importlib
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ap = argparse.ArgumentParser()
ap.add_argument("--pairs", type=int, default=8)
ap.add_argument("--n", type=int, default=100_000)
a = ap.parse_args()
pkg = importlib.import_module("mola-fe-lidar_amd")
synth = importlib.import_module("mola-fe-lidar_amd.synth")
pairs = [synth.make_pair(a.n, a.n, seed=100 + s)[:2] for s in range(a.pairs)]
p = pkg.Parameters()
p.max_iterations, p.matcher_threshold, p.min_abs_step_trans, p.min_abs_step_rot = 100, 1.0, 5e-5, 1e-5
icp = pkg.ICP(device=0)
icp.set_profiling(True)  # kernel times / executed pairs are printed below
icp.align(pairs[0][0], pairs[0][1], np.eye(4), p)  # warm-up
t0 = time.perf_counter()
res = icp.align_batch(pairs, [np.eye(4)] * len(pairs), p)
dt = time.perf_counter() - t0
its = sum(r.nIterations for r in res)
print(f"align_batch: {len(pairs)} pairs of {a.n}x{a.n} in {dt*1e3:.1f} ms = {len(pairs)/dt:.1f} pairs/s, "
      f"{its} iterations total = {its/dt:.0f} it/s aggregate (avg {its/len(pairs):.1f} its/pair)")
t0 = time.perf_counter()
res1 = [icp.align(g, l, np.eye(4), p) for g, l in pairs]
dt1 = time.perf_counter() - t0
print(f"sequential align: {dt1*1e3:.1f} ms = {len(pairs)/dt1:.1f} pairs/s")
assert all(np.array_equal(x.optimal_tf, y.optimal_tf) for x, y in zip(res, res1))
for k, (g, l) in enumerate(pairs):
    icp.cloud_put(2 * k, g)
    icp.cloud_put(2 * k + 1, l)
t0 = time.perf_counter()
res2 = [icp.align_cached(2 * k, 2 * k + 1, np.eye(4), p) for k in range(len(pairs))]
dt2 = time.perf_counter() - t0
print(f"align_cached (clouds resident + prepared): {dt2*1e3:.1f} ms = {len(pairs)/dt2:.1f} pairs/s")
assert all(np.array_equal(x.optimal_tf, y.optimal_tf) for x, y in zip(res, res2))
