#!/usr/bin/env python3
"""The numbers the headline bench line does not carry, on ONE GPU (run on the MI355X box, from the repo root):
  - per-iteration time of the point-to-point and the shipped point-to-plane pipeline on resident clouds at
    odometry sizes (100k x 100k, 120k x 120k) and at C3;
  - end-to-end `mola_icp_align` latency from host buffers (upload + sort + iterations with the stall test + quality);
  - the loop-closure Monte-Carlo: K guesses on one pair through `align_multi_init` vs K stand-alone aligns;
  - config[3]'s shape: independent 100k x 100k pairs through `align_batch`.
Prints one JSON object per line (`"what": ...`)."""
import argparse
import importlib
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ap = argparse.ArgumentParser()
ap.add_argument("--pairs", type=int, default=16)
ap.add_argument("--guesses", type=int, default=10)
ap.add_argument("--skip-1m", action="store_true")
a = ap.parse_args()
pkg = importlib.import_module("mola-fe-lidar_amd")
synth = importlib.import_module("mola-fe-lidar_amd.synth")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def emit(**kw):
    print(json.dumps(kw), flush=True)


icp = pkg.ICP(device=0)


p2p = pkg.Parameters()
p2p.matcher_threshold = 1.0
p2pl = pkg.Parameters.load_from_file(os.path.join(ROOT, "params", "icp-settings-regular.yaml"))

sizes = [(100_000, 100_000), (120_000, 120_000)] + ([] if a.skip_1m else [(1_000_000, 1_000_000)])
for n, m in sizes:
    g, l, _ = synth.make_pair(n, m, seed=42)
    icp.set_map(g)
    icp.set_local(l)
    for name, base in (("p2p", p2p), ("p2pl", p2pl)):
        p = base.copy()
        p.fixed_iterations, p.skip_quality, p.max_iterations = 1, 1, 40
        icp.align_resident(np.eye(4), p)
        t0 = time.perf_counter()
        r = icp.align_resident(np.eye(4), p)
        dt = time.perf_counter() - t0
        icp.set_profiling(True)   # the matcher's own duration: HIP events around every launch (a separate, slower run)
        r = icp.align_resident(np.eye(4), p)
        icp.set_profiling(False)
        emit(what="resident_iteration", pipeline=name, n=n, m=m, us_per_iteration=dt / 40 * 1e6,
             kernel_us=r.ms_nn_kernel / max(1, r.n_nn_launches) * 1e3)
        q = base.copy()
        icp.align(g, l, np.eye(4), q)
        t0 = time.perf_counter()
        r = icp.align(g, l, np.eye(4), q)
        dt = time.perf_counter() - t0
        emit(what="align_e2e", pipeline=name, n=n, m=m, ms=dt * 1e3, ms_upload_prepare=r.ms_upload,
             ms_iterations=r.ms_iterations, ms_quality=r.ms_quality, iterations=r.nIterations,
             termination=r.termination_name, quality=r.quality)

# loop-closure Monte-Carlo (src/LidarOdometry.cpp:767-788): K perturbed guesses on one 100k x 100k pair
g, l, Tgt = synth.make_pair(100_000, 100_000, seed=42)
rng = np.random.default_rng(7)
guesses = []
for _ in range(a.guesses):
    d = rng.normal(0, 1, 4) * np.array([0.3, 0.3, 0.3, np.deg2rad(2.0)])
    guesses.append(synth.pose_from_xyzypr(d[0], d[1], d[2], d[3], 0, 0))
for name, base in (("p2p", p2p), ("p2pl", p2pl)):
    p = base.copy()
    icp.align_multi_init(g, l, guesses[:2], p)
    t0 = time.perf_counter()
    res, best = icp.align_multi_init(g, l, guesses, p)
    dt = time.perf_counter() - t0
    t0 = time.perf_counter()
    singles = [icp.align(g, l, T0, p) for T0 in guesses]
    dt1 = time.perf_counter() - t0
    same = all(np.array_equal(x.optimal_tf, y.optimal_tf) and x.nIterations == y.nIterations for x, y in zip(res, singles))
    emit(what="multi_init", pipeline=name, guesses=a.guesses, ms=dt * 1e3, ms_standalone_aligns=dt1 * 1e3,
         iterations=[r.nIterations for r in res], best=best, bit_equal_to_standalone=same)

# config[3]'s shape on one GPU
pairs = [synth.make_pair(100_000, 100_000, seed=100 + s)[:2] for s in range(a.pairs)]
p = p2p.copy()
p.max_iterations = 100
icp.align(pairs[0][0], pairs[0][1], np.eye(4), p)
t0 = time.perf_counter()
res = icp.align_batch(pairs, [np.eye(4)] * len(pairs), p)
dt = time.perf_counter() - t0
its = sum(r.nIterations for r in res)
emit(what="align_batch", pairs=len(pairs), ms=dt * 1e3, pairs_per_s=len(pairs) / dt, iterations_total=its,
     iterations_per_s=its / dt)
t0 = time.perf_counter()
res1 = [icp.align(gg, ll, np.eye(4), p) for gg, ll in pairs]
dt1 = time.perf_counter() - t0
emit(what="align_sequential", pairs=len(pairs), ms=dt1 * 1e3, pairs_per_s=len(pairs) / dt1,
     bit_equal_to_batch=all(np.array_equal(x.optimal_tf, y.optimal_tf) for x, y in zip(res, res1)))
