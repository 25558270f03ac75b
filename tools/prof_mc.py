#!/usr/bin/env python3
"""the loop-closure Monte-Carlo (K guesses on one 100k x 100k pair) through the shipped point-to-plane settings -- the target of
tools/rocprof_mc.sh.  --pipeline p2p runs the point-to-point batch instead."""
import argparse, importlib, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
ap = argparse.ArgumentParser()
ap.add_argument("--n", type=int, default=100_000)
ap.add_argument("--guesses", type=int, default=10)
ap.add_argument("--reps", type=int, default=3)
ap.add_argument("--pipeline", default="p2pl")
a = ap.parse_args()
pkg = importlib.import_module("mola-fe-lidar_amd"); synth = importlib.import_module("mola-fe-lidar_amd.synth")
g, l, _ = synth.make_pair(a.n, a.n, seed=42)
if a.pipeline == "p2pl":
    p = pkg.Parameters.load_from_file(os.path.join(ROOT, "params", "icp-settings-loop-closure.yaml"))
else:
    p = pkg.Parameters(); p.matcher_threshold, p.max_iterations, p.min_abs_step_trans, p.min_abs_step_rot = 1.0, 100, 5e-5, 1e-5
rng = np.random.default_rng(7)
guesses = []
for _ in range(a.guesses):
    d = rng.normal(0, 1, 4) * np.array([0.3, 0.3, 0.3, np.deg2rad(2.0)])
    guesses.append(synth.pose_from_xyzypr(d[0], d[1], d[2], d[3], 0, 0))
icp = pkg.ICP(device=0)
icp.align_multi_init(g, l, guesses, p)
for _ in range(a.reps):
    t0 = time.perf_counter()
    res, best = icp.align_multi_init(g, l, guesses, p)
    print("multi_init %.3f ms, iterations %s, best %d" % ((time.perf_counter() - t0) * 1e3, [r.nIterations for r in res], best))
