#!/usr/bin/env python3
"""The shipped Point2Plane + Gauss-Newton pipeline alone on a seeded N x M pair: ms per iteration."""
import argparse
import importlib
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ap = argparse.ArgumentParser()
ap.add_argument("--n", type=int, default=1_000_000)
ap.add_argument("--iters", type=int, default=10)
a = ap.parse_args()
pkg = importlib.import_module("mola-fe-lidar_amd")
synth = importlib.import_module("mola-fe-lidar_amd.synth")
g, l, _ = synth.make_pair(a.n, a.n, seed=42)
p = pkg.Parameters.load_from_file(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "params",
                                               "icp-settings-regular.yaml"))
icp = pkg.ICP(device=0)
icp.set_profiling(True)  # kernel times / executed pairs are printed below
icp.set_map(g)
icp.set_local(l)
for its in (1, a.iters):
    p.fixed_iterations, p.skip_quality, p.max_iterations = 1, 1, its
    t0 = time.perf_counter()
    r = icp.align_resident(np.eye(4), p)
    dt = time.perf_counter() - t0
    print(f"n={a.n}: {its} iterations in {dt*1e3:.2f} ms = {dt*1e3/its:.3f} ms/iteration; kernel {r.ms_nn_kernel:.3f} ms total, pairs evaluated {r.nn_pairs_evaluated:#x}", flush=True)
