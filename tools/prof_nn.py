#!/usr/bin/env python3
"""Runs only the NN matcher (the dominant kernel) a few times on a seeded N x M pair -- the target
for `rocprofv3 --kernel-trace --stats` / `--pmc` runs.  No torch: nothing else launches kernels."""
import argparse
import importlib
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ap = argparse.ArgumentParser()
ap.add_argument("--n", type=int, default=1_000_000)
ap.add_argument("--m", type=int, default=1_000_000)
ap.add_argument("--reps", type=int, default=5)
ap.add_argument("--kernel", choices=["valu", "mfma", "tiled"], default="mfma")
ap.add_argument("--gate", type=float, default=1.0)
a = ap.parse_args()
pkg = importlib.import_module("mola-fe-lidar_amd")
synth = importlib.import_module("mola-fe-lidar_amd.synth")
g, l, _ = synth.make_pair(a.n, a.m, seed=42)
icp = pkg.ICP(device=0)
icp.set_profiling(True)  # kernel times / executed pairs are printed below
icp.set_map(g)
icp.set_local(l)
k = {"mfma": pkg.NN_MFMA, "valu": pkg.NN_VALU, "tiled": pkg.NN_TILED}[a.kernel]
p = pkg.Parameters()
p.max_iterations, p.matcher_threshold, p.fixed_iterations, p.skip_quality, p.nn_kernel = 1, a.gate, 1, 1, k
t0 = time.perf_counter()
icp.match(np.eye(4), a.gate, a.n, k, copy=False)  # first call: map preparation + un-seeded match
icp.align_resident(np.eye(4), p)  # flushes the MOLA_ICP_DEBUG_STATS counters of the un-seeded launch
print(f"{a.kernel}: first match incl. map preparation {1e3*(time.perf_counter()-t0):.2f} ms")
t0 = time.perf_counter()
for _ in range(a.reps):
    _, _, n = icp.match(np.eye(4), a.gate, a.n, k, copy=False)
dt = (time.perf_counter() - t0) / a.reps
print(f"{a.kernel}: {dt*1e3:.2f} ms per match (host wall, incl. launch+sync), pairs={n}, "
      f"{8.0*a.n*a.m/dt/1e12:.1f} TFLOP/s algorithmic")
icp.align_resident(np.eye(4), p)  # prints MOLA_ICP_DEBUG_STATS counters, if enabled
