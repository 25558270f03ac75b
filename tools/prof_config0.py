#!/usr/bin/env python3
"""configs[0] (KITTI-like 120k pair, shipped pipeline through kitti-default.yaml): where the end-to-end time goes."""
import importlib
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pkg = importlib.import_module("mola-fe-lidar_amd")
synth = importlib.import_module("mola-fe-lidar_amd.synth")
lp = pkg.LidarOdometryParams.load_from_file(os.path.join(ROOT, "params", "kitti-default.yaml"), ROOT)
p = lp.icp_case("with_vel")
a = synth.lidar_scan(synth.pose_from_xyzypr(-10.0, 0.2, 0, 0.01, 0, 0), seed=11)
b = synth.lidar_scan(synth.pose_from_xyzypr(-9.0, 0.25, 0, 0.02, 0, 0), seed=12)
icp = pkg.ICP(device=0)
icp.align(a, b, np.eye(4), p)
for prof in (False, True):
    icp.set_profiling(prof)
    ts = []
    for _ in range(5):
        t0 = time.perf_counter()
        r = icp.align(a, b, np.eye(4), p)
        ts.append(time.perf_counter() - t0)
    print(f"profiling={prof}: align {np.median(ts)*1e3:.2f} ms; prepare {r.ms_upload:.2f}, loop {r.ms_iterations:.2f} ({r.nIterations} its), quality {r.ms_quality:.2f}; "
          f"matcher {r.ms_nn_kernel:.2f} ms over {r.n_nn_launches} launches; n={a.shape[1]}x{b.shape[1]}", flush=True)
for its in (1, 2, 3, 4, 8):
    q = p.copy()
    q.fixed_iterations, q.skip_quality, q.max_iterations = 1, 1, its
    icp.set_map(a); icp.set_local(b)
    icp.set_profiling(True)
    r = icp.align_resident(np.eye(4), q)
    print(f"  first {its} iterations: matcher {r.ms_nn_kernel*1e3:.0f} us total, loop {r.ms_iterations*1e3:.0f} us", flush=True)
