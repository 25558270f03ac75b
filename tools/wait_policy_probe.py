#!/usr/bin/env python3
"""What each host wait policy (mola_icp_set_wait_policy: spin / yield / block) costs per ICP iteration, and how much of a host core the
waiting thread burns: a 100k x 100k point-to-point align and the shipped Point2Plane + Gauss-Newton pipeline on a ~120k-point scan pair
(what the reference's odometry thread runs: src/LidarOdometry.cpp:278-299), resident clouds, fixed iterations.  CPU share = process CPU
time / wall time over the timed aligns (1.0 = one core busy for the whole align)."""
import importlib
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pkg = importlib.import_module("mola-fe-lidar_amd")
synth = importlib.import_module("mola-fe-lidar_amd.synth")

IT = 20
p2p = pkg.Parameters()
p2p.matcher_threshold, p2p.fixed_iterations, p2p.skip_quality, p2p.max_iterations = 1.0, 1, 1, IT
shipped = pkg.Parameters.load_from_file(os.path.join(ROOT, "params", "icp-settings-regular.yaml"))
shipped.fixed_iterations, shipped.skip_quality, shipped.max_iterations = 1, 1, IT
g, l, _ = synth.make_pair(100_000, 100_000, seed=42)
a = synth.lidar_scan(synth.pose_from_xyzypr(-10.0, 0.2, 0, 0.01, 0, 0), seed=11)
b = synth.lidar_scan(synth.pose_from_xyzypr(-9.0, 0.25, 0, 0.02, 0, 0), seed=12)
cases = [("100k x 100k point-to-point", g, l, p2p), ("~120k-point scan pair, shipped Point2Plane + Gauss-Newton", a, b, shipped)]
ref = {}
for policy in ("spin", "yield", "block", "spin"):
    pkg.ICP.set_wait_policy(policy)
    for name, gg, ll, p in cases:
        icp = pkg.ICP(device=0)
        icp.set_map(gg); icp.set_local(ll)
        for _ in range(5):
            icp.align_resident(np.eye(4), p)
        reps = 40
        c0, t0 = time.process_time(), time.perf_counter()
        for _ in range(reps):
            icp.forget_warm_start()
            r = icp.align_resident(np.eye(4), p)
        c1, t1 = time.process_time(), time.perf_counter()
        key = name
        same = key not in ref or np.array_equal(ref[key], r.optimal_tf)
        ref.setdefault(key, np.array(r.optimal_tf))
        print("%-6s %-58s %7.1f us per iteration   CPU share of the calling thread %.2f%s" %
              (policy, name, (t1 - t0) / (reps * IT) * 1e6, (c1 - c0) / (t1 - t0), "" if same else "   RESULT DIFFERS"), flush=True)
        icp.close()
pkg.ICP.set_wait_policy("spin")
