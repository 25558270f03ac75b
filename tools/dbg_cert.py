#!/usr/bin/env python3
"""per-launch certificate statistics of the plane matcher over the odometry stream's last scans (MOLA_ICP_DEBUG_STATS=3)"""
import importlib, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("MOLA_ICP_DEBUG_STATS", "3")
pkg = importlib.import_module("mola-fe-lidar_amd")
synth = importlib.import_module("mola-fe-lidar_amd.synth")
lp = pkg.LidarOdometryParams.load_from_file(os.path.join(ROOT, "params", "kitti-default.yaml"), ROOT)
icp = pkg.ICP(device=0)
lo = pkg.LidarOdometry(lp, icp=icp)
for k in range(int(sys.argv[1]) if len(sys.argv) > 1 else 16):   # (scan 6 of this drive is the pair that runs into the 100-iteration cap)
    pose = synth.pose_from_xyzypr(-14.0 + 1.0 * k, 0.3 * np.sin(0.3 * k), 0.0, 0.005 * k, 0, 0)
    print("== scan", k, file=sys.stderr, flush=True)
    lo.on_new_observation(100.0 + 0.1 * k, synth.lidar_scan(pose, seed=50 + k))
