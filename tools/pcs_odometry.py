#!/usr/bin/env python3
"""target of the PC-sampling runs: the odometry stream's align (kitti-default.yaml: Point2Plane knn 6 + Gauss-Newton) on the synthetic
drive, scans delivered back to back, many passes -- so that nearly all samples fall into the matcher launches of an odometry step"""
import importlib, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pkg = importlib.import_module("mola-fe-lidar_amd")
synth = importlib.import_module("mola-fe-lidar_amd.synth")
lp = pkg.LidarOdometryParams.load_from_file(os.path.join(ROOT, "params", "kitti-default.yaml"), ROOT)
scans = []
for k in range(6):    # (the first six scans of the drive: none of them runs into the iteration cap)
    pose = synth.pose_from_xyzypr(-14.0 + 1.0 * k, 0.3 * np.sin(0.3 * k), 0.0, 0.005 * k, 0, 0)
    scans.append(synth.lidar_scan(pose, seed=50 + k))
icp = pkg.ICP(device=0)
lo = pkg.LidarOdometry(lp, icp=icp)
for rep in range(int(sys.argv[1]) if len(sys.argv) > 1 else 40):
    lo.reset()
    for k, pc in enumerate(scans):
        lo.on_new_observation(100.0 + 0.1 * k + 1000.0 * rep, pc)
print("done")
