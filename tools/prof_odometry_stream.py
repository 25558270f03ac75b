#!/usr/bin/env python3
"""per-scan wall time of the front-end mirror over a synthetic drive (bench.py's odometry_stream leg, every pass printed)"""
import importlib, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pkg = importlib.import_module("mola-fe-lidar_amd")
synth = importlib.import_module("mola-fe-lidar_amd.synth")
lp = pkg.LidarOdometryParams.load_from_file(os.path.join(ROOT, "params", "kitti-default.yaml"), ROOT)
n_scans = int(sys.argv[1]) if len(sys.argv) > 1 else 24
decimate = int(sys.argv[2]) if len(sys.argv) > 2 else 1   # (every n-th point of each scan; 10 = the reference's full_pointcloud_decimation)
scans = []
for k in range(n_scans):
    pose = synth.pose_from_xyzypr(-14.0 + 1.0 * k, 0.3 * np.sin(0.3 * k), 0.0, 0.005 * k, 0, 0)
    scans.append((100.0 + 0.1 * k, np.ascontiguousarray(synth.lidar_scan(pose, seed=50 + k)[:, ::decimate])))
icp = pkg.ICP(device=0)
lo = pkg.LidarOdometry(lp, icp=icp)
for rep in range(4):
    lo.reset()
    row = []
    for k, (t, pc) in enumerate(scans):
        t0 = time.perf_counter()
        st = lo.on_new_observation(t + 1000.0 * rep, pc)
        dt = (time.perf_counter() - t0) * 1e3
        r = st.icp
        row.append("%.2f(%s)" % (dt, "-" if r is None else "%d its %.2f+%.2f+%.2f" % (r.nIterations, r.ms_upload, r.ms_iterations, r.ms_quality)))
    print("pass", rep, " ".join(row), flush=True)
