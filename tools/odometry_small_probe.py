#!/usr/bin/env python3
"""Which decimation of the synthetic 64-ring scan makes a 10k-30k point odometry leg that CONVERGES like the full scan does?
(16 rings x 1250 azimuths ran into the 100-iteration cap on most scans: rings further apart than the plane matcher's gate leave
collinear neighbourhoods, whose 'planes' are noise.)  Prints iterations and milliseconds per scan for a few candidates."""
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
icp = pkg.ICP(device=0)
for name, kw, voxel in (("64x1875", dict(n_rings=64, n_az=1875), 0), ("64x313", dict(n_rings=64, n_az=313), 0), ("32x625", dict(n_rings=32, n_az=625), 0),
                        ("64x1875 voxel 0.10", dict(n_rings=64, n_az=1875), 0.10), ("64x1875 voxel 0.15", dict(n_rings=64, n_az=1875), 0.15),
                        ("16x1250", dict(n_rings=16, n_az=1250), 0)):
    scans = []
    for k in range(24):
        pose = synth.pose_from_xyzypr(-14.0 + 1.0 * k, 0.3 * np.sin(0.3 * k), 0.0, 0.005 * k, 0, 0)
        pc = synth.lidar_scan(pose, seed=50 + k, **kw)
        if voxel:
            pc = icp.voxel_downsample(pc, voxel)
        scans.append((100.0 + 0.1 * k, np.ascontiguousarray(pc)))
    lo = pkg.LidarOdometry(lp, icp=icp)
    for rep in range(2):
        lo.reset()
        ms, its = [], []
        for k, (t, pc) in enumerate(scans):
            t0 = time.perf_counter()
            st = lo.on_new_observation(t + 1000.0 * rep, pc)
            ms.append((time.perf_counter() - t0) * 1e3)
            if st.icp is not None:
                its.append(int(st.icp.nIterations))
    lo.close()
    print(f"{name:22s} ~{int(np.mean([pc.shape[1] for _, pc in scans]))} points: median {np.median(ms[2:]):.3f} ms, iterations {its}", flush=True)
