#!/usr/bin/env python3
"""Does it pay a caller to hand over scans in PINNED host memory?  The odometry stream with pageable numpy arrays (bench.py's leg) and
with the same scans in page-locked memory (torch's pin_memory; the library's hipMemcpyAsync is then a DMA, not a staged copy)."""
import importlib, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
pkg = importlib.import_module("mola-fe-lidar_amd")
synth = importlib.import_module("mola-fe-lidar_amd.synth")
lp = pkg.LidarOdometryParams.load_from_file(os.path.join(ROOT, "params", "kitti-default.yaml"), ROOT)
scans, pinned, keep = [], [], []
for k in range(24):
    pose = synth.pose_from_xyzypr(-14.0 + 1.0 * k, 0.3 * np.sin(0.3 * k), 0.0, 0.005 * k, 0, 0)
    pc = synth.lidar_scan(pose, seed=50 + k)
    scans.append(pc)
    t = torch.from_numpy(pc).pin_memory()
    keep.append(t)
    pinned.append(t.numpy())
icp = pkg.ICP(device=0)
lo = pkg.LidarOdometry(lp, icp=icp)
for name, src in (("pageable", scans), ("pinned", pinned), ("pageable again", scans), ("pinned again", pinned)):
    for rep in range(2):
        lo.reset()
        ms, up = [], []
        for k, pc in enumerate(src):
            st = lo.on_new_observation(100.0 + 0.1 * k + 1000.0 * rep, pc)
            ms.append(st.ms_native)
            if st.icp is not None:
                up.append(st.icp.ms_upload)
    print("%-16s median %.3f ms per scan (C call), upload part %.3f ms" % (name, float(np.median(ms[2:])), float(np.median(up[1:]))), flush=True)
