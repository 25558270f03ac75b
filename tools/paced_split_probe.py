#!/usr/bin/env python3
"""Where inside the C call does a scan delivered at 10 Hz cost more than one delivered back to back?  The front-end mirror over the
synthetic drive, four ways -- back to back / every 100 ms after a sleep / every 100 ms with the caller spinning through the last
2 ms before the scan is due (its core stays awake) / the same from page-locked scan buffers -- with the library's own split of each
call: upload + prepare, iterations, quality."""
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


def run(src, period, spin_ms, passes=3):
    nat, up, it, q, n_it = [], [], [], [], []
    for rep in range(1 + passes):
        lo.reset()
        t_next = time.perf_counter()
        for k, pc in enumerate(src):
            if period and rep >= 1:
                t_next += period
                slack = t_next - time.perf_counter() - spin_ms * 1e-3
                if slack > 0:
                    time.sleep(slack)
                while time.perf_counter() < t_next:
                    pass
            st = lo.on_new_observation(100.0 + 0.1 * k + 1000.0 * rep, pc)
            if rep >= 1 and k >= 2 and st.icp is not None and st.icp.nIterations <= 6:
                nat.append(st.ms_native); up.append(st.icp.ms_upload); it.append(st.icp.ms_iterations); q.append(st.icp.ms_quality); n_it.append(st.icp.nIterations)
    m = lambda v: float(np.median(v))
    return m(nat), m(up), m(it), m(q), m(n_it)


for name, src, period, spin in (("back to back", scans, None, 0), ("10 Hz, sleep", scans, 0.1, 0), ("10 Hz, spin the last 2 ms", scans, 0.1, 2.0),
                                ("10 Hz, sleep, pinned scans", pinned, 0.1, 0), ("back to back, pinned scans", pinned, None, 0), ("10 Hz, sleep (again)", scans, 0.1, 0),
                                ("back to back (again)", scans, None, 0)):
    r = run(src, period, spin)
    print("%-30s C call %.3f ms = upload+prepare %.3f + iterations %.3f (%.0f its) + quality %.3f + rest %.3f" % (name, r[0], r[1], r[2], r[4], r[3], r[0] - r[1] - r[2] - r[3]), flush=True)
