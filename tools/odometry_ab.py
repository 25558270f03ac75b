#!/usr/bin/env python3
"""The two numbers the reference's operating point is judged by, for same-lease A/B runs (tools/gpu_lib_ab_script.sh):
bench.py's odometry_stream leg (median ms per scan over three passes) and config 0 end to end (align from host buffers)."""
import importlib, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
pkg = importlib.import_module("mola-fe-lidar_amd")
synth = importlib.import_module("mola-fe-lidar_amd.synth")
meds = []
for rep in range(3):
    r = bench.odometry_stream_leg(pkg, synth)
    meds.append(r["ms_per_scan_median"])
print("odometry_stream ms_per_scan_median: " + " ".join("%.3f" % m for m in meds) + "  (min %.3f, its/scan %.1f)" % (r["ms_per_scan_min"], r["iterations_per_scan_median"]))
# config 0: the KITTI-like 120k pair through kitti-default.yaml's ICP settings, host buffers in
g = synth.lidar_scan(synth.pose_from_xyzypr(-10.0, 0.2, 0, 0.01, 0, 0), seed=11)    # (bench.py: align_e2e's config0 pair)
l = synth.lidar_scan(synth.pose_from_xyzypr(-9.0, 0.25, 0, 0.02, 0, 0), seed=12)
p = pkg.Parameters.load_from_file(os.path.join(ROOT, "params", "icp-settings-regular.yaml"))
icp = pkg.ICP(device=0)
ts = []
for k in range(12):
    t0 = time.perf_counter()
    res = icp.align(g, l, np.eye(4), p)
    ts.append((time.perf_counter() - t0) * 1e3)
print("config0 align from host buffers: median %.3f ms (min %.3f), %d iterations, prepare %.3f ms" % (float(np.median(ts[2:])), min(ts[2:]), res.nIterations, res.ms_upload))
icp.close()
