#!/usr/bin/env python3
"""k_nn_tiled's quad flavour on / off (MOLA_ICP_QUADS read per launch through reload): ms per point-to-point iteration for a few (queries, map) sizes"""
import importlib, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pkg = importlib.import_module("mola-fe-lidar_amd")
synth = importlib.import_module("mola-fe-lidar_amd.synth")
lib = importlib.import_module("mola-fe-lidar_amd._lib")
p = pkg.Parameters(); p.matcher_threshold, p.fixed_iterations, p.skip_quality, p.max_iterations = 1.0, 1, 1, 20
for (n, m) in ((1_000_000, 1_000_000), (1_000_000, 3_000_000), (500_000, 5_000_000), (250_000, 2_500_000), (1_000_000, 10_000_000)):
    g, l, _ = synth.make_pair(n, m, seed=42)
    row = []
    for q in ("0", "1", "0", "1"):
        os.environ["MOLA_ICP_QUADS"] = q
        lib.lib().mola_icp_debug_reload_env()
        icp = pkg.ICP(device=0); icp.set_map(g); icp.set_local(l)
        icp.align_resident(np.eye(4), p); icp.align_resident(np.eye(4), p)
        ts = []
        for _ in range(3):
            t0 = time.perf_counter(); icp.align_resident(np.eye(4), p); ts.append((time.perf_counter() - t0) / 20 * 1e3)
        row.append("%s %.4f" % ("quads" if q == "1" else "plain", float(np.median(ts))))
        icp.close()
    print("%8d queries x %9d map points: ms per iteration  %s" % (n, m, "   ".join(row)), flush=True)
