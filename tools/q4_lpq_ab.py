#!/usr/bin/env python3
"""k_nn_q4 at four and at two lanes per query against k_nn_tiled (MOLA_ICP_Q4 / MOLA_ICP_Q4_LPQ read per launch through reload): us per
point-to-point iteration and per matcher launch for a few (queries, map) sizes; the pairing of the last launch is compared bit for bit.
Needs tools/experiments/nn_q4_two_lanes_per_query.patch applied (the product's k_nn_q4 has four lanes per query only: the measurement is
profiles/r06/dropped_nn_q4_two_lanes.txt)."""
import importlib, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pkg = importlib.import_module("mola-fe-lidar_amd")
synth = importlib.import_module("mola-fe-lidar_amd.synth")
lib = importlib.import_module("mola-fe-lidar_amd._lib")
IT = 20
p = pkg.Parameters(); p.matcher_threshold, p.fixed_iterations, p.skip_quality, p.max_iterations = 1.0, 1, 1, IT
sizes = [tuple(int(v) for v in a.split("x")) for a in sys.argv[1:]] or [(125_000, 1_000_000), (200_000, 200_000), (400_000, 400_000), (1_000_000, 1_000_000)]
for (n, m) in sizes:
    g, l, _ = synth.make_pair(n, m, seed=42)
    row, ref = [], None
    for name, env in (("tiled", {"MOLA_ICP_Q4": "0"}), ("q4 x4", {"MOLA_ICP_Q4": "1", "MOLA_ICP_Q4_LPQ": "4"}), ("q4 x2", {"MOLA_ICP_Q4": "1", "MOLA_ICP_Q4_LPQ": "2"})) * 2:
        for k in ("MOLA_ICP_Q4", "MOLA_ICP_Q4_LPQ"):
            os.environ.pop(k, None)
        os.environ.update(env)
        lib.lib().mola_icp_debug_reload_env()
        icp = pkg.ICP(device=0); icp.set_map(g); icp.set_local(l)
        icp.align_resident(np.eye(4), p); icp.align_resident(np.eye(4), p)
        ts = []
        for _ in range(5):
            icp.forget_warm_start()
            t0 = time.perf_counter(); r = icp.align_resident(np.eye(4), p); ts.append((time.perf_counter() - t0) / IT * 1e6)
        icp.set_profiling(True); icp.forget_warm_start(); r = icp.align_resident(np.eye(4), p); icp.set_profiling(False)
        k_us = r.ms_nn_kernel / max(1, r.n_nn_launches) * 1e3
        idx, d2, npairs = icp.match(r.optimal_tf, 1.0, n, pkg.NN_TILED)
        if ref is None:
            ref = (idx.copy(), d2.copy(), np.array(r.optimal_tf))
        same = np.array_equal(idx, ref[0]) and np.array_equal(d2, ref[1]) and np.array_equal(np.array(r.optimal_tf), ref[2])
        row.append("%s %.1f (%.1f)%s" % (name, float(np.median(ts)), k_us, "" if same else " MISMATCH"))
        icp.close()
    print("%8d x %8d: us per iteration (launch)  %s" % (n, m, "   ".join(row)), flush=True)
