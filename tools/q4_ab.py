#!/usr/bin/env python3
"""k_nn_q4 (four lanes per query) against the matcher it replaces (MOLA_ICP_Q4 read per launch through reload): us per point-to-point
iteration and per matcher launch (HIP events) for a few (queries, map) sizes; the pairing of the last launch is compared bit for bit."""
import importlib, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pkg = importlib.import_module("mola-fe-lidar_amd")
synth = importlib.import_module("mola-fe-lidar_amd.synth")
lib = importlib.import_module("mola-fe-lidar_amd._lib")
IT = 20
p = pkg.Parameters(); p.matcher_threshold, p.fixed_iterations, p.skip_quality, p.max_iterations = 1.0, 1, 1, IT
sizes = [(50_000, 50_000), (100_000, 100_000), (120_000, 120_000), (125_000, 1_000_000), (200_000, 200_000), (260_000, 260_000), (390_000, 390_000)]
if len(sys.argv) > 1:
    sizes = [tuple(int(v) for v in a.split("x")) for a in sys.argv[1:]]
for (n, m) in sizes:
    g, l, _ = synth.make_pair(n, m, seed=42)
    row, ref = [], None
    for q in ("0", "1", "0", "1"):
        if os.environ.get("Q4_AB_MODE") == "fuse":   # (with tools/experiments/q4_fused_row_reduction.patch applied: k_nn_q4's rows reduced by k_reduce_items ("0") / inside the launch ("1"))
            os.environ.pop("MOLA_ICP_Q4_NO_FUSE", None)
            if q == "0":
                os.environ["MOLA_ICP_Q4_NO_FUSE"] = "1"
        else:
            os.environ["MOLA_ICP_Q4"] = q
        lib.lib().mola_icp_debug_reload_env()
        icp = pkg.ICP(device=0); icp.set_map(g); icp.set_local(l)
        icp.align_resident(np.eye(4), p); icp.align_resident(np.eye(4), p)
        ts = []
        for _ in range(5):
            icp.forget_warm_start()
            t0 = time.perf_counter(); r = icp.align_resident(np.eye(4), p); ts.append((time.perf_counter() - t0) / IT * 1e6)
        icp.set_profiling(True)
        icp.forget_warm_start()
        r = icp.align_resident(np.eye(4), p)
        icp.set_profiling(False)
        k_us = r.ms_nn_kernel / max(1, r.n_nn_launches) * 1e3
        idx, d2, npairs = icp.match(r.optimal_tf, 1.0, n, pkg.NN_TILED)
        if ref is None:
            ref = (idx.copy(), d2.copy(), np.array(r.optimal_tf))
        same = np.array_equal(idx, ref[0]) and np.array_equal(d2, ref[1]) and np.array_equal(np.array(r.optimal_tf), ref[2])
        row.append("%s %.1f (launch %.1f)%s" % (("q4" if q == "1" else "base") if os.environ.get("Q4_AB_MODE") != "fuse" else ("fused" if q == "1" else "2 launches"), float(np.median(ts)), k_us, "" if same else " MISMATCH"))
        icp.close()
    print("%8d queries x %9d map points: us per iteration  %s" % (n, m, "   ".join(row)), flush=True)
