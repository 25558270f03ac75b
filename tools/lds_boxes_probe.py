"""The upper box levels in LDS or not (MOLA_ICP_LDS_BOXES_KB): ms per iteration of the plane / NN matchers for scans against maps of 1M .. 3M points"""
import importlib, os, sys, time
import numpy as np
sys.path.insert(0, os.getcwd())
pkg = importlib.import_module("mola-fe-lidar_amd"); synth = importlib.import_module("mola-fe-lidar_amd.synth"); lib = importlib.import_module("mola-fe-lidar_amd._lib")
ROOT = os.getcwd()
pp = pkg.Parameters.load_from_file(os.path.join(ROOT, "params", "icp-settings-regular.yaml")); pp.fixed_iterations, pp.skip_quality, pp.max_iterations = 1, 1, 12
pn = pkg.Parameters(); pn.matcher_threshold, pn.fixed_iterations, pn.skip_quality, pn.max_iterations = 1.0, 1, 1, 20
for (n, m) in ((120_000, 1_000_000), (120_000, 2_000_000), (120_000, 3_000_000), (1_000_000, 2_000_000), (1_000_000, 3_000_000)):
    g, l, _ = synth.make_pair(n, m, seed=42)
    for name, p in (("planes", pp), ("nn", pn)):
        if name == "nn" and n > 200_000: continue
        row = []
        for kb in ("40", "0", "40", "0"):   # (MOLA_ICP_LDS_BOXES_KB caps the limit: 40 = the occupancy-aware default, 0 = never in LDS)
            os.environ["MOLA_ICP_LDS_BOXES_KB"] = kb
            lib.lib().mola_icp_debug_reload_env()
            icp = pkg.ICP(device=0); icp.set_map(g); icp.set_local(l)
            icp.align_resident(np.eye(4), p); icp.align_resident(np.eye(4), p)
            ts = []
            for _ in range(3):
                t0 = time.perf_counter(); icp.align_resident(np.eye(4), p); ts.append((time.perf_counter() - t0) / p.max_iterations * 1e3)
            row.append("%s KB %.4f" % (kb, float(np.median(ts)))); icp.close()
        print("%8d x %9d %-6s: %s" % (n, m, name, "   ".join(row)), flush=True)
