import importlib, os, sys, time
import numpy as np
sys.path.insert(0, os.getcwd())
pkg = importlib.import_module("mola-fe-lidar_amd"); synth = importlib.import_module("mola-fe-lidar_amd.synth")
g, l, _ = synth.make_pair(1_000_000, 10_000_000, seed=42)
icp = pkg.ICP(device=0); icp.set_map(g); icp.set_local(l)
p = pkg.Parameters(); p.matcher_threshold, p.fixed_iterations, p.skip_quality, p.max_iterations = 1.0, 1, 1, 20
icp.align_resident(np.eye(4), p)
ts = []
for k in range(14):
    icp.forget_warm_start()
    t0 = time.perf_counter(); icp.align_resident(np.eye(4), p); ts.append((time.perf_counter() - t0) / 20 * 1e3)
print("stateless C5 aligns, ms per step:", " ".join("%.3f" % t for t in ts))
p.max_iterations = 1
for k in range(4):
    icp.forget_warm_start()
    t0 = time.perf_counter(); icp.align_resident(np.eye(4), p); print("one unseeded launch + turn: %.3f ms" % ((time.perf_counter() - t0) * 1e3))
