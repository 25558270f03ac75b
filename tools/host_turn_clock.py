import importlib, os, sys, time
import numpy as np
sys.path.insert(0, os.getcwd())
pkg = importlib.import_module("mola-fe-lidar_amd"); synth = importlib.import_module("mola-fe-lidar_amd.synth")
g, l, _ = synth.make_pair(1000000, 1000000, seed=42)
icp = pkg.ICP(device=0); icp.set_map(g); icp.set_local(l)
p = pkg.Parameters(); p.max_iterations, p.matcher_threshold, p.fixed_iterations, p.skip_quality = 40, 1.0, 1, 1
for k in range(12):
    t0 = time.perf_counter(); r = icp.align_resident(np.eye(4), p); dt = time.perf_counter() - t0
print("C3: %.4f ms per iteration (last align)" % (dt * 1e3 / 40))
