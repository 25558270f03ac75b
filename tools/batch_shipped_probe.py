import importlib, os, sys, time
import numpy as np
ROOT = "/root/repo" if os.path.isdir("/root/repo/tools") else os.getcwd()
sys.path.insert(0, ROOT)
pkg = importlib.import_module("mola-fe-lidar_amd"); synth = importlib.import_module("mola-fe-lidar_amd.synth")
pairs = [synth.make_pair(100_000, 100_000, seed=100 + s)[:2] for s in range(24)]
p = pkg.Parameters.load_from_file(os.path.join(ROOT, "params", "icp-settings-regular.yaml"))
icp = pkg.ICP(device=0)
icp.align_batch(pairs[:4], [np.eye(4)] * 4, p)
for rep in range(3):
    t0 = time.perf_counter()
    res = icp.align_batch(pairs, [np.eye(4)] * len(pairs), p)
    dt = time.perf_counter() - t0
    print("shipped batch: %d pairs in %.1f ms = %.1f pairs/s (its %d)" % (len(pairs), dt * 1e3, len(pairs) / dt, sum(r.nIterations for r in res)))
