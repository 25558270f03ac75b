#!/usr/bin/env python3
"""headline step only (1M x 1M point-to-point, 40 fixed iterations), five timed repetitions: for same-lease library A/Bs"""
import importlib, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
pkg = importlib.import_module("mola-fe-lidar_amd")
synth = importlib.import_module("mola-fe-lidar_amd.synth")
g, l, _ = synth.make_pair(1_000_000, 1_000_000, seed=42)
tg, tl = torch.from_numpy(g).cuda(), torch.from_numpy(np.ascontiguousarray(l)).cuda()
icp = pkg.ICP(device=0)
icp.set_map(tg); icp.set_local(tl)
p = pkg.Parameters()
p.matcher_threshold, p.fixed_iterations, p.skip_quality, p.max_iterations = 1.0, 1, 1, 40
for _ in range(60):
    icp.align_resident(np.eye(4), p)
ts = []
for _ in range(9):
    t0 = time.perf_counter(); icp.align_resident(np.eye(4), p); ts.append((time.perf_counter() - t0) / 40 * 1e3)
icp.set_profiling(True)
r = icp.align_resident(np.eye(4), p)
print("C3 step: median %.4f ms (min %.4f), matcher %.4f ms per launch" % (float(np.median(ts)), min(ts), r.ms_nn_kernel / r.n_nn_launches), flush=True)
