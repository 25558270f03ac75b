#!/usr/bin/env python3
"""Step time of ONE rank's shard of the 1M x 1M job, alone on the GPU (no collective): what a rank of an N-GPU
run spends per iteration before the all-reduce.  Compares Z-order shards with index-order (random) shards."""
import importlib
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pkg = importlib.import_module("mola-fe-lidar_amd")
synth = importlib.import_module("mola-fe-lidar_amd.synth")
sharded = importlib.import_module("mola-fe-lidar_amd.sharded")
g, l, _ = synth.make_pair(1_000_000, 1_000_000, seed=42)
order = sharded.spatial_order(l)
p = pkg.Parameters()
p.matcher_threshold, p.fixed_iterations, p.skip_quality, p.max_iterations = 1.0, 1, 1, 20
icp = pkg.ICP(device=0)
icp.set_profiling(True)  # kernel times / executed pairs are printed below
icp.set_map(g)
for world in (1, 2, 4, 8):
    lo, hi = sharded.shard_bounds(l.shape[1], 0, world)
    for name, sh in (("z-order", l[:, order[lo:hi]]), ("index-order", l[:, lo:hi])):
        icp.set_local(np.ascontiguousarray(sh))
        icp.set_global_sizes(l.shape[1], g.shape[1])
        icp.align_resident(np.eye(4), p)
        t0 = time.perf_counter()
        r = icp.align_resident(np.eye(4), p)
        dt = (time.perf_counter() - t0) / 20
        print(f"world {world} rank 0 {name:12s}: {hi-lo} queries, {dt*1e3:.3f} ms/iteration, matcher {r.ms_nn_kernel/r.n_nn_launches:.3f} ms/launch", flush=True)
