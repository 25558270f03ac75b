#!/usr/bin/env python3
"""align_batch of 12 / 24 pairs of 100k x 100k with the batched matcher forced either way (MOLA_ICP_BATCH_TILED set by the
caller): wall time, matcher time per launch (profiled repetition), upload share."""
import importlib
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pkg = importlib.import_module("mola-fe-lidar_amd")
synth = importlib.import_module("mola-fe-lidar_amd.synth")
pairs = [synth.make_pair(100_000, 100_000, seed=100 + s)[:2] for s in range(24)]
p = pkg.Parameters()
p.max_iterations, p.matcher_threshold, p.min_abs_step_trans, p.min_abs_step_rot = 100, 1.0, 5e-5, 1e-5
icp = pkg.ICP(device=0)
for n in (12, 24):
    icp.set_profiling(False)
    icp.align_batch(pairs[:n], [np.eye(4)] * n, p)
    t0 = time.perf_counter()
    res = icp.align_batch(pairs[:n], [np.eye(4)] * n, p)
    dt = time.perf_counter() - t0
    icp.set_profiling(True)
    rp = icp.align_batch(pairs[:n], [np.eye(4)] * n, p)
    print(f"mode={os.environ.get('MOLA_ICP_BATCH_TILED', 'auto')} pairs={n}: {dt*1e3:.1f} ms = {n/dt:.0f} pairs/s; upload+prepare {res[0].ms_upload:.2f} ms per chunk, "
          f"loop {res[0].ms_iterations:.2f} ms, matcher {rp[0].ms_nn_kernel/max(1,rp[0].n_nn_launches)*1e3:.1f} us per launch over {rp[0].n_nn_launches} launches, "
          f"pairs evaluated per query {rp[0].nn_pairs_evaluated/max(1,rp[0].n_nn_launches)/(min(n,12)*100000):.0f}", flush=True)
