#!/usr/bin/env python3
"""ICP iteration time (resident, point-to-point) over cloud sizes, cooperative vs persistent matcher (MOLA_ICP_COOP set by the caller)."""
import importlib
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pkg = importlib.import_module("mola-fe-lidar_amd")
synth = importlib.import_module("mola-fe-lidar_amd.synth")
icp = pkg.ICP(device=0)
p = pkg.Parameters()
p.matcher_threshold, p.fixed_iterations, p.skip_quality, p.max_iterations = 1.0, 1, 1, 40
out = []
for n in (50_000, 100_000, 150_000, 200_000, 260_000, 330_000, 390_000, 500_000):
    g, l, _ = synth.make_pair(n, n, seed=42)
    icp.set_map(g); icp.set_local(l)
    icp.align_resident(np.eye(4), p)
    t0 = time.perf_counter()
    icp.align_resident(np.eye(4), p)
    out.append("%dk:%.1f" % (n // 1000, (time.perf_counter() - t0) / 40 * 1e6))
print("COOP=%s us/iteration " % os.environ.get("MOLA_ICP_COOP", "auto"), " ".join(out))
