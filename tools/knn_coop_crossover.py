#!/usr/bin/env python3
"""Shipped point-to-plane pipeline (8 fixed iterations, resident) over cloud sizes, cooperative vs persistent plane matcher
(MOLA_ICP_KNN_COOP set by the caller): ms per align."""
import importlib
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pkg = importlib.import_module("mola-fe-lidar_amd")
synth = importlib.import_module("mola-fe-lidar_amd.synth")
icp = pkg.ICP(device=0)
p = pkg.Parameters.load_from_file(os.path.join(ROOT, "params", "icp-settings-regular.yaml"))
p.fixed_iterations, p.skip_quality, p.max_iterations = 1, 1, 8
out = []
for n in (60_000, 120_000, 160_000, 200_000, 260_000, 330_000):
    g, l, _ = synth.make_pair(n, n, seed=42)
    icp.set_map(g); icp.set_local(l)
    for _ in range(3):
        icp.align_resident(np.eye(4), p)
    t0 = time.perf_counter()
    for _ in range(5):
        icp.align_resident(np.eye(4), p)
    out.append("%dk:%.3f" % (n // 1000, (time.perf_counter() - t0) / 5 * 1e3))
print("KNN_COOP=%s ms/align " % os.environ.get("MOLA_ICP_KNN_COOP", "auto"), " ".join(out))
