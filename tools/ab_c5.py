#!/usr/bin/env python3
"""configs[4] on one GPU (1M queries vs a 10M-point map, point-to-point) and one equal-count shard of eight with its map slab:
step time, for same-lease library / knob A/Bs"""
import importlib, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
pkg = importlib.import_module("mola-fe-lidar_amd")
synth = importlib.import_module("mola-fe-lidar_amd.synth")
sharded = importlib.import_module("mola-fe-lidar_amd.sharded")
g5, l, _ = synth.make_pair(1_000_000, 10_000_000, seed=42)
tg5, tl = torch.from_numpy(g5).cuda(), torch.from_numpy(np.ascontiguousarray(l)).cuda()
del g5
p = pkg.Parameters()
p.matcher_threshold, p.fixed_iterations, p.skip_quality, p.max_iterations = 1.0, 1, 1, 20
icp = pkg.ICP(device=0)


def step_ms():
    icp.align_resident(np.eye(4), p); icp.align_resident(np.eye(4), p)
    ts = []
    for _ in range(3):
        t0 = time.perf_counter(); icp.align_resident(np.eye(4), p); ts.append((time.perf_counter() - t0) / 20 * 1e3)
    return float(np.median(ts))


icp.set_map(tg5); icp.set_local(tl)
print("C5 one GPU step: %.4f ms" % step_ms(), flush=True)
icp.set_local_shard(tl, 3, 4)
lo0, hi0 = icp.shard_reach_box(np.eye(4), 0.0)
m = sharded.slab_margin_for_guess(lo0, hi0, 1.0, 1.0, np.deg2rad(3.0))
lo, hi = icp.shard_reach_box(np.eye(4), m)
icp.set_map_slab(tg5, lo, hi)
print("C5 shard 3 of 4 (250k queries, persistent kernel): %.4f ms" % step_ms(), flush=True)
