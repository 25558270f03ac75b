#!/usr/bin/env python3
"""The numbers a kernel change is judged by, for same-lease library A/Bs (tools/gpu_lib_ab_script.sh): the headline step (1M x 1M),
the shipped pipeline at 1M, configs[4] on one GPU and the slowest equal-count shard of eight, the odometry stream and config 0."""
import importlib, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
pkg = importlib.import_module("mola-fe-lidar_amd")
synth = importlib.import_module("mola-fe-lidar_amd.synth")
sharded = importlib.import_module("mola-fe-lidar_amd.sharded")


def step_ms(icp, p, iters, reps=3):
    icp.align_resident(np.eye(4), p)
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter()
        icp.align_resident(np.eye(4), p)
        ts.append((time.perf_counter() - t0) / iters * 1e3)
    return float(np.median(ts))


g, l, _ = synth.make_pair(1_000_000, 1_000_000, seed=42)
tg, tl = torch.from_numpy(g).cuda(), torch.from_numpy(np.ascontiguousarray(l)).cuda()
icp = pkg.ICP(device=0)
icp.set_map(tg); icp.set_local(tl)
p = pkg.Parameters()
p.matcher_threshold, p.fixed_iterations, p.skip_quality, p.max_iterations = 1.0, 1, 1, 40
for _ in range(30):
    icp.align_resident(np.eye(4), p)
print("C3 headline step: %.4f ms" % step_ms(icp, p, 40, 5), flush=True)
ps = pkg.Parameters.load_from_file(os.path.join(ROOT, "params", "icp-settings-regular.yaml"))
ps.fixed_iterations, ps.skip_quality, ps.max_iterations = 1, 1, 20
ts = []
for _ in range(3):
    icp.forget_warm_start()
    t0 = time.perf_counter(); icp.align_resident(np.eye(4), ps); ts.append((time.perf_counter() - t0) / 20 * 1e3)
print("C3 shipped pipeline, first align: %.4f ms per iteration" % float(np.median(ts)), flush=True)
g5, _, _ = synth.make_pair(10, 10_000_000, seed=42)
tg5 = torch.from_numpy(g5).cuda()
del g5
p.max_iterations = 20
icp.set_map(tg5); icp.set_local(tl)
print("C5 one GPU step: %.4f ms" % step_ms(icp, p, 20), flush=True)
worst = []
for rank in (1, 5):
    icp.set_local_shard(tl, rank, 8)
    lo0, hi0 = icp.shard_reach_box(np.eye(4), 0.0)
    m = sharded.slab_margin_for_guess(lo0, hi0, 1.0, 1.0, np.deg2rad(3.0))
    lo, hi = icp.shard_reach_box(np.eye(4), m)
    icp.set_map_slab(tg5, lo, hi)
    worst.append(step_ms(icp, p, 20))
print("C5 equal-count shards 1 and 5 of 8 (the two that hold a 100-m item): %.4f / %.4f ms" % tuple(worst), flush=True)
icp.close()
del tg5
import bench
g1, l1, _ = synth.make_pair(100_000, 100_000, seed=42)
icp = pkg.ICP(device=0)
icp.set_map(g1); icp.set_local(l1)
p.max_iterations = 40
print("100k x 100k point-to-point step: %.4f ms" % step_ms(icp, p, 40, 5), flush=True)
mc = bench.montecarlo_leg(pkg, synth, icp)
print("loop-closure Monte-Carlo, 10 guesses at 100k: point-to-point %.2f ms, shipped YAML %.2f ms" % (mc["point_to_point"]["ms"], mc["shipped_loop_closure_yaml"]["ms"]), flush=True)
b3 = bench.config3_batch(pkg, synth, icp, 64, False, None, shipped=True)
print("64 pairs through the loop-closure YAML: %.0f pairs/s" % b3["gpu"]["pairs_per_s"], flush=True)
icp.close()
meds = [bench.odometry_stream_leg(pkg, synth)["ms_per_scan_median"] for _ in range(3)]
print("odometry_stream ms_per_scan_median: " + " ".join("%.3f" % m for m in meds), flush=True)
gg = synth.lidar_scan(synth.pose_from_xyzypr(-10.0, 0.2, 0, 0.01, 0, 0), seed=11)
ll = synth.lidar_scan(synth.pose_from_xyzypr(-9.0, 0.25, 0, 0.02, 0, 0), seed=12)
pp = pkg.Parameters.load_from_file(os.path.join(ROOT, "params", "icp-settings-regular.yaml"))
icp = pkg.ICP(device=0)
ts = []
for k in range(12):
    t0 = time.perf_counter(); res = icp.align(gg, ll, np.eye(4), pp); ts.append((time.perf_counter() - t0) * 1e3)
print("config0 align from host buffers: median %.3f ms, %d iterations" % (float(np.median(ts[2:])), res.nIterations), flush=True)
