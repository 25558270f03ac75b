#!/usr/bin/env python3
"""bench.py -- ICP iterations/sec at 1M x 1M on MI355X (BASELINE.json metric, configs[2]).

A "step" = ONE ICP iteration of the hot path on HBM-resident clouds: transform the N queries,
brute-force nearest neighbour over the M map points, distance gate, fp64 centroid/covariance
accumulation, (all-reduce of the 24-double block when sharded), Horn solve, stall-test
quantities.  `--steps K` runs exactly K such iterations (stall test disabled, quality pass
skipped) between two barrier+synchronize brackets; `value` = K / max-over-ranks seconds.

N GPUs: strong scaling of the SAME 1M x 1M job -- the queries are sharded contiguously over the
ranks (map replicated), one RCCL all-reduce of the accumulator block per iteration.

    python bench.py                      # 1 GPU, 40 steps, 3 warmup
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 \
        --master-port 29500 bench.py --gpus 8 --steps 40 --warmup 3
"""
import argparse
import importlib
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

PEAK_FP32_TFLOPS = 157.3   # /opt/skills/guides/MI355X_MICROARCH.md: fp32 vector == fp32 MFMA dense peak
PEAK_HBM_GBS = 8000.0      # same guide: HBM3E 8.0 TB/s spec (6.29 TB/s measured copy)
GATE_M = 1.0               # gate 1.0 m for the point-to-point benchmark


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=40)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--n-local", type=int, default=1_000_000)
    ap.add_argument("--n-map", type=int, default=1_000_000)
    ap.add_argument("--seed", type=int, default=42)
    ap.add_argument("--nn-kernel", choices=["auto", "valu", "mfma", "tiled"], default="auto")
    ap.add_argument("--cpu-baseline-iters", type=int, default=5, help="0 disables the CPU baseline leg")
    ap.add_argument("--force-dist", action="store_true",
                    help="run the distributed code path (process group + RCCL communicator) even with one rank")
    ap.add_argument("--allreduce", choices=["rccl", "hook"], default="rccl",
                    help="rccl: native RCCL on the device block; hook: torch.distributed from the host hook")
    ap.add_argument("--share-gpu", action="store_true",
                    help="testing aid for 1-GPU boxes: ranks share the visible GPUs (rank %% device_count), the process "
                         "group is gloo and the accumulator all-reduce goes through the host hook -- exercises the "
                         "N>1 sharding path end to end without RCCL")
    ap.add_argument("--shipped-iters", type=int, default=20,
                    help="iterations of the shipped Point2Plane+GaussNewton pipeline measured beside the default path")
    ap.add_argument("--dense-iters", type=int, default=3, help="iterations of the dense MFMA kernel measured beside the default path (0 = skip)")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run "
                         f"--nproc-per-node {args.gpus}")
    if args.share_gpu:
        local_rank = local_rank % max(1, torch.cuda.device_count())
        args.allreduce = "hook"
    torch.cuda.set_device(local_rank)
    use_dist = world > 1 or args.force_dist
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        if args.share_gpu:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    pkg = importlib.import_module("mola-fe-lidar_amd")
    synth = importlib.import_module("mola-fe-lidar_amd.synth")
    sharded = importlib.import_module("mola-fe-lidar_amd.sharded")

    N, M = args.n_local, args.n_map
    g, l, T_gt = synth.make_pair(N, M, seed=args.seed)
    lo, hi = sharded.shard_bounds(N, rank, world)

    # inputs resident in HBM before the timed region
    dev = torch.device("cuda", local_rank)
    cdev = torch.device("cpu") if args.share_gpu else dev   # where the bench's own small collectives live
    tg = torch.from_numpy(g).to(dev)
    # spatially compact shards (Z-order slices of the scan): a random 1/W subsample would be W times sparser than
    # the map and every 128-query group would sweep W times more map tiles
    shard = l[:, lo:hi] if world == 1 else l[:, sharded.spatial_order(l)[lo:hi]]
    tl = torch.from_numpy(np.ascontiguousarray(shard)).to(dev)
    icp = pkg.ICP(device=local_rank)
    icp.set_map(tg)
    icp.set_local(tl)
    icp.set_global_sizes(N, M)
    allreduce_used = None
    if use_dist:
        allreduce_used = args.allreduce
        if args.allreduce == "rccl":
            try:
                icp.comm_init()
            except Exception as e:  # keep the run alive: torch.distributed carries the 24 doubles instead
                print(f"[bench] native RCCL communicator failed on rank {rank} ({e}); using the torch.distributed hook",
                      file=sys.stderr, flush=True)
                allreduce_used = "hook"
        # every rank must take the same path
        flag = torch.tensor([1 if allreduce_used == "hook" else 0], device=cdev)
        dist.all_reduce(flag, op=dist.ReduceOp.MAX)
        if int(flag.item()) == 1:
            if allreduce_used == "rccl":
                icp.comm_destroy()
            allreduce_used = "hook"
            icp.set_allreduce(sharded.make_allreduce(device=dev))

    p = pkg.Parameters()
    p.matcher_threshold = GATE_M
    p.fixed_iterations = 1
    p.skip_quality = 1
    p.nn_kernel = {"auto": pkg.NN_AUTO, "valu": pkg.NN_VALU, "mfma": pkg.NN_MFMA, "tiled": pkg.NN_TILED}[args.nn_kernel]

    def barrier():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    T0 = np.eye(4)
    if args.warmup > 0:
        p.max_iterations = args.warmup
        icp.align_resident(T0, p)

    p.max_iterations = args.steps
    barrier()
    t0 = time.perf_counter()
    res = icp.align_resident(T0, p)
    barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=cdev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    assert res.nIterations == args.steps, (res.nIterations, args.steps)

    # dominant kernel: the NN matcher; duration from HIP events on the kernel's own stream
    def roofline_of(r, n_local):
        nn_ms = r.ms_nn_kernel / max(1, r.n_nn_launches)
        kern = {1: "valu", 2: "mfma", 3: "tiled"}.get(r.nn_kernel_used, "?")
        flops_alg = 8.0 * n_local * M            # 8 flop per (query, map point) pair, all N*M pairs
        pairs_exec = r.nn_pairs_evaluated / max(1, r.n_nn_launches)
        tf_alg = flops_alg / (nn_ms * 1e-3) / 1e12 if nn_ms > 0 else 0.0
        tf_exe = 8.0 * pairs_exec / (nn_ms * 1e-3) / 1e12 if nn_ms > 0 else 0.0
        flop_view = {"algorithmic_tflops": tf_alg, "algorithmic_frac_of_fp32_peak": tf_alg / PEAK_FP32_TFLOPS,
                     "flops_per_launch": flops_alg, "pairs_evaluated_per_launch": pairs_exec,
                     "executed_tflops": tf_exe, "executed_frac_of_fp32_peak": tf_exe / PEAK_FP32_TFLOPS}
        if kern == "tiled":
            # exact tile culling evaluates ~5e8 of the 1e12 pairs, so the flop view says little about the kernel;
            # its compulsory traffic does: both sorted clouds once + the sorted pairing (pos, idx, d2) written
            bytes_alg = 12.0 * n_local + 12.0 * M + 12.0 * n_local
            gbs = bytes_alg / (nn_ms * 1e-3) / 1e9 if nn_ms > 0 else 0.0
            return {"bound": "hbm", "achieved": gbs, "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": gbs / PEAK_HBM_GBS,
                    "traffic": None, "kernel": "k_nn_tiled", "kernel_ms": nn_ms, "bytes_per_launch": bytes_alg,
                    "culled": True, "flop_view": flop_view}
        return {"bound": "mfma", "achieved": tf_alg, "peak": PEAK_FP32_TFLOPS, "unit": "TFLOP/s",
                "frac": tf_alg / PEAK_FP32_TFLOPS, "traffic": None, "kernel": "k_nn_" + kern, "kernel_ms": nn_ms,
                "flops_per_launch": flops_alg, "culled": False,
                "pipe": "fp32 MFMA" if kern == "mfma" else "fp32 VALU (same 157.3 TFLOP/s peak as the fp32 MFMA)",
                "flop_view": flop_view}

    roof = roofline_of(res, hi - lo)
    roof["traffic"] = _recorded_traffic(roof["kernel"], N, M) if world == 1 else None
    if world == 1 and roof["kernel"] == "k_nn_tiled" and (N, M) == (1_000_000, 1_000_000):
        roof["pmc"] = _recorded_pmc()   # committed counters of this kernel on this workload (VALU busy, waits, L2 latency)
    if world > 1:
        t = torch.tensor([roof["achieved"]], dtype=torch.float64, device=cdev)
        dist.all_reduce(t, op=dist.ReduceOp.MIN)     # the slowest rank's kernel
        roof["achieved"] = float(t[0])
        roof["frac"] = roof["achieved"] / roof["peak"]

    out = {
        "metric": "icp_iterations_per_sec_1Mx1M",
        "value": args.steps / dt,
        "unit": "iterations/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": dt / args.steps * 1e3,
        "higher_is_better": True,
        "scaling": "strong",
        "vs_baseline": None,
        "dtype": "f32",
        "data": "synthetic",
        "config": {"workload": f"configs[2]: {N} scan points vs {M} local-map points, {args.steps} fixed ICP "
                               f"iterations, point-to-point NN (gate {GATE_M} m) + Horn, seed {args.seed}",
                   "n_local": N, "n_map": M, "gate_m": GATE_M, "queries_per_gpu": hi - lo,
                   "parallelism": (f"query-shard x{world}, {allreduce_used} all-reduce" if use_dist else "single GPU"),
                   "nn_kernel": roof["kernel"]},
        "roofline": roof,
        "pose_err_vs_gt": dict(zip(("rot_rad", "trans_m"), _pose_err(res.optimal_tf, T_gt))),
    }

    if rank == 0 and world == 1 and args.nn_kernel in ("auto", "tiled") and args.dense_iters > 0:
        # the dense N x M kernel (no culling) on the same clouds, outside the timed region: its roofline
        pd = p.copy()
        pd.nn_kernel = pkg.NN_MFMA
        pd.max_iterations = args.dense_iters
        t0 = time.perf_counter()
        rd = icp.align_resident(T0, pd)
        torch.cuda.synchronize()
        td = time.perf_counter() - t0
        out["dense_mfma"] = {"value": args.dense_iters / td, "unit": "iterations/s", "iterations": args.dense_iters,
                             "roofline": roofline_of(rd, hi - lo),
                             "pose_err_vs_default_path": dict(zip(("rot_rad", "trans_m"), _pose_err(
                                 rd.optimal_tf, _first_iters_pose(icp, T0, p, args.dense_iters))))}

    if rank == 0 and world == 1 and args.shipped_iters > 0:
        # the reference's shipped pipeline (Point2Plane knn 6 + Gauss-Newton, icp-settings-regular.yaml) on the same clouds
        ps = pkg.Parameters.load_from_file(os.path.join(ROOT, "params", "icp-settings-regular.yaml"))
        ps.fixed_iterations, ps.skip_quality, ps.max_iterations = 1, 1, args.shipped_iters
        icp.align_resident(T0, ps)  # warm-up
        t0 = time.perf_counter()
        rs = icp.align_resident(T0, ps)
        torch.cuda.synchronize()
        ts = time.perf_counter() - t0
        out["shipped_point2plane_gn"] = {"value": args.shipped_iters / ts, "unit": "iterations/s",
                                         "iterations": args.shipped_iters, "knn": int(ps.knn),
                                         "gate_m": float(ps.matcher_threshold),
                                         "kernel_ms": rs.ms_nn_kernel / max(1, rs.n_nn_launches), "pairs": int(rs.n_pairs),
                                         "pose_err_vs_gt": dict(zip(("rot_rad", "trans_m"), _pose_err(rs.optimal_tf, T_gt)))}

    if rank == 0 and world == 1 and args.cpu_baseline_iters > 0:
        out["cpu_baseline"], ref_T = cpu_baseline(g, l, args.cpu_baseline_iters)
        # pose parity on the same pair: GPU vs the CPU oracle after the same number of iterations
        p.max_iterations = args.cpu_baseline_iters
        r5 = icp.align_resident(T0, p)
        rot, trans = _pose_err(r5.optimal_tf, ref_T)
        out["pose_err_vs_cpu"] = {"rot_rad": rot, "trans_m": trans, "iterations": args.cpu_baseline_iters,
                                  "tolerance": "1e-4 rad / 1e-3 m"}

    if use_dist:
        if allreduce_used == "rccl":
            icp.comm_destroy()
        dist.destroy_process_group()
    if rank == 0:
        import ctypes
        ctypes.CDLL(None).fflush(None)   # RCCL's banner sits in C stdio: the JSON must be the LAST line
        sys.stdout.flush()
        print(json.dumps(out), flush=True)


def _recorded_traffic(kernel, N, M):
    """HBM bytes per launch of the dominant kernel from the committed rocprofv3 PMC passes
    (profiles/*/traffic.json, written by tools/rocprof_traffic.sh: FETCH_SIZE and WRITE_SIZE in separate
    passes, FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for gfx950).  None if not recorded."""
    import glob
    best = None
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "*", "traffic.json"))):
        try:
            for e in json.load(open(f)):
                if e.get("kernel") == kernel and e.get("n_local") == N and e.get("n_map") == M:
                    best = e.get("hbm_bytes_per_launch")
        except Exception:
            pass
    return best


def _recorded_pmc():
    """Derived counters of the tiled matcher from the committed rocprofv3 --pmc passes (profiles/*/tiled_pmc_summary.json,
    written from tools/rocprof_lat.sh runs).  None if not recorded."""
    import glob
    best = None
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "*", "tiled_pmc_summary.json"))):
        try:
            best = json.load(open(f)).get("derived")
        except Exception:
            pass
    return best


def _first_iters_pose(icp, T0, p, iters):
    q = p.copy()
    q.max_iterations = iters
    return icp.align_resident(T0, q).optimal_tf


def _pose_err(T, Tref):
    dR = Tref[:3, :3].T @ T[:3, :3]
    c = float(np.clip((np.trace(dR) - 1) / 2, -1, 1))
    return float(np.arccos(c)), float(np.linalg.norm(T[:3, 3] - Tref[:3, 3]))


def cpu_baseline(g, l, iters):
    """The CPU oracle (single-thread exact kd-tree ICP, a port of the reference's mp2p_icp CPU path,
    which cannot be built here) on a bounded sample of the same workload: `iters` fixed iterations of
    the same pair.  A reported baseline, not the optimisation target."""
    from oracle import oracle as O
    op = O.params(max_iterations=iters, matcher_threshold=GATE_M, fixed_iterations=True, use_kdtree=True)
    r = O.align(g, l, np.eye(4), op)
    return ({"value": iters / r["iter_s"], "unit": "iterations/s", "cores": 1, "kind": "port",
             "sample": f"{iters} fixed iterations of the same {l.shape[1]}x{g.shape[1]} pair, single thread, "
                       f"kd-tree build ({r['kdtree_build_s']:.2f} s) excluded",
             "kdtree_build_s": r["kdtree_build_s"], "host_cores_available": os.cpu_count()}, r["T"])


if __name__ == "__main__":
    main()
