#!/usr/bin/env python3
"""bench.py -- ICP iterations/sec at 1M x 1M on MI355X (BASELINE.json metric, configs[2]).

A "step" = ONE ICP iteration of the hot path on HBM-resident clouds: transform the N queries,
brute-force nearest neighbour over the M map points, distance gate, fp64 centroid/covariance
accumulation, (all-reduce of the 24-double block when sharded), Horn solve, stall-test
quantities.  `--steps K` runs exactly K such iterations (stall test disabled, quality pass
skipped) between two barrier+synchronize brackets; `value` = K / max-over-ranks seconds.

N GPUs: strong scaling of the SAME 1M x 1M job -- every rank keeps a compact slice of the scan's Hilbert order and the part of
the map that slice can reach, one all-reduce of the accumulator block per iteration (--allreduce local: the node-local
shared-memory communicator, the default on one node; rccl; hook) -- and, beside it, `c5_sharded`: configs[4], a 10M-point map.

Beside the headline the single-GPU run reports (all outside the timed region, all in the ONE JSON line):
  roofline      the dominant kernel against the HBM roofline: HIP-event duration of the matcher launches of an identical
                profiled repetition; `traffic` / `pmc` from the committed rocprofv3 passes, stamped with the commit and
                date they were recorded at and withheld when the kernel sources have changed since;
  align_e2e     `mola_icp_align` from HOST buffers (upload + Hilbert sort + iterations with the stall test + quality) for
                configs[0], [1], [2], `prepare_ms` broken out, with the CPU checker's end-to-end time INCLUDING its kd-tree
                build beside it;
  config3_batch configs[3]'s shape on this one GPU (64 independent 100k x 100k pairs through `align_batch`) and the
                all-core CPU leg SURVEY.md section 8(d)(ii) asks for: max(2, nproc/2) threads, one pair each
                (src/LidarOdometry.cpp:94-96);
  cpu_baseline  the single-thread CPU leg on a bounded sample of the headline workload, built -O3 -march=native on the
                machine that runs it;
  shipped_point2plane_gn, time_to_pose, cold_start, dense_mfma, loop_closure_montecarlo, config3_batch_shipped
                the reference's own pipeline (Point2Plane + Gauss-Newton) at 1M, its time to termination, the first align
                after an idle second, the dense MFMA kernel's roofline, the batched legs through the loop-closure YAML;
  odometry_stream[_10hz|_small|_small_10hz]
                the front-end mirror over a synthetic drive: scans back to back / arriving every 100 ms / decimated as the
                reference's pipeline decimates them (every 10th point);
  c5_sharded    configs[4] (at N = 1: the whole job on one GPU).

    python bench.py                      # 1 GPU, 40 steps, 3 warmup
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 \
        --master-port 29500 bench.py --gpus 8 --steps 40 --warmup 3
"""
import argparse
import hashlib
import importlib
import json
import os
import sys
import threading
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
    ap.add_argument("--device-warmup-aligns", type=int, default=60,
                    help="untimed repetitions of the K-step align before the W warm-up steps (GPU clocks; 0 = none)")
    ap.add_argument("--n-local", type=int, default=1_000_000)
    ap.add_argument("--n-map", type=int, default=1_000_000)
    ap.add_argument("--seed", type=int, default=42)
    ap.add_argument("--nn-kernel", choices=["auto", "valu", "mfma", "tiled"], default="auto")
    ap.add_argument("--cpu-baseline-iters", type=int, default=-1,
                    help="iterations of the single-thread CPU leg on the headline pair; -1 (default) = --steps, at most 40 (~0.5 s each: the "
                         "pose of the WHOLE timed workload is then compared with the CPU checker's); 0 disables every CPU leg")
    ap.add_argument("--force-dist", action="store_true",
                    help="run the distributed code path (process group + RCCL communicator) even with one rank")
    ap.add_argument("--allreduce", choices=["both", "local", "rccl", "hook"], default="both",
                    help="both (default): the timed align runs through the node-local shared-memory mailbox AND through native RCCL on "
                         "the device block, one after the other -- the line carries both (`allreduce`), `value` is the faster; "
                         "local / rccl / hook: that transport only (hook: torch.distributed from the host hook)")
    ap.add_argument("--guess-dt", type=float, default=1.0, help="sharded runs: how far (m) the pose may end from the guess -- sizes each rank's map slab")
    ap.add_argument("--guess-drot-deg", type=float, default=3.0, help="sharded runs: how far (deg) the pose may rotate from the guess")
    ap.add_argument("--balance-rounds", type=int, default=3,
                    help="configs[4] leg, sharded: rounds of cost-balanced re-cutting of the query shards before anything is timed -- the best cuts measured are kept (0 = equal counts)")
    ap.add_argument("--balance-headline", type=int, default=0, help="the same for the headline's shards (default: equal counts)")
    ap.add_argument("--c5-map", type=int, default=10_000_000, help="map points of the configs[4] leg (0 = skip)")
    ap.add_argument("--c5-steps", type=int, default=20)
    ap.add_argument("--share-gpu", action="store_true",
                    help="testing aid for 1-GPU boxes: ranks share the visible GPUs (rank %% device_count), the process "
                         "group is gloo and the accumulator all-reduce goes through the host hook -- exercises the "
                         "N>1 sharding path end to end without RCCL")
    ap.add_argument("--shipped-iters", type=int, default=20,
                    help="iterations of the shipped Point2Plane+GaussNewton pipeline measured beside the default path")
    ap.add_argument("--dense-iters", type=int, default=3, help="iterations of the dense MFMA kernel measured beside the default path (0 = skip)")
    ap.add_argument("--e2e", type=int, default=1, help="0 skips the align_e2e legs (configs[0], [1], [2] from host buffers)")
    ap.add_argument("--paced-passes", type=int, default=3, help="passes of the 24-scan drive timed with the scans delivered at 10 Hz of wall time")
    ap.add_argument("--batch-pairs", type=int, default=64, help="pairs of the configs[3] leg (0 = skip)")
    args = ap.parse_args()
    if args.cpu_baseline_iters < 0:
        args.cpu_baseline_iters = min(args.steps, 40)

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        # `python bench.py --gpus N` without a launcher: this process becomes the launcher.  It has not touched the GPU
        # (and never will): N FRESH children, one per rank, do -- see _self_launch.
        raise SystemExit(_self_launch(args))

    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: the launcher's world size and --gpus must agree "
                         f"(python bench.py --gpus {args.gpus} launches its own ranks)")
    n_dev = torch.cuda.device_count()   # (counting devices does not initialise the GPU)
    if world > n_dev and not args.share_gpu:
        # fewer devices than ranks (a 1-GPU box rehearsing the N>1 path): the ranks share what is there, gloo carries the
        # all-reduce -- the line says so (config.parallelism, "ranks_share_gpus"); it is not a scaling measurement
        if rank == 0:
            print(f"[bench] {world} ranks on {n_dev} visible GPU(s): sharing them (--share-gpu)", file=sys.stderr, flush=True)
        args.share_gpu = True
    if args.share_gpu:
        local_rank = local_rank % max(1, torch.cuda.device_count())
        if args.allreduce == "rccl" and world > 1:   # (RCCL cannot put two ranks on one device)
            args.allreduce = "local"
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
    lo, hi = sharded.shard_bounds(N, rank, world)   # (the sizes; which points a rank gets is decided by the device cut)

    # inputs resident in HBM before the timed region
    dev = torch.device("cuda", local_rank)
    cdev = torch.device("cpu") if args.share_gpu else dev   # where the bench's own small collectives live
    tg = torch.from_numpy(g).to(dev)
    tl = torch.from_numpy(np.ascontiguousarray(l)).to(dev)
    icp = pkg.ICP(device=local_rank)
    guess_dt, guess_drot = args.guess_dt, np.deg2rad(args.guess_drot_deg)

    def place_clouds(icp_, tg_, tl_, margin_scale=1.0, cuts=None):
        """one GPU: both clouds whole.  Sharded: this rank's slice of the scan's Hilbert order, cut on the device (a random 1/W
        subsample would be W times sparser than the map: every 128-query group would sweep W times more map tiles) -- and only the
        part of the map that shard can reach from any pose within (guess_dt, guess_drot) of the guess: sharded.slab_margin_for_guess,
        from the gate, the stated uncertainty of the guess and how far from the origin THIS shard lies -- not W copies of the map.
        `cuts` = explicit boundaries (cost-balanced: balance_shards below); None = equal counts."""
        if world == 1:
            icp_.set_map(tg_)
            icp_.set_local(tl_)
            return int(tl_.shape[1]), None
        if cuts is None:
            n_shard = icp_.set_local_shard(tl_, rank, world)
        else:
            n_shard = icp_.set_local_shard_range(tl_, cuts[rank], cuts[rank + 1])
        blo0, bhi0 = icp_.shard_reach_box(np.eye(4), 0.0)
        margin = margin_scale * sharded.slab_margin_for_guess(blo0, bhi0, GATE_M, guess_dt, guess_drot)
        blo, bhi = icp_.shard_reach_box(np.eye(4), margin)
        kept = icp_.set_map_slab(tg_, blo, bhi)
        return n_shard, {"map_points_kept": kept, "map_points_total": int(tg_.shape[1]), "margin_m": margin,
                         "margin_rule": f"gate {GATE_M} m + {guess_dt} m + 2 sin({args.guess_drot_deg} deg / 2) x the shard's farthest corner from the origin"}

    def balance_shards(icp_, tg_, tl_, n_total, pp, rounds):
        """Cuts of equal COST (sharded.balanced_cuts): where the guess is far off, the matcher's cost per query varies several-fold
        along the scan -- a rotation error displaces distant queries by metres, their search balls hold a hundred tiles -- and a
        step is as long as its slowest rank (profiles/r04/sharded/shard_step_c5*.jsonl: equal counts leave one rank of eight at
        2.2x the mean).  Every round: each rank times a short align on the shards in force, the W times are summed into one vector
        (torch.distributed; not timed), all ranks cut again at the same places.  Returns (n_shard, slab, record)."""
        cuts = [sharded.shard_bounds(n_total, r, world)[0] for r in range(world)] + [n_total]
        n_shard, slab = place_clouds(icp_, tg_, tl_)
        rec = []
        q = pp.copy()
        q.max_iterations, q.fixed_iterations, q.skip_quality = 4, 1, 1
        best = None   # (cuts, slowest rank's step): the cuts in force at the end are the best ones MEASURED, not the last ones tried
        for rnd in range(rounds + 1 if world > 1 else 0):
            try:
                icp_.align_resident(np.eye(4), q)
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                icp_.align_resident(np.eye(4), q)
                torch.cuda.synchronize()
                mine = (time.perf_counter() - t0) / 4
            except pkg.IcpError:
                mine = float("nan")   # (a slab too small for this probe: keep the cuts in force, the rehearsal below re-cuts the slabs)
            v = torch.zeros(world, dtype=torch.float64, device=cdev)
            v[rank] = mine
            dist.all_reduce(v, op=dist.ReduceOp.SUM)
            cost = v.cpu().numpy()
            if not np.all(np.isfinite(cost)):
                break
            rec.append({"cuts": list(cuts), "step_ms": [float(c) * 1e3 for c in cost]})
            if best is None or float(cost.max()) < best[1]:
                best = (list(cuts), float(cost.max()))
            if rnd == rounds:
                break
            cuts = sharded.balanced_cuts(cuts, cost)
            n_shard, slab = place_clouds(icp_, tg_, tl_, cuts=cuts)
        if best is not None and best[0] != list(cuts):
            cuts = best[0]
            n_shard, slab = place_clouds(icp_, tg_, tl_, cuts=cuts)
        return n_shard, slab, {"rounds": rec, "cuts": list(cuts)}, cuts

    probe_p = pkg.Parameters()
    probe_p.matcher_threshold = GATE_M
    # (the headline keeps equal-count shards unless asked otherwise: at 1M x 1M the ranks' steps are within ~10 % of each other and
    # a re-cut can push a rank across the 131k-query line between the cooperative and the persistent matcher; the configs[4] leg,
    # where one stretch of the scan costs several times the rest, re-cuts by default)
    if world > 1 and args.balance_headline > 0 and not args.share_gpu:
        n_shard, slab, balance, cuts = balance_shards(icp, tg, tl, N, probe_p, args.balance_headline)
    else:
        (n_shard, slab), balance, cuts = place_clouds(icp, tg, tl), None, None
    lo, hi = 0, n_shard                 # (queries_per_gpu below)
    icp.set_global_sizes(N, M)
    allreduce_used = None
    comm_nranks = None
    local_comm = None

    def agree(ok):   # every rank must take the same path
        if not use_dist:
            return ok
        flag = torch.tensor([0 if ok else 1], device=cdev)
        dist.all_reduce(flag, op=dist.ReduceOp.MAX)
        return int(flag.item()) == 0

    def attach(kind):
        """attach one transport to the headline's handle on every rank (or on none: the ranks agree); returns (ok, nranks)"""
        nonlocal local_comm
        if kind == "local":
            try:
                local_comm = icp.comm_init_local()
                n = icp.comm_nranks()   # the ranks that joined the mailbox
                ok = n == world
            except Exception as e:  # noqa: BLE001
                print(f"[bench] node-local communicator failed on rank {rank} ({e})", file=sys.stderr, flush=True)
                ok, n = False, None
            if not agree(ok):
                icp.comm_destroy()
                local_comm = None
                return False, None
            return True, n
        if kind == "rccl":
            if args.share_gpu and world > 1:   # (RCCL cannot put two ranks on one device)
                return False, None
            try:
                icp.comm_init()
                n = icp.comm_nranks()   # what RCCL itself reports (ncclCommCount)
                ok = True
            except Exception as e:
                print(f"[bench] native RCCL communicator failed on rank {rank} ({e})", file=sys.stderr, flush=True)
                ok, n = False, None
            if not agree(ok):
                if ok:
                    icp.comm_destroy()
                return False, None
            return True, n
        icp.set_allreduce(sharded.make_allreduce(device=dev))
        return True, None

    def detach(kind):
        nonlocal local_comm
        if kind in ("local", "rccl"):
            icp.comm_destroy()
            local_comm = None
        else:
            icp.set_allreduce(None)

    p = pkg.Parameters()
    p.matcher_threshold = GATE_M
    p.fixed_iterations = 1
    p.skip_quality = 1
    p.nn_kernel = {"auto": pkg.NN_AUTO, "valu": pkg.NN_VALU, "mfma": pkg.NN_MFMA, "tiled": pkg.NN_TILED}[args.nn_kernel]

    def barrier():
        # (over the node-local communicator when there is one: a process-group barrier on the nccl backend is a kernel launch
        # and a stream wait of its own -- 0.2-0.4 ms, a tenth of the 40 timed steps)
        if use_dist and local_comm is not None:
            local_comm.allreduce(np.zeros(1))
        elif use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    T0 = np.eye(4)

    def rehearse():
        # The whole trajectory of the timed align must stay inside every rank's map slab (the matcher checks it at every
        # pose; a violation on ANY rank reaches all of them through the all-reduce, so they fail -- and re-cut -- together).
        # Rehearse the full run once per margin before anything is timed.
        nonlocal slab
        p.max_iterations = args.steps
        for attempt in range(4):
            try:
                icp.align_resident(T0, p)
                break
            except pkg.IcpError as e:
                if "map slab" not in str(e) or attempt == 3:
                    raise
                _, slab = place_clouds(icp, tg, tl, margin_scale=2.0 ** (attempt + 1), cuts=cuts)   # (the guess was worse than stated)
                slab["recut"] = attempt + 1

    def timed_run(warm_aligns):
        """W warm-up steps, then the timed K steps between barrier + synchronize brackets (max over ranks), then the same align on
        warm state and as the very first align on the pair.  Returns the seconds and the timed align's result."""
        # Device warm-up (untimed, the same count on every rank): the GPU has idled through seconds of host-side cloud
        # generation and its clocks take a few hundred milliseconds of work to settle -- behind 3 warm-up steps alone the
        # matcher launch read 106.5 us, behind 300 of them 100.1 us.  Then the W warm-up steps the caller asked for.
        # (behind a process group the warm-up is longer: for ~0.5-1 s after the last torch.distributed collective -- the rendezvous, the
        # `agree` flags of attach() -- an align runs 20 % slow on this pool, RCCL's progress threads still polling beside the host thread
        # that spins for the sums; 60 aligns = 0.3 s end inside that window, 200 behind it: tools/stateless_probe2.py, profiles/r05)
        if use_dist and warm_aligns >= 10:   # (a caller that asks for a token warm-up -- the tests -- gets it)
            warm_aligns = max(warm_aligns, 250)
        p.max_iterations = args.steps
        for k in range(warm_aligns):
            if k == warm_aligns - 1:
                icp.forget_warm_start(schedule=True)   # (the last one as the timed one will be: an unseeded, unordered first launch has run on this transport before the clock starts)
            icp.align_resident(T0, p)
        if args.warmup > 0:
            p.max_iterations = args.warmup
            icp.align_resident(T0, p)
        # The timed align is STRICTLY STATELESS (round 6): everything the untimed aligns left for the clouds in place is dropped first
        # -- the last pairing (next launch's seeds), the neighbour lists, AND the work queue's per-item cost order, which is made of
        # cycle counts measured in EARLIER aligns (mola_icp_forget_warm_start + mola_icp_forget_cloud_schedule) -- as
        # mp2p_icp::ICP::align() keeps nothing between calls (src/LidarOdometry.cpp:869-871).  The GPU's clocks stay warm; the
        # clouds' sorted form stays resident (the metric's premise).  Beside it: the same align once more on the state the first one
        # left = `value_repeat_on_warm_state` (what rounds 1-4 printed as `value`), and stateless but for the cost order =
        # `value_with_cloud_schedule_kept` (what round 5 printed as `value`).
        p.max_iterations = args.steps
        icp.forget_warm_start(schedule=True)
        barrier()
        t0 = time.perf_counter()
        res = icp.align_resident(T0, p)
        torch.cuda.synchronize()
        dt_own = time.perf_counter() - t0      # (this rank's K steps; the closing barrier below is the contract's)
        barrier()
        dt = time.perf_counter() - t0
        t0 = time.perf_counter()
        res_warm = icp.align_resident(T0, p)
        torch.cuda.synchronize()
        barrier()
        dt_warm = time.perf_counter() - t0
        icp.forget_warm_start()                # (the pairing / seeds go, the work queue's cost order of the aligns above stays)
        barrier()
        t0 = time.perf_counter()
        res_first = icp.align_resident(T0, p)
        torch.cuda.synchronize()
        barrier()
        dt_first = time.perf_counter() - t0
        if world > 1:
            t = torch.tensor([dt, dt_own, dt_warm, dt_first], dtype=torch.float64, device=cdev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt, dt_own, dt_warm, dt_first = float(t[0]), float(t[1]), float(t[2]), float(t[3])
        assert res.nIterations == args.steps, (res.nIterations, args.steps)
        # (the state changes the cost of an align, never its result)
        assert np.array_equal(res_warm.optimal_tf, res.optimal_tf) and np.array_equal(res_first.optimal_tf, res.optimal_tf)
        return {"dt": dt, "dt_own": dt_own, "dt_warm": dt_warm, "dt_first": dt_first, "res": res}

    # Sharded: the timed align runs through EVERY transport asked for (default: the node-local mailbox, then native RCCL on the
    # device block), the line carries each one's step time and the rank count the communicator itself reports (`allreduce`), and
    # `value` is the faster one's -- so that a run on a node answers "did RCCL see N ranks, and what did it cost" in the default line.
    per_transport = {}
    if use_dist:
        kinds = ["local", "rccl"] if args.allreduce == "both" else [args.allreduce]
        runs, first = {}, True
        for kind in kinds:
            ok, n = attach(kind)
            if not ok:
                why = ("two ranks cannot share one device under RCCL (ranks_share_gpus)" if kind == "rccl" and args.share_gpu and world > 1
                       else "the communicator could not be created on every rank (stderr)")
                per_transport[kind] = {"ms_per_step": None, "nranks": None, "note": why}
                continue
            if first and world > 1:
                rehearse()
            runs[kind] = timed_run(args.device_warmup_aligns)
            first = False
            per_transport[kind] = {"ms_per_step": runs[kind]["dt"] / args.steps * 1e3, "nranks": n,
                                   "ms_per_step_repeat_on_warm_state": runs[kind]["dt_warm"] / args.steps * 1e3}
            detach(kind)
        if not runs:   # keep the run alive: torch.distributed carries the 24 doubles instead
            attach("hook")
            if world > 1:
                rehearse()
            runs["hook"] = timed_run(args.device_warmup_aligns)
            per_transport["hook"] = {"ms_per_step": runs["hook"]["dt"] / args.steps * 1e3, "nranks": None}
            detach("hook")
        allreduce_used = min(runs, key=lambda k: runs[k]["dt"])
        _, comm_nranks = attach(allreduce_used)   # (for the profiled repetitions and the configs[4] leg below)
        run = runs[allreduce_used]
        # every transport must have produced the same pose (rank-ordered sums on the host, RCCL's own order on the device: same to
        # the last bits that survive the solve's rounding)
        for k, r in runs.items():
            assert np.allclose(r["res"].optimal_tf, run["res"].optimal_tf, atol=1e-9), (k, allreduce_used)
    else:
        run = timed_run(args.device_warmup_aligns)
    dt, dt_own, dt_warm, dt_first, res = run["dt"], run["dt_own"], run["dt_warm"], run["dt_first"], run["res"]
    # Kernel statistics -- HIP events around every matcher launch, recorded on the library's own stream, and the
    # executed-pair counters -- come from an IDENTICAL repetition right after the timed region (stateless like it): the two event
    # packets per launch cost ~8 us per iteration, which the timed region does not pay (mola_icp_set_profiling, off by default).
    icp.set_profiling(True)
    profs = []
    for _ in range(3):   # (three repetitions, the median one is reported: one in six single repetitions read 25-30 % high on this pool)
        icp.forget_warm_start(schedule=True)
        profs.append(icp.align_resident(T0, p))
        barrier()
    icp.set_profiling(False)
    profs.sort(key=lambda r: r.ms_nn_kernel)
    res_prof = profs[1]
    assert all(np.array_equal(r.optimal_tf, res.optimal_tf) for r in profs)

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
                     "pairs_evaluated_per_query": pairs_exec / max(1, n_local),
                     "executed_tflops": tf_exe, "executed_frac_of_fp32_peak": tf_exe / PEAK_FP32_TFLOPS}
        if kern == "tiled":
            # exact tile culling evaluates ~5e8 of the 1e12 pairs, so the flop view says little about the kernel; its
            # compulsory traffic does.  SURVEY section 8(d): 12 N + 12 M (each cloud once) + 8 N (idx, d2 materialised).
            # (The kernel also writes each neighbour's position and coordinates -- next launch's seeds, the accumulation's
            # g -- and reads the seeds back: 16 N written + 16 N read on top, by design; `traffic` shows all of it.)
            bytes_alg = 12.0 * n_local + 12.0 * M + 8.0 * n_local
            gbs = bytes_alg / (nn_ms * 1e-3) / 1e9 if nn_ms > 0 else 0.0
            return {"bound": "hbm", "achieved": gbs, "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": gbs / PEAK_HBM_GBS,
                    "traffic": None, "kernel": "k_nn_tiled", "kernel_ms": nn_ms, "bytes_per_launch": bytes_alg,
                    "seed_bytes_per_launch_by_design": 32.0 * n_local, "culled": True, "flop_view": flop_view}
        return {"bound": "mfma", "achieved": tf_alg, "peak": PEAK_FP32_TFLOPS, "unit": "TFLOP/s",
                "frac": tf_alg / PEAK_FP32_TFLOPS, "traffic": None, "kernel": "k_nn_" + kern, "kernel_ms": nn_ms,
                "flops_per_launch": flops_alg, "culled": False,
                "pipe": "fp32 MFMA" if kern == "mfma" else "fp32 VALU (same 157.3 TFLOP/s peak as the fp32 MFMA)",
                "flop_view": flop_view}

    roof = roofline_of(res_prof, hi - lo)
    roof["kernel_ms_repetitions"] = [r.ms_nn_kernel / max(1, r.n_nn_launches) for r in profs]   # (sorted; kernel_ms = the median)
    if world == 1:
        roof.update(_recorded_counters(roof["kernel"], N, M))
        # what actually bounds the kernel, beside the HBM fraction the contract asks for: after exact culling the tiled matcher is
        # bound by vector ISSUE (and each item's chain of dependent trips), not by bytes.  valu_busy from the committed counter
        # pass; the share of the VALU instructions that is distance math from THIS run's executed pairs: a pair costs 3.5
        # instructions in the packed sweep (v_pk_add x3, v_pk_mul / v_pk_fma x3 and one v_pk_min per TWO pairs)
        pmc = roof.get("pmc") if isinstance(roof.get("pmc"), dict) else {}
        per_item = pmc.get("valu_insts_per_64_query_item")
        ppq = roof["flop_view"]["pairs_evaluated_per_query"]
        roof["binding"] = {"resource": "VALU issue + per-item chain of dependent round trips (not HBM bytes)",
                           "valu_busy": pmc.get("valu_busy_frac"), "wave_cycles_waiting": pmc.get("wave_cycles_waiting_frac"),
                           "valu_insts_per_64_query_item": per_item,
                           "distance_share_of_valu": (3.5 * ppq / per_item) if per_item else None,
                           "executed_frac_of_fp32_peak": roof["flop_view"]["executed_frac_of_fp32_peak"],
                           "note": "counter-derived entries are null when the committed counters belong to other kernel sources"}
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
        "device_warmup": {"aligns": args.device_warmup_aligns, "steps_each": args.steps, "note": "untimed, before the W warm-up steps: GPU clocks"},
        "ms_per_step": dt / args.steps * 1e3,
        "ms_per_step_before_closing_barrier": dt_own / args.steps * 1e3,
        "state": "strictly stateless align: no pairing / seeds / lists AND no work-queue cost order from earlier aligns (mola_icp_forget_warm_start + "
                 "mola_icp_forget_cloud_schedule before the timed region); resident: the clouds' sorted form; hot clocks.  Round 5's `value` kept the cost "
                 "order (now value_with_cloud_schedule_kept), rounds 1-4's was the repeat on warm state (value_repeat_on_warm_state)",
        "value_repeat_on_warm_state": args.steps / dt_warm,
        "ms_per_step_repeat_on_warm_state": dt_warm / args.steps * 1e3,
        "value_with_cloud_schedule_kept": args.steps / dt_first,
        "ms_per_step_with_cloud_schedule_kept": dt_first / args.steps * 1e3,
        "higher_is_better": True,
        "scaling": "strong",
        "vs_baseline": None,
        "dtype": "f32",
        "data": "synthetic",
        "config": {"workload": f"configs[2]: {N} scan points vs {M} local-map points, {args.steps} fixed ICP "
                               f"iterations, point-to-point NN (gate {GATE_M} m) + Horn, seed {args.seed}",
                   "n_local": N, "n_map": M, "gate_m": GATE_M, "queries_per_gpu": hi - lo,
                   "parallelism": (f"query-shard x{world}, {allreduce_used} all-reduce" if use_dist else "single GPU"),
                   "ranks_share_gpus": bool(args.share_gpu and world > 1), "comm_nranks": comm_nranks,
                   "allreduce": per_transport if use_dist else None,
                   "nn_kernel": roof["kernel"], "map_slab_rank0": slab, "shard_balance": balance},
        "roofline": roof,
        "pose_err_vs_gt": dict(zip(("rot_rad", "trans_m"), _pose_err(res.optimal_tf, T_gt))),
    }
    if args.c5_map > 0:
        out["c5_sharded"] = c5_leg(pkg, synth, sharded, torch, dist, args, rank, world, local_rank, dev, cdev, use_dist, allreduce_used,
                                   local_comm, place_clouds, balance_shards, barrier)
    extras = rank == 0 and world == 1
    if world > 1 and args.batch_pairs > 0:
        # configs[3] on N GPUs: independent pairs, no collective -- the pairs dealt over the RANKS (replicas), then once more from
        # rank 0 alone through the library's own device pool (mola_icp_pool_*) over every visible GPU
        out["config3_batch"] = config3_replicas(pkg, synth, torch, dist, args, rank, world, local_rank, cdev, n_dev)
    if world > 1 and rank == 0 and args.cpu_baseline_iters > 0:
        # the CPU leg belongs in the N > 1 line too (rank 0's host cores; the other ranks wait at the closing barrier)
        from oracle import oracle as O
        out["cpu_baseline"], ref_T = cpu_baseline(g, l, args.cpu_baseline_iters, O.use_native())
        if args.cpu_baseline_iters == args.steps:
            rot, trans = _pose_err(res.optimal_tf, ref_T)
            out["pose_err_vs_cpu"] = {"rot_rad": rot, "trans_m": trans, "iterations": args.steps, "whole_timed_workload": True,
                                      "note": f"the sharded align's pose after all {args.steps} iterations against the CPU checker's", "tolerance": "1e-4 rad / 1e-3 m"}
    if world > 1:
        dist.barrier()

    if extras and args.nn_kernel in ("auto", "tiled") and args.dense_iters > 0:
        # the dense N x M kernel (no culling) on the same clouds, outside the timed region: its roofline
        pd = p.copy()
        pd.nn_kernel = pkg.NN_MFMA
        pd.max_iterations = args.dense_iters
        icp.set_profiling(True)
        t0 = time.perf_counter()
        rd = icp.align_resident(T0, pd)
        torch.cuda.synchronize()
        td = time.perf_counter() - t0
        icp.set_profiling(False)
        out["dense_mfma"] = {"value": args.dense_iters / td, "unit": "iterations/s", "iterations": args.dense_iters,
                             "roofline": roofline_of(rd, hi - lo),
                             "pose_err_vs_default_path": dict(zip(("rot_rad", "trans_m"), _pose_err(
                                 rd.optimal_tf, _first_iters_pose(icp, T0, p, args.dense_iters))))}

    if extras and args.shipped_iters > 0:
        # the reference's shipped pipeline (Point2Plane knn 6 + Gauss-Newton, icp-settings-regular.yaml) on the same clouds
        ps = pkg.Parameters.load_from_file(os.path.join(ROOT, "params", "icp-settings-regular.yaml"))
        ps.fixed_iterations, ps.skip_quality, ps.max_iterations = 1, 1, args.shipped_iters
        icp.align_resident(T0, ps)  # warm-up (allocations, clocks)
        # ... whose neighbour lists, seeds and plane cache would make the timed repeat a different, cheaper run than a registration
        # of a NEW pair: dropped (the sorted clouds stay resident, as the metric says); the warm repeat is reported beside it
        icp.forget_warm_start()
        t0 = time.perf_counter()
        rs = icp.align_resident(T0, ps)
        torch.cuda.synchronize()
        ts = time.perf_counter() - t0
        t0 = time.perf_counter()
        icp.align_resident(T0, ps)
        torch.cuda.synchronize()
        ts_warm = time.perf_counter() - t0
        icp.forget_warm_start()
        icp.set_profiling(True)
        rs = icp.align_resident(T0, ps)
        icp.set_profiling(False)
        # the plane matcher against the HBM roofline, PER ITERATION: an iteration's matcher work is one to three launches
        # (the search, then verification / counting launches of microseconds) -- an average over launches, as round 3
        # printed, weights those as if they were searches.  Bytes: DESIGN section 4 -- both sorted clouds once, and per query
        # the plane pairing written (56 B) + its cached twin (56 B), the K + 1 list positions read and written back
        # (4 (K + 1) B) and the certified bound (4 B): 12 N + 12 M + (112 + 4 (K + 1) + 4) N
        it_ms = rs.ms_nn_kernel / max(1, int(rs.nIterations))
        bytes_alg = 12.0 * N + 12.0 * M + (112.0 + 4.0 * (int(ps.knn) + 1) + 4.0) * N
        pairs_it = rs.nn_pairs_evaluated / max(1, int(rs.nIterations))
        roof_s = {"bound": "hbm", "achieved": bytes_alg / (it_ms * 1e-3) / 1e9 if it_ms > 0 else 0.0, "peak": PEAK_HBM_GBS, "unit": "GB/s",
                  "kernel": "k_knn_q4 (one lane per query)" if N > 245_000 and int(ps.knn) <= 9 else "k_knn_q4", "matcher_ms_per_iteration": it_ms, "bytes_per_iteration": bytes_alg,
                  "launches": int(rs.n_nn_launches), "iterations": int(rs.nIterations),
                  "note": f"matcher time of the {int(rs.nIterations)}-iteration run (HIP events around its {rs.n_nn_launches} launches: key bootstrap, "
                          "list-seeded searches, certified / counting launches) divided by the ITERATIONS, first align on the pair "
                          "(no lists from an earlier align); the kernel is compute-bound in its distance sweep, not HBM-bound: "
                          "`executed_tflops` is the rate that explains its time",
                  "pairs_evaluated_per_query_per_iteration": pairs_it / max(1, N),
                  "executed_tflops": 8.0 * pairs_it / (it_ms * 1e-3) / 1e12 if it_ms > 0 else 0.0}
        roof_s["frac"] = roof_s["achieved"] / PEAK_HBM_GBS
        rec = _recorded_counters("k_knn_planes", N, M)
        if isinstance(rec.get("pmc"), dict):   # (a per-LAUNCH fraction of the recorded pass: not comparable with `frac` above)
            rec["pmc"] = {k: v for k, v in rec["pmc"].items() if k != "frac_of_hbm_peak"}
        roof_s.update(rec)
        out["shipped_point2plane_gn"] = {"value": args.shipped_iters / ts, "unit": "iterations/s",
                                         "state": "resident sorted clouds, no lists / seeds / plane cache from earlier aligns (mola_icp_forget_warm_start)",
                                         "value_repeat_on_warm_lists": args.shipped_iters / ts_warm,
                                         "iterations": args.shipped_iters, "knn": int(ps.knn),
                                         "gate_m": float(ps.matcher_threshold),
                                         "matcher_ms_per_iteration": it_ms, "pairs": int(rs.n_pairs), "roofline": roof_s,
                                         "pose_err_vs_gt": dict(zip(("rot_rad", "trans_m"), _pose_err(rs.optimal_tf, T_gt)))}

    if extras and args.shipped_iters > 0:
        # time to pose: "iterations/s" on a pair whose point-to-point run is still creeping after 40 iterations says nothing about
        # how long a registration takes -- the shipped pipeline with its stall test, resident clouds, from the identity to termination
        pt = pkg.Parameters.load_from_file(os.path.join(ROOT, "params", "icp-settings-regular.yaml"))
        pt.skip_quality = 1
        icp.align_resident(T0, pt)   # warm-up; its lists / seeds / plane cache are dropped: a registration meets a pair once
        firsts = []
        for _ in range(3):           # (median of three: a single 2.6-ms run read 3.6 ms once in four bench lines)
            icp.forget_warm_start()
            t0 = time.perf_counter()
            rt = icp.align_resident(T0, pt)
            torch.cuda.synchronize()
            firsts.append((time.perf_counter() - t0) * 1e3)
        t_first = float(np.median(firsts))
        t0 = time.perf_counter()
        icp.align_resident(T0, pt)
        torch.cuda.synchronize()
        t_rep = (time.perf_counter() - t0) * 1e3
        out["time_to_pose"] = {"pipeline": "icp-settings-regular.yaml (Point2Plane knn 6 + Gauss-Newton, stall test 5e-5 m / 1e-5 rad), resident 1M x 1M clouds, from the identity",
                               "ms": t_first, "state": "sorted clouds resident, nothing kept from earlier aligns (mola_icp_forget_warm_start)",
                               "ms_repetitions": firsts, "ms_repeat_on_warm_lists": t_rep, "iterations": int(rt.nIterations), "termination": rt.termination_name,
                               "pose_err_vs_gt": dict(zip(("rot_rad", "trans_m"), _pose_err(rt.optimal_tf, T_gt)))}
        # ... and what the headline hides: the same 40-step align from a GPU that has idled (clocks down), one shot
        time.sleep(1.5)
        p.max_iterations = args.steps
        t0 = time.perf_counter()
        icp.align_resident(T0, p)
        torch.cuda.synchronize()
        out["cold_start"] = {"ms_per_step": (time.perf_counter() - t0) / args.steps * 1e3,
                             "note": f"the timed {args.steps}-step align once more after the GPU idled 1.5 s: the headline is a hot-clock number"}

    cpu_flags = None
    if extras and args.cpu_baseline_iters > 0:
        from oracle import oracle as O
        cpu_flags = O.use_native()   # -O3 -march=native, built on this machine
        out["cpu_baseline"], ref_T = cpu_baseline(g, l, args.cpu_baseline_iters, cpu_flags)
        # a real mp2p_icp on this box?  (expected absent: reported either way; if it is there and the harness builds, IT is
        # the CPU baseline -- kind "reference" -- and the oracle's leg stays beside it as `cpu_baseline_port`)
        out["reference_probe"] = _probe_reference()
        ref_leg = _reference_baseline(out["reference_probe"], g, l, args.cpu_baseline_iters)
        if ref_leg is not None:
            out["cpu_baseline_port"] = out["cpu_baseline"]
            out["cpu_baseline"], T_ref_real = ref_leg
            out["oracle_vs_reference_pose"] = dict(zip(("rot_rad", "trans_m"), _pose_err(ref_T, T_ref_real)))
        # pose parity on the same pair: GPU vs the CPU oracle after the same number of iterations (the timed run did
        # args.steps iterations; the pairings are bit-identical per iteration, so the first iterations are the same ones)
        p.max_iterations = args.cpu_baseline_iters
        r5 = icp.align_resident(T0, p)
        rot, trans = _pose_err(r5.optimal_tf, ref_T)
        out["pose_err_vs_cpu"] = {"rot_rad": rot, "trans_m": trans, "iterations": args.cpu_baseline_iters,
                                  "whole_timed_workload": args.cpu_baseline_iters == args.steps,
                                  "note": (f"the pose after all {args.steps} iterations of the timed workload against the CPU checker's {args.steps}"
                                           if args.cpu_baseline_iters == args.steps else
                                           f"compared after {args.cpu_baseline_iters} iterations (the timed region runs {args.steps})"),
                                  "tolerance": "1e-4 rad / 1e-3 m"}

    if extras and args.e2e:
        out["align_e2e"] = align_e2e(pkg, synth, icp, g, l, args.seed, args.cpu_baseline_iters > 0, cpu_flags)
    if extras and args.batch_pairs > 0:
        out["config3_batch"] = config3_batch(pkg, synth, icp, args.batch_pairs, args.cpu_baseline_iters > 0, cpu_flags)
        # ... and the same 64 pairs through the settings the reference's nearby / loop-closure checks actually load
        # (params/icp-settings-loop-closure.yaml = Point2Plane + Gauss-Newton): the lockstep device batch since round 3
        out["config3_batch_shipped"] = config3_batch(pkg, synth, icp, args.batch_pairs, False, cpu_flags, shipped=True)
        out["loop_closure_montecarlo"] = montecarlo_leg(pkg, synth, icp)
    if extras and args.e2e:
        out["odometry_stream"] = odometry_stream_leg(pkg, synth)
        # what a robot sees: the same drive with the scans arriving at the sensor's 10 Hz, not back to back (every call follows
        # ~99 ms of sleep: host wake-up, cold caches; tools/paced_probe.py), and on the cloud sizes the reference's filters actually hand to align()
        out["odometry_stream_10hz"] = odometry_stream_leg(pkg, synth, period_s=0.1, passes=args.paced_passes)
        # ... and from page-locked scan buffers (profiles/r05/paced_split_probe.txt: what is dearer at 10 Hz is the upload of a pageable
        # scan to a device that has idled, and the call's entry; the iterations are not)
        out["odometry_stream_10hz_pinned"] = odometry_stream_leg(pkg, synth, period_s=0.1, passes=args.paced_passes, pinned=True)
        # (the decimated drive also runs through the CPU-driven front-end: ~24 aligns of 12k points, seconds of CPU)
        out["odometry_stream_small"] = odometry_stream_leg(pkg, synth, decimate=10, oracle_front_end=args.cpu_baseline_iters != 0)
        out["odometry_stream_small_10hz"] = odometry_stream_leg(pkg, synth, period_s=0.1, decimate=10, passes=args.paced_passes)
        out["mixed_load"] = mixed_load_leg(pkg, synth)

    if use_dist:
        if allreduce_used in ("rccl", "local"):
            icp.comm_destroy()
        dist.destroy_process_group()
    if rank == 0:
        import ctypes
        ctypes.CDLL(None).fflush(None)   # RCCL's banner sits in C stdio: the JSON must be the LAST line
        sys.stdout.flush()
        print(json.dumps(out), flush=True)


def c5_leg(pkg, synth, sharded, torch, dist, args, rank, world, local_rank, dev, cdev, use_dist, allreduce_used, local_comm, place_clouds,
           balance_shards, barrier):
    """BASELINE.json configs[4]: a 10M-point global map vs the 1M-point scan, query-sharded with a map slab per rank and the
    per-iteration all-reduce of the accumulators -- the config the sharded path is FOR (at 1M x 1M a rank's shard is a ~15 us kernel
    behind a fixed ~100 us of launches and host turns; ten times the map puts the time back into the matcher).  Same scan, same
    ground truth, same collective as the headline; its own handle.  At N = 1: the whole job on one GPU -- the base of the curve."""
    N, M5 = args.n_local, args.c5_map
    g5, l5, T_gt = synth.make_pair(N, M5, seed=args.seed)
    tg5 = torch.from_numpy(g5).to(dev)
    tl5 = torch.from_numpy(np.ascontiguousarray(l5)).to(dev)
    del g5
    icp5 = pkg.ICP(device=local_rank)
    p = pkg.Parameters()
    p.matcher_threshold, p.fixed_iterations, p.skip_quality, p.max_iterations = GATE_M, 1, 1, args.c5_steps
    if world > 1 and args.balance_rounds > 0 and not args.share_gpu:   # (ranks time-slicing one device measure each other, not their shards)
        n_shard, slab, balance, cuts = balance_shards(icp5, tg5, tl5, N, p, args.balance_rounds)
    else:
        (n_shard, slab), balance, cuts = place_clouds(icp5, tg5, tl5), None, None
    icp5.set_global_sizes(N, M5)
    T0 = np.eye(4)

    def attach5(kind):
        try:
            if kind == "local":
                icp5.comm_init_local()
                return True, icp5.comm_nranks()
            if kind == "rccl":
                if args.share_gpu and world > 1:
                    return False, None
                icp5.comm_init()
                return True, icp5.comm_nranks()
            icp5.set_allreduce(sharded.make_allreduce(device=dev))
            return True, None
        except Exception as e:  # noqa: BLE001
            print(f"[bench] configs[4] leg: {kind} communicator failed on rank {rank} ({e})", file=sys.stderr, flush=True)
            return False, None

    def detach5(kind):
        if kind in ("local", "rccl"):
            icp5.comm_destroy()
        else:
            icp5.set_allreduce(None)

    def rehearse5():   # = warm-up (allocations, the slab check over the whole trajectory: see the headline)
        nonlocal slab
        for attempt in range(4):
            try:
                icp5.align_resident(T0, p)
                break
            except pkg.IcpError as e:
                if world == 1 or "map slab" not in str(e) or attempt == 3:
                    raise
                _, slab = place_clouds(icp5, tg5, tl5, margin_scale=2.0 ** (attempt + 1), cuts=cuts)
                slab["recut"] = attempt + 1

    def timed5(T0=T0, warm=True):
        icp5.align_resident(T0, p)
        if warm and use_dist and args.device_warmup_aligns >= 10:   # (the same settling time behind the transport's collectives as the headline's
            for _ in range(150):                          # warm-up; a COUNT, the same on every rank: every align is a series of all-reduces)
                icp5.align_resident(T0, p)
        dts, dts_warm = [], []
        for _ in range(3):   # (the median of three timed aligns: one bench line in five caught a 50-ms stall of the box in a single one)
            for warm in (False, True):   # stateless like the headline (nothing kept from the align before), then the repeat on its state
                if not warm:
                    icp5.forget_warm_start()
                barrier()
                t0 = time.perf_counter()
                r = icp5.align_resident(T0, p)
                barrier()
                dt = time.perf_counter() - t0
                if world > 1:
                    t = torch.tensor([dt], dtype=torch.float64, device=cdev)
                    dist.all_reduce(t, op=dist.ReduceOp.MAX)
                    dt = float(t.item())
                (dts_warm if warm else dts).append(dt)
        return {"dt": float(np.median(dts)), "dt_warm": float(np.median(dts_warm)), "dts": dts, "r": r}

    per_transport, chosen = {}, None
    if use_dist:
        kinds = ["local", "rccl"] if args.allreduce == "both" else [allreduce_used]
        runs = {}
        for kind in kinds:
            ok, n = attach5(kind)
            flag = torch.tensor([0 if ok else 1], device=cdev)
            dist.all_reduce(flag, op=dist.ReduceOp.MAX)   # (every rank takes the same path)
            if int(flag.item()) != 0:
                if ok:
                    detach5(kind)
                per_transport[kind] = {"ms_per_step": None, "nranks": None}
                continue
            if not runs:
                rehearse5()
            runs[kind] = timed5()
            per_transport[kind] = {"ms_per_step": runs[kind]["dt"] / args.c5_steps * 1e3, "nranks": n}
            detach5(kind)
        if not runs:
            attach5("hook")
            rehearse5()
            runs["hook"] = timed5()
            per_transport["hook"] = {"ms_per_step": runs["hook"]["dt"] / args.c5_steps * 1e3, "nranks": None}
            detach5("hook")
        chosen = min(runs, key=lambda k: runs[k]["dt"])
        attach5(chosen)
        run = runs[chosen]
    else:
        rehearse5()
        run = timed5()
    dt, dt_warm, dts, r = run["dt"], run["dt_warm"], run["dts"], run["r"]
    icp5.forget_warm_start()
    icp5.set_profiling(True)
    rp = icp5.align_resident(T0, p)
    icp5.set_profiling(False)
    k_ms = rp.ms_nn_kernel / max(1, rp.n_nn_launches)
    # ... and the regime an odometry / map-matching align spends its last iterations in: the same job from a guess 5 cm / 0.2 deg off
    # the ground truth (inside the stated uncertainty of the guess, so the same map slabs hold), through the transport chosen above
    T_near = np.array(T_gt, dtype=np.float64) @ synth.pose_from_xyzypr(0.03, -0.03, 0.03, np.deg2rad(0.2), 0.0, 0.0)
    try:
        near = timed5(T_near, warm=False)
        icp5.forget_warm_start()
        icp5.set_profiling(True)
        rn = icp5.align_resident(T_near, p)
        icp5.set_profiling(False)
        near_leg = {"guess": "ground truth perturbed by (0.03, -0.03, 0.03) m and 0.2 deg of yaw", "value": args.c5_steps / near["dt"], "unit": "iterations/s",
                    "ms_per_step": near["dt"] / args.c5_steps * 1e3, "ms_per_step_repetitions": [x / args.c5_steps * 1e3 for x in near["dts"]],
                    "ms_per_step_repeat_on_warm_state": near["dt_warm"] / args.c5_steps * 1e3,
                    "matcher_ms_per_launch_rank0": rn.ms_nn_kernel / max(1, rn.n_nn_launches),
                    "pairs_evaluated_per_query_rank0": rn.nn_pairs_evaluated / max(1, rn.n_nn_launches) / max(1, n_shard),
                    "pose_err_vs_gt": dict(zip(("rot_rad", "trans_m"), _pose_err(near["r"].optimal_tf, T_gt)))}
    except pkg.IcpError as e:   # (a slab the perturbed guess leaves: reported, never fatal for the line)
        near_leg = {"error": str(e)}
    leg = {"workload": f"configs[4]: {N} scan points vs a {M5}-point global map, {args.c5_steps} fixed iterations, point-to-point (gate {GATE_M} m) + Horn, "
                       "query-sharded, one map slab per rank, one all-reduce of 24 doubles per iteration",
           "value": args.c5_steps / dt, "unit": "iterations/s", "ms_per_step": dt / args.c5_steps * 1e3,
           "ms_per_step_repetitions": [x / args.c5_steps * 1e3 for x in dts], "n_gpus": world, "scaling": "strong",
           "state": "stateless aligns (mola_icp_forget_warm_start before each timed one)",
           "value_repeat_on_warm_state": args.c5_steps / dt_warm, "ms_per_step_repeat_on_warm_state": dt_warm / args.c5_steps * 1e3,
           "regime": "fixed iterations from the identity: the pair is still far from converged when the leg ends (see pose_err_vs_gt) -- the far-from-converged regime only",
           "n_local": N, "n_map": M5, "queries_per_gpu_rank0": n_shard, "map_slab_rank0": slab, "shard_balance": balance,
           "matcher_ms_per_launch_rank0": k_ms, "pairs_evaluated_per_query_rank0": rp.nn_pairs_evaluated / max(1, rp.n_nn_launches) / max(1, n_shard),
           "all_reduce": chosen, "allreduce": per_transport if use_dist else None,
           "pose_err_vs_gt": dict(zip(("rot_rad", "trans_m"), _pose_err(r.optimal_tf, T_gt))),
           "near_converged": near_leg}
    if use_dist and chosen is not None:
        detach5(chosen)
    icp5.close()
    return leg


def _self_launch(args):
    """`python bench.py --gpus N` with no WORLD_SIZE in the environment: start N fresh child processes of this script, one
    per rank (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT set as torch.distributed.run would), BEFORE
    anything in this process has touched the GPU -- a process that has initialised HIP is never re-executed.  Rank 0's
    stdout is this process's stdout (the ONE JSON line); the other ranks' stdout goes to stderr.  Any rank failing -> the
    others are stopped (by their exact PIDs) and the exit code is non-zero."""
    import subprocess
    # (the rendezvous port is picked by bind-and-close: another process can take it before rank 0 binds it.  A launch whose ranks
    #  ALL fail within the first seconds -- the signature of a lost rendezvous -- is repeated once on a new port.)
    for attempt in range(2):
        t_try = time.perf_counter()
        rc, n_failed = _launch_ranks(args, subprocess)
        if rc == 0 or attempt == 1 or n_failed < args.gpus or time.perf_counter() - t_try > 45.0:
            return rc
        print("[bench] every rank failed at once: launching again on another port", file=sys.stderr, flush=True)
    return rc


def _launch_ranks(args, subprocess):
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env0 = dict(os.environ)
    env0.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC only on this pool (RCCL across processes)
    env0["MASTER_ADDR"] = "127.0.0.1"
    env0["MASTER_PORT"] = str(port)
    env0["WORLD_SIZE"] = str(args.gpus)
    env0["LOCAL_WORLD_SIZE"] = str(args.gpus)
    procs = []
    for r in range(args.gpus):
        env = dict(env0)
        env["RANK"] = env["LOCAL_RANK"] = str(r)
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=None if r == 0 else sys.stderr))
    rc = 0
    n_failed = 0
    alive = set(range(args.gpus))
    t_start = time.perf_counter()
    limit_s = float(os.environ.get("MOLA_BENCH_LAUNCH_TIMEOUT_S", "1500"))   # a rank that never returns must not hold the node for ever
    while alive:
        if time.perf_counter() - t_start > limit_s:
            print(f"[bench] ranks {sorted(alive)} still running after {limit_s:.0f} s: stopping them", file=sys.stderr, flush=True)
            for o in sorted(alive):
                procs[o].kill()
            return 124, 0
        for r in sorted(alive):
            code = procs[r].poll()
            if code is None:
                continue
            alive.discard(r)
            n_failed += 1 if code != 0 else 0
            if code != 0 and rc == 0:
                rc = code if code > 0 else 1
                print(f"[bench] rank {r} exited with code {code}: stopping the other ranks", file=sys.stderr, flush=True)
                for o in sorted(alive):
                    procs[o].terminate()
        time.sleep(0.05)
    return rc, n_failed


def kernel_sources_sha1():
    """identifies the device code the committed counter files were recorded with"""
    h = hashlib.sha1()
    d = os.path.join(ROOT, "mola-fe-lidar_amd", "csrc")
    for f in sorted(os.listdir(d)):
        if f.startswith("kernels_") or f in ("hip_backend.hip", "map_sort.hip", "q4_launch.hip", "knn_q4_launch.hip"):
            h.update(f.encode())
            h.update(open(os.path.join(d, f), "rb").read())
    return h.hexdigest()


def _recorded_counters(kernel, N, M):
    """`traffic` (HBM bytes per launch) and `pmc` of the dominant kernel from the committed rocprofv3 --pmc passes
    (profiles/*/counters.json, written by tools/rocprof_headline.sh + tools/pmc_record.py: FETCH_SIZE and WRITE_SIZE in
    separate passes, FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for gfx950).  Every record carries the commit,
    the date and the SHA-1 of the kernel sources it was measured with: a record of OTHER sources is withheld."""
    import glob
    sha = kernel_sources_sha1()
    stale = None
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "*", "counters.json")), reverse=True):
        try:
            for e in json.load(open(f)):
                if e.get("kernel") == kernel and e.get("n_local") == N and e.get("n_map") == M:
                    stamp = {"commit": e.get("commit"), "date": e.get("date"), "file": os.path.relpath(f, ROOT)}
                    if e.get("kernel_sources_sha1") == sha:
                        return {"traffic": e.get("hbm_bytes_per_launch"), "pmc": e.get("derived"), "counters_recorded_at": stamp}
                    stale = stale or stamp
        except Exception:
            pass
    return {"traffic": None, "counters_recorded_at": None,
            "counters_note": ("withheld: the committed counters were recorded with other kernel sources "
                              f"({stale})" if stale else "no committed counters for this kernel and workload")}


def _first_iters_pose(icp, T0, p, iters):
    q = p.copy()
    q.max_iterations = iters
    return icp.align_resident(T0, q).optimal_tf


def _pose_err(T, Tref):
    dR = Tref[:3, :3].T @ T[:3, :3]
    c = float(np.clip((np.trace(dR) - 1) / 2, -1, 1))
    return float(np.arccos(c)), float(np.linalg.norm(T[:3, 3] - Tref[:3, 3]))


def _probe_reference():
    """SURVEY section 8(c), last row / 8(d)(iii): is a REAL mp2p_icp (the third-party library the reference calls at
    src/LidarOdometry.cpp:869-871 and requires at CMakeLists.txt:23) installed on the box this runs on?  It is the only
    thing that could pin the oracle.  Looks for the Python module, the shared library and the CMake package in the
    usual prefixes; builds oracle/ref_probe/mp2p_icp_time.cpp against it when headers and library are there.  Never
    raises; returns a small report (found / where / harness path or why not)."""
    import glob
    import shutil
    import subprocess
    rep = {"found": False}
    try:
        import importlib.util
        if importlib.util.find_spec("mp2p_icp") is not None:
            rep["python_module"] = True
    except Exception:
        pass
    prefixes = [p for p in (os.environ.get("CMAKE_PREFIX_PATH", "").split(":") + ["/usr", "/usr/local", "/opt/ros/*", os.path.expanduser("~/.local")]) if p]
    libs, cfgs, incs = [], [], []
    for pre in prefixes:
        for pp in glob.glob(pre):   # fixed, shallow patterns only (a recursive walk of /usr takes for ever)
            for sub in ("lib", "lib64", "lib/x86_64-linux-gnu"):
                libs += glob.glob(os.path.join(pp, sub, "libmp2p_icp.so*"))
                for cm in ("cmake/mp2p_icp", "mp2p_icp/cmake"):
                    cfgs += glob.glob(os.path.join(pp, sub, cm, "mp2p_icp*onfig.cmake"))
            cfgs += glob.glob(os.path.join(pp, "share", "mp2p_icp", "cmake", "mp2p_icp*onfig.cmake"))
            incs += glob.glob(os.path.join(pp, "include", "mp2p_icp", "ICP.h")) + glob.glob(os.path.join(pp, "include", "*", "mp2p_icp", "ICP.h"))
    if libs or cfgs or rep.get("python_module"):
        rep.update(found=True, libraries=libs[:3], cmake_packages=cfgs[:3], headers=incs[:1])
    if libs and incs and shutil.which("g++"):
        src = os.path.join(ROOT, "oracle", "ref_probe", "mp2p_icp_time.cpp")
        out = os.path.join(ROOT, "oracle", "_ref", "mp2p_icp_time")
        os.makedirs(os.path.dirname(out), exist_ok=True)
        inc = os.path.dirname(os.path.dirname(incs[0]))
        cmd = ["g++", "-O2", "-std=c++17", src, "-o", out, "-I", inc, "-L", os.path.dirname(libs[0]), "-lmp2p_icp",
               "-lmrpt-maps", "-lmrpt-obs", "-lmrpt-poses", "-lmrpt-math", "-lmrpt-containers", "-lmrpt-rtti", "-lmrpt-serialization", "-lmrpt-core", "-lmrpt-system",
               "-Wl,-rpath," + os.path.dirname(libs[0])]
        try:
            r = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
            rep["harness"] = out if r.returncode == 0 else None
            if r.returncode != 0:
                rep["harness_build_error"] = r.stderr[-400:]
        except Exception as e:  # noqa: BLE001
            rep["harness_build_error"] = str(e)
    return rep


def _reference_baseline(rep, g, l, iters):
    """times the real mp2p_icp through oracle/_ref/mp2p_icp_time (see _probe_reference); None if that is not possible"""
    import subprocess
    import tempfile
    if not rep.get("harness"):
        return None
    try:
        with tempfile.NamedTemporaryFile(suffix=".bin", delete=False) as f:
            np.array([g.shape[1], l.shape[1]], dtype=np.uint64).tofile(f)
            np.ascontiguousarray(g.T, dtype=np.float32).tofile(f)
            np.ascontiguousarray(l.T, dtype=np.float32).tofile(f)
            path = f.name
        r = subprocess.run([rep["harness"], path, str(iters), str(GATE_M)], capture_output=True, text=True, timeout=900)
        os.unlink(path)
        if r.returncode != 0:
            return None
        kv = r.stdout.strip()
        n = int(kv.split("iterations=")[1].split()[0])
        s = float(kv.split("seconds=")[1].split()[0])
        T = np.array([float(v) for v in kv.split("T=")[1].split("quality=")[0].split()]).reshape(4, 4)
        return ({"value": n / s, "unit": "iterations/s", "cores": 1, "kind": "reference",
                 "sample": f"{n} fixed iterations of the same pair through the installed mp2p_icp (kd-tree build inside the time)",
                 "host_cores_available": os.cpu_count()}, T)
    except Exception:
        return None


def cpu_baseline(g, l, iters, flags):
    """The CPU oracle (single-thread exact kd-tree ICP, a port of the reference's mp2p_icp CPU path,
    which cannot be built here) on a bounded sample of the same workload: `iters` fixed iterations of
    the same pair.  A reported baseline, not the optimisation target."""
    from oracle import oracle as O
    op = O.params(max_iterations=iters, matcher_threshold=GATE_M, fixed_iterations=True, use_kdtree=True)
    r = O.align(g, l, np.eye(4), op)
    return ({"value": iters / r["iter_s"], "unit": "iterations/s", "cores": 1, "kind": "port",
             "sample": f"{iters} fixed iterations of the same {l.shape[1]}x{g.shape[1]} pair, single thread, "
                       f"kd-tree build ({r['kdtree_build_s']:.2f} s) excluded here and included in align_e2e",
             "kdtree_build_s": r["kdtree_build_s"], "host_cores_available": os.cpu_count(), "compiler_flags": flags}, r["T"])


def _median_align(icp, g, l, T0, p, reps=5):
    """wall time of `mola_icp_align` from host buffers (upload + sort + iterations + quality): the median of `reps` runs
    and THAT run's result (its prepare / loop / quality split)"""
    icp.align(g, l, T0, p)   # first call: allocations
    tw = time.perf_counter()
    while time.perf_counter() - tw < 0.3:   # (the GPU idles through the CPU checker's seconds before every leg but the first: clocks)
        icp.align(g, l, T0, p)
    runs = []
    for _ in range(reps):
        t0 = time.perf_counter()
        r = icp.align(g, l, T0, p)
        runs.append((time.perf_counter() - t0, r))
    runs.sort(key=lambda x: x[0])
    t, r = runs[len(runs) // 2]
    return t * 1e3, r


def align_e2e(pkg, synth, icp, g1m, l1m, seed, with_cpu, cpu_flags):
    """End-to-end `mola_icp_align` from HOST buffers per config, the CPU checker's end-to-end (kd-tree build included) beside."""
    from oracle import oracle as O
    out = {}
    # configs[0]: one KITTI-like scan pair (64-ring model of the scene, ~115k points, 1 m apart) through the odometry
    # case of params/kitti-default.yaml = the reference's shipped Point2Plane + Gauss-Newton pipeline
    lp = pkg.LidarOdometryParams.load_from_file(os.path.join(ROOT, "params", "kitti-default.yaml"), ROOT)
    p0 = lp.icp_case("with_vel")
    a = synth.lidar_scan(synth.pose_from_xyzypr(-10.0, 0.2, 0, 0.01, 0, 0), seed=11)
    b = synth.lidar_scan(synth.pose_from_xyzypr(-9.0, 0.25, 0, 0.02, 0, 0), seed=12)
    cases = [("config0", "KITTI-like pair through kitti-default.yaml (Point2Plane knn 6 + Gauss-Newton, <= 100 its, stall test)", a, b, p0, "p2pl"),
             ("config1", "100k x 100k, point-to-point (gate 1.0 m) + Horn, <= 100 its, stall test", None, None, None, "p2p"),
             ("config2", "1M x 1M, point-to-point (gate 1.0 m) + Horn, 40 fixed iterations + quality", g1m, l1m, None, "p2p")]
    for name, what, gg, ll, pp, kind in cases:
        if gg is None:
            gg, ll, _ = synth.make_pair(100_000, 100_000, seed=seed)
        if pp is None:
            pp = pkg.Parameters()
            pp.matcher_threshold = GATE_M
            if name == "config2":
                pp.max_iterations, pp.fixed_iterations = 40, 1
            else:
                pp.max_iterations, pp.min_abs_step_trans, pp.min_abs_step_rot = 100, 5e-5, 1e-5
        ms, r = _median_align(icp, gg, ll, np.eye(4), pp)
        e = {"workload": what, "n_from": int(gg.shape[1]), "n_to": int(ll.shape[1]),
             "gpu": {"ms": ms, "prepare_ms": r.ms_upload, "iterations_ms": r.ms_iterations, "quality_ms": r.ms_quality,
                     "iterations": int(r.nIterations), "termination": r.termination_name, "quality": r.quality,
                     "note": "host buffers in, pose out: H2D + Hilbert sort + tile boxes (prepare_ms), the loop, the PairedRatio pass"}}
        if with_cpu:
            op = O.params_from_product(pp)
            if name == "config2":    # bounded: 5 of the 40 iterations, the kd-tree build in full
                op.max_iterations = 5
            t0 = time.perf_counter()
            rc = (O.align_p2pl(gg, ll, np.eye(4), op, pp.plane_eigen_threshold, int(pp.knn), int(pp.solver_max_iterations))
                  if kind == "p2pl" else O.align(gg, ll, np.eye(4), op))
            wall = (time.perf_counter() - t0) * 1e3
            c = {"ms": wall, "iterations": int(rc["n_iterations"]), "cores": 1, "kind": "port", "compiler_flags": cpu_flags,
                 "note": "kd-tree build + iterations + quality pass, single thread"}
            if "kdtree_build_s" in rc:
                c["kdtree_build_ms"] = rc["kdtree_build_s"] * 1e3
            if name == "config2":
                per_it = rc["iter_s"] / max(1, rc["n_iterations"]) * 1e3
                c["note"] += "; bounded sample: 5 of the 40 iterations timed"
                c["ms_extrapolated_40_iterations"] = wall + 35 * per_it
            else:
                rot, trans = _pose_err(r.optimal_tf, rc["T"])
                e["pose_err_vs_cpu"] = {"rot_rad": rot, "trans_m": trans, "same_iterations": int(r.nIterations) == int(rc["n_iterations"])}
            e["cpu"] = c
        out[name] = e
    return out


def odometry_stream_leg(pkg, synth, n_scans=24, period_s=None, decimate=1, passes=1, pinned=False, oracle_front_end=False):
    """rows f1 + f4 (src/LidarOdometry.cpp:190-514): a drive down the scene at 10 m/s, one 64-ring scan (~115k points) every 0.1 s,
    through the front-end mirror (`LidarOdometry.on_new_observation` = `mola_lo_process_scan`) with params/kitti-default.yaml:
    per scan, the new cloud is uploaded, sorted and boxed ONCE (it is `to` now and `from` for the next scan: the cloud cache),
    aligned against the previous scan with the constant-velocity guess, the keyframe / twist bookkeeping runs on the host."""
    lp = pkg.LidarOdometryParams.load_from_file(os.path.join(ROOT, "params", "kitti-default.yaml"), ROOT)
    scans, pinned_keep, gt = [], [], []
    for k in range(n_scans):
        pose = synth.pose_from_xyzypr(-14.0 + 1.0 * k, 0.3 * np.sin(0.3 * k), 0.0, 0.005 * k, 0, 0)
        gt.append(np.array(pose, dtype=np.float64) @ synth.pose_from_xyzypr(0, 0, 1.73, 0, 0, 0))   # the SENSOR's pose (synth.lidar_scan)
        pc = synth.lidar_scan(pose, seed=50 + k)
        # decimate = 10: what the reference's own pipeline hands to align() -- `full_pointcloud_decimation: 10`
        # (params/kitti-default.yaml:27; src/LidarOdometry.cpp:215-224): every tenth point of the scan, ~12k of ~120k
        pc = np.ascontiguousarray(pc[:, ::decimate])
        if pinned:   # the scan in page-locked host memory, as a sensor driver that DMAs into a registered buffer hands it over: the
            import torch   # library's hipMemcpyAsync is then a DMA, not the runtime's staged copy of pageable memory
            keep = torch.from_numpy(pc).pin_memory()
            pinned_keep.append(keep)
            pc = keep.numpy()
        scans.append((100.0 + 0.1 * k, pc))
    icp = pkg.ICP(device=0)
    lo = pkg.LidarOdometry(lp, icp=icp)
    ms, its, ran, kfs, steady, steady_native = [], [], 0, [], [], []
    for rep in range(1 + passes):   # (the first pass warms allocations and clocks; the others are reported)
        lo.reset()
        ms, its, ran, kfs, ms_nat, rels = [], [], 0, [], [], []
        paced = period_s is not None and rep >= 1   # (the warming pass runs back to back)
        t_next = time.perf_counter()
        for k, (t, pc) in enumerate(scans):
            if paced:   # the scan arrives when the sensor delivers it: the GPU and the host thread have idled since the last one
                t_next += period_s
                time.sleep(max(0.0, t_next - time.perf_counter()))
            t0 = time.perf_counter()
            st = lo.on_new_observation(t + 1000.0 * rep, pc)
            ms.append((time.perf_counter() - t0) * 1e3)
            ms_nat.append(st.ms_native)
            rels.append(np.array(st.rel_pose) if st.icp is not None else np.eye(4))
            if st.icp is not None:
                ran += 1
                its.append(int(st.icp.nIterations))
            if st.keyframe_created:
                kfs.append(k)
        if rep >= 1:
            steady += ms[2:]   # (scan 0 has no partner, scan 1 no velocity yet)
            steady_native += ms_nat[2:]
    lo.close()
    trajectory = _trajectory_error(rels, gt)
    if oracle_front_end:   # the same drive through the front-end logic with the CPU checker behind it (src/LidarOdometry.cpp:305-337)
        trajectory["vs_cpu_front_end"] = _oracle_front_end_trajectory(pkg, lp, scans, rels)
    med = float(np.median(steady))
    arrival = (f"delivered every {period_s * 1e3:.0f} ms of wall time (the sensor's rate: host and GPU as a robot meets them)"
               if period_s is not None else "delivered back to back (GPU clocks stay up)")
    return {"workload": f"{passes} x {n_scans} scans of ~{int(np.mean([pc.shape[1] for _, pc in scans]))} points ({'every ' + str(decimate) + 'th point of the ' if decimate > 1 else 'the full '}64-ring scan), 0.1 s and 1 m apart, "
                        f"{arrival}, params/kitti-default.yaml, {'PAGE-LOCKED ' if pinned else ''}host buffers in, pose out",
            "ms_per_scan_median": med, "ms_per_scan_median_c_call_only": float(np.median(steady_native)),
            "ms_per_scan_p99": float(np.percentile(steady, 99)),
            "ms_per_scan_min": float(np.min(steady)), "ms_per_scan_max": float(np.max(steady)),
            "scans_per_s": 1e3 / med, "realtime_factor_at_10_hz": 100.0 / med, "icp_ran": ran, "iterations_per_scan_median": float(np.median(its)) if its else 0.0,
            "ms_per_scan": [round(float(v), 3) for v in ms], "keyframes": kfs, "trajectory": trajectory}


def mixed_load_leg(pkg, synth, n_scans=24, period_s=0.1, n_threads=None):
    """The reference's concurrent load as a measured case (src/LidarOdometry.cpp:94-96, 183-184, 711-712, 767-788, 869): ONE handle; the
    odometry stream at the sensor's 10 Hz on its own thread -- its device work on streams of the greatest priority (mola_lo_process_scan
    raises the calling thread's class) -- while T threads loop loop-closure checks (10 Monte-Carlo guesses on a 100k x 100k pair each,
    params/icp-settings-loop-closure.yaml) on normal-priority streams.  T = min(8, max(2, nproc / 2)): the reference sizes its pool
    max(2, hw / 2) but posts at most max_nearby_align_checks + 1 = 6 checks per keyframe.  Odometry p50 / p99 with and without the
    load (the C call alone: Python's marshalling shares the GIL with the load threads), and checks per second."""
    lp = pkg.LidarOdometryParams.load_from_file(os.path.join(ROOT, "params", "kitti-default.yaml"), ROOT)
    scans = []
    for k in range(n_scans):
        pose = synth.pose_from_xyzypr(-14.0 + 1.0 * k, 0.3 * np.sin(0.3 * k), 0.0, 0.005 * k, 0, 0)
        scans.append((100.0 + 0.1 * k, synth.lidar_scan(pose, seed=50 + k)))
    g, l, _ = synth.make_pair(100_000, 100_000, seed=42)
    n_thr = n_threads or min(8, max(2, (os.cpu_count() or 4) // 2))
    lod = importlib.import_module("mola-fe-lidar_amd.lidar_odometry")
    icp = pkg.ICP(device=0)
    lo = pkg.LidarOdometry(lp, icp=icp)
    guess = np.array([0.1, 0.05, 0.0, 0.01, 0.0, 0.0])

    def odometry_pass(tag):
        lo.reset()
        ms = []
        t_next = time.perf_counter()
        for k, (t, pc) in enumerate(scans):
            t_next += period_s
            time.sleep(max(0.0, t_next - time.perf_counter()))
            st = lo.on_new_observation(t + 1000.0 * tag, pc)
            if k >= 2:
                ms.append(st.ms_native)
        return ms

    for k, (t, pc) in enumerate(scans):   # warm-up pass, back to back
        lo.on_new_observation(t, pc)
    lod.check_nonadjacent(lp, g, l, guess, True, seed=1, icp=icp)   # (allocations of the batched path)
    quiet = odometry_pass(1) + odometry_pass(2)
    stop = threading.Event()
    counts = [0] * n_thr

    def load(i):
        while not stop.is_set():
            lod.check_nonadjacent(lp, g, l, guess, True, seed=100 + i, icp=icp)
            counts[i] += 1
    th = [threading.Thread(target=load, args=(i,)) for i in range(n_thr)]
    for t in th:
        t.start()
    time.sleep(0.3)
    c0, t0 = sum(counts), time.perf_counter()
    loaded = odometry_pass(3) + odometry_pass(4)
    c1, t1 = sum(counts), time.perf_counter()
    stop.set()
    for t in th:
        t.join()
    lo.close()
    icp.close()
    q50, l50 = float(np.median(quiet)), float(np.median(loaded))
    return {"workload": f"odometry stream of ~{int(np.mean([pc.shape[1] for _, pc in scans]))}-point scans at 10 Hz (params/kitti-default.yaml) on one thread; {n_thr} threads "
                        f"looping loop-closure checks (10 Monte-Carlo guesses, 100k x 100k, icp-settings-loop-closure.yaml) on the SAME handle",
            "load_threads": n_thr, "odometry_ms_c_call": {"quiet_p50": q50, "quiet_p99": float(np.percentile(quiet, 99)),
                                                         "loaded_p50": l50, "loaded_p99": float(np.percentile(loaded, 99)), "loaded_max": float(np.max(loaded))},
            "p50_loaded_over_quiet": l50 / q50, "checks_per_s_under_odometry": (c1 - c0) / (t1 - t0),
            "stream_priorities": "odometry: the device's greatest stream priority (mola_icp_set_thread_priority inside mola_lo_process_scan); checks: default",
            "wait_policy": "spin (default)"}


def _trajectory_error(rels, gt):
    """row f1's end-to-end metric: the scan-to-scan poses the front-end adopted (src/LidarOdometry.cpp:299-311), chained from the
    first scan's true pose, against the synthetic drive's ground-truth sensor poses"""
    X = np.array(gt[0])
    et, er, rel_t = [], [], []
    for k in range(1, len(gt)):
        X = X @ rels[k]
        dt, dr = X[:3, 3] - gt[k][:3, 3], _pose_err(X, gt[k])[0]
        et.append(float(np.linalg.norm(dt)))
        er.append(dr)
        rel_t.append(_pose_err(rels[k], np.linalg.inv(gt[k - 1]) @ gt[k])[1])
    path = float(sum(np.linalg.norm(gt[k][:3, 3] - gt[k - 1][:3, 3]) for k in range(1, len(gt))))
    return {"against": "ground-truth sensor poses of the synthetic drive (scan-to-scan poses chained from the first scan's true pose)",
            "scans": len(gt), "path_m": path, "translation_rmse_m": float(np.sqrt(np.mean(np.square(et)))), "translation_max_m": float(np.max(et)),
            "final_drift_m": et[-1], "final_drift_percent_of_path": 100.0 * et[-1] / path, "final_rot_err_rad": er[-1],
            "scan_to_scan_translation_err_median_m": float(np.median(rel_t)), "scan_to_scan_translation_err_max_m": float(np.max(rel_t))}


def _oracle_front_end_trajectory(pkg, lp, scans, rels_gpu):
    """the same scans through the same front-end logic (`mola_lo_process_scan`) with the CPU checker as its registration: per-scan
    and accumulated difference between the two trajectories (tolerance of the path: 1e-4 rad / 1e-3 m per align)"""
    from oracle import oracle as O

    def align(f, t, T0, p):
        op = O.params_from_product(p)
        r = (O.align_p2pl(f, t, T0, op, p.plane_eigen_threshold, int(p.knn), int(p.solver_max_iterations))
             if p.matcher_class == pkg._lib.MATCHER_POINT2PLANE else O.align(f, t, T0, op))
        return r["T"], r["quality"], r["n_iterations"], r["termination"]
    lo = pkg.LidarOdometry(lp, align_fn=align)
    Xg, Xc, worst_rot, worst_trans, same_its = np.eye(4), np.eye(4), 0.0, 0.0, True
    t0 = time.perf_counter()
    for k, (t, pc) in enumerate(scans):
        st = lo.on_new_observation(t, pc)
        rel = np.array(st.rel_pose) if st.icp is not None else np.eye(4)
        rot, trans = _pose_err(rels_gpu[k], rel)
        worst_rot, worst_trans = max(worst_rot, rot), max(worst_trans, trans)
        Xg, Xc = Xg @ rels_gpu[k], Xc @ rel
    lo.close()
    rot, trans = _pose_err(Xg, Xc)
    return {"scans": len(scans), "worst_scan_to_scan": {"rot_rad": worst_rot, "trans_m": worst_trans},
            "accumulated": {"rot_rad": rot, "trans_m": trans}, "cpu_seconds": time.perf_counter() - t0, "tolerance_per_align": "1e-4 rad / 1e-3 m"}


def montecarlo_leg(pkg, synth, icp, n_guesses=10, n=100_000):
    """row f2 (src/LidarOdometry.cpp:767-788): 10 perturbed guesses on one 100k x 100k pair through `align_multi_init`, host
    buffers in -- the point-to-point pipeline and the reference's own loop-closure settings (Point2Plane + Gauss-Newton)"""
    g, l, _ = synth.make_pair(n, n, seed=42)
    rng = np.random.default_rng(7)
    guesses = []
    for _ in range(n_guesses):
        d = rng.normal(0, 1, 4) * np.array([0.3, 0.3, 0.3, np.deg2rad(2.0)])
        guesses.append(synth.pose_from_xyzypr(d[0], d[1], d[2], d[3], 0, 0))
    out = {"workload": f"{n_guesses} guesses (sigma 0.3 m / 2 deg) on one {n} x {n} pair, <= 100 its with the stall test, host buffers in"}
    p2p = pkg.Parameters()
    p2p.matcher_threshold, p2p.max_iterations, p2p.min_abs_step_trans, p2p.min_abs_step_rot = GATE_M, 100, 5e-5, 1e-5
    shipped = pkg.Parameters.load_from_file(os.path.join(ROOT, "params", "icp-settings-loop-closure.yaml"))
    for name, p in (("point_to_point", p2p), ("shipped_loop_closure_yaml", shipped)):
        tw = time.perf_counter()
        while time.perf_counter() - tw < 0.3:
            icp.align_multi_init(g, l, guesses, p)
        ts = []
        for _ in range(5):
            t0 = time.perf_counter()
            res, best = icp.align_multi_init(g, l, guesses, p)
            ts.append(time.perf_counter() - t0)
        out[name] = {"ms": float(np.median(ts)) * 1e3, "iterations": [int(r.nIterations) for r in res], "best": int(best)}
    return out


def config3_replicas(pkg, synth, torch, dist, args, rank, world, local_rank, cdev, n_dev):
    """BASELINE.json configs[3] as it is meant: "64 independent 100k-pt scan pairs sharded across 8 x MI355X, stream-per-pair, no RCCL"
    (the batch the reference's pool threads run: src/LidarOdometry.cpp:704-741).  (i) replicas: pair k belongs to rank k % W, every
    rank runs its pairs through `align_batch` on its own GPU, pairs/s = all pairs / the slowest rank's time; (ii) rank 0 alone, the
    same pairs through `mola_icp_pool_*` (one handle per visible device inside ONE process) while the other ranks wait.  Both through
    the point-to-point settings and through params/icp-settings-loop-closure.yaml."""
    n_pairs = args.batch_pairs
    mine = [k for k in range(n_pairs) if k % world == rank]
    pairs = {k: synth.make_pair(100_000, 100_000, seed=100 + k)[:2] for k in (range(n_pairs) if rank == 0 else mine)}
    p2p = pkg.Parameters()
    p2p.matcher_threshold, p2p.max_iterations, p2p.min_abs_step_trans, p2p.min_abs_step_rot = GATE_M, 100, 5e-5, 1e-5
    shipped = pkg.Parameters.load_from_file(os.path.join(ROOT, "params", "icp-settings-loop-closure.yaml"))
    out = {"workload": f"{n_pairs} independent 100k x 100k pairs (seeds 100..{99 + n_pairs}), <= 100 its with the stall test, host buffers in, no collective",
           "n_gpus": world, "pairs_per_rank": [len([k for k in range(n_pairs) if k % world == r]) for r in range(world)]}
    icp_b = pkg.ICP(device=local_rank)
    my_pairs = [pairs[k] for k in mine]
    for name, p in (("point_to_point", p2p), ("shipped_loop_closure_yaml", shipped)):
        tw = time.perf_counter()
        while my_pairs and time.perf_counter() - tw < 0.5:
            icp_b.align_batch(my_pairs[:12], [np.eye(4)] * min(len(my_pairs), 12), p)
        dts = []
        for _ in range(3):
            dist.barrier()
            t0 = time.perf_counter()
            res = icp_b.align_batch(my_pairs, [np.eye(4)] * len(my_pairs), p) if my_pairs else []
            dt = time.perf_counter() - t0
            t = torch.tensor([dt], dtype=torch.float64, device=cdev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dts.append(float(t.item()))
        its = torch.tensor([float(sum(r.nIterations for r in res))], dtype=torch.float64, device=cdev)
        dist.all_reduce(its, op=dist.ReduceOp.SUM)
        dt = float(np.median(dts))
        out[name] = {"replicas": {"pairs_per_s": n_pairs / dt, "iterations_per_s": float(its.item()) / dt, "ms": dt * 1e3,
                                  "ms_repetitions": [x * 1e3 for x in dts], "n_gpus": world,
                                  "note": "pair k on rank k % W, one align_batch per rank, the slowest rank's wall time (median of three)"}}
    icp_b.close()
    dist.barrier()
    if rank == 0:   # the other ranks have released their batch handles and wait at the barrier below
        devices = list(range(n_dev)) if not args.share_gpu else [0] * world
        all_pairs = [pairs[k] for k in range(n_pairs)]
        try:
            pool = pkg.DevicePool(devices)
            for name, p in (("point_to_point", p2p), ("shipped_loop_closure_yaml", shipped)):
                pool.align_batch(all_pairs[:2 * len(devices)], [np.eye(4)] * min(n_pairs, 2 * len(devices)), p)
                dts = []
                for _ in range(3):
                    t0 = time.perf_counter()
                    res = pool.align_batch(all_pairs, [np.eye(4)] * n_pairs, p)
                    dts.append(time.perf_counter() - t0)
                dt = float(np.median(dts))
                out[name]["device_pool"] = {"pairs_per_s": n_pairs / dt, "iterations_per_s": sum(r.nIterations for r in res) / dt, "ms": dt * 1e3,
                                            "devices": devices, "note": "ONE process, mola_icp_pool_align_batch over every visible device (the other ranks idle)"}
            pool.close()
        except Exception as e:  # noqa: BLE001
            out["device_pool_error"] = str(e)
    dist.barrier()
    return out


def config3_batch(pkg, synth, icp, n_pairs, with_cpu, cpu_flags, shipped=False):
    """configs[3] on ONE GPU: independent 100k x 100k pairs (seeds 100...), <= 100 iterations with the stall test, through
    `align_batch`; and SURVEY §8(d)(ii)'s all-core CPU leg: max(2, nproc/2) threads, one pair each (cpp:94-96).
    shipped=True: the same pairs through params/icp-settings-loop-closure.yaml (Point2Plane + Gauss-Newton)."""
    pairs = [synth.make_pair(100_000, 100_000, seed=100 + s)[:2] for s in range(n_pairs)]
    if shipped:
        p = pkg.Parameters.load_from_file(os.path.join(ROOT, "params", "icp-settings-loop-closure.yaml"))
    else:
        p = pkg.Parameters()
        p.matcher_threshold, p.max_iterations, p.min_abs_step_trans, p.min_abs_step_rot = GATE_M, 100, 5e-5, 1e-5
    # warm-up: the batch scratch of a full 12-problem chunk, and the GPU's clocks (this leg follows seconds of CPU-only work)
    tw = time.perf_counter()
    while time.perf_counter() - tw < 0.8:   # (the GPU has idled through seconds of CPU legs: 570 pairs/s behind a 60-ms warm-up, 900 without the CPU legs)
        icp.align_batch(pairs[:min(n_pairs, 12)], [np.eye(4)] * min(n_pairs, 12), p)
    dts = []
    for _ in range(3):   # (one call is 40-70 ms of mostly host-driven uploads and launches: the median of three)
        t0 = time.perf_counter()
        res = icp.align_batch(pairs, [np.eye(4)] * n_pairs, p)
        dts.append(time.perf_counter() - t0)
    dt = float(np.median(dts))
    its = int(sum(r.nIterations for r in res))
    out = {"workload": f"{n_pairs} independent 100k x 100k pairs (seeds 100..{99 + n_pairs}), "
                       + ("Point2Plane knn 6 + Gauss-Newton (icp-settings-loop-closure.yaml)" if shipped else "point-to-point + Horn") + ", <= 100 its, "
                       "host buffers in (uploads and sorts inside the time)",
           "gpu": {"pairs_per_s": n_pairs / dt, "iterations_per_s": its / dt, "ms": dt * 1e3, "iterations_total": its,
                   "ms_repetitions": [x * 1e3 for x in dts],
                   "n_gpus": 1, "note": "one MI355X, median of three calls; the 8-GPU form deals the pairs round-robin (mola_icp_pool_*), no collective"}}
    if with_cpu:
        from oracle import oracle as O
        n_thr = max(2, (os.cpu_count() or 2) // 2)          # worker_pool_past_KFs_ size, src/LidarOdometry.cpp:94-96
        n_thr = min(n_thr, n_pairs)
        op = O.params_from_product(p)
        O.lib()
        done = [None] * n_thr

        def work(k):
            done[k] = O.align(pairs[k][0], pairs[k][1], np.eye(4), op)   # (ctypes releases the GIL inside the call)
        th = [threading.Thread(target=work, args=(k,)) for k in range(n_thr)]
        t0 = time.perf_counter()
        for t in th:
            t.start()
        for t in th:
            t.join()
        dtc = time.perf_counter() - t0
        itc = int(sum(d["n_iterations"] for d in done))
        out["cpu"] = {"pairs_per_s": n_thr / dtc, "iterations_per_s": itc / dtc, "threads": n_thr, "cores": n_thr, "kind": "port",
                      "host_cores_available": os.cpu_count(), "compiler_flags": cpu_flags,
                      "sample": f"{n_thr} of the pairs, one per thread (kd-tree build included), {dtc:.1f} s wall"}
        rot, trans = _pose_err(res[0].optimal_tf, done[0]["T"])
        out["pose_err_vs_cpu_pair0"] = {"rot_rad": rot, "trans_m": trans}
    return out


if __name__ == "__main__":
    main()
