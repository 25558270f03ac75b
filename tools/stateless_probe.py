#!/usr/bin/env python3
"""Where does a STATELESS align (mola_icp_forget_warm_start first) spend more than its warm repeat -- with and without a transport
attached (one rank: the node-local mailbox / the torch hook / RCCL)?  Prints wall ms per step and the matcher's own time per launch."""
import importlib, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
pkg = importlib.import_module("mola-fe-lidar_amd")
synth = importlib.import_module("mola-fe-lidar_amd.synth")
sharded = importlib.import_module("mola-fe-lidar_amd.sharded")
mode = sys.argv[1] if len(sys.argv) > 1 else "plain"
N = 1_000_000
g, l, _ = synth.make_pair(N, N, seed=42)
dev = torch.device("cuda", 0)
tg, tl = torch.from_numpy(g).to(dev), torch.from_numpy(np.ascontiguousarray(l)).to(dev)
icp = pkg.ICP(device=0)
icp.set_map(tg); icp.set_local(tl)
if mode != "plain":
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
    os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1")
    if mode in ("local_gloo", "fn_gloo"):
        dist.init_process_group("gloo")
    else:
        dist.init_process_group("nccl", device_id=dev)
    if mode.startswith("local"):
        lc = icp.comm_init_local()
    elif mode == "rccl":
        icp.comm_init()
    elif mode.startswith("fn"):
        icp.set_allreduce(lambda acc: None)
p = pkg.Parameters()
p.matcher_threshold, p.fixed_iterations, p.skip_quality, p.max_iterations = 1.0, 1, 1, 40
T0 = np.eye(4)
for _ in range(40):
    icp.align_resident(T0, p)
def timed(tag, prep):
    out = []
    for rep in range(4):
        prep()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        r = icp.align_resident(T0, p)
        torch.cuda.synchronize()
        out.append((time.perf_counter() - t0) / 40 * 1e3)
    icp.set_profiling(True)
    prep()
    r = icp.align_resident(T0, p)
    icp.set_profiling(False)
    print(f"{mode:10s} {tag:26s} ms/step " + " ".join("%.4f" % v for v in out) + f" | loop {r.ms_iterations / 40:.4f} ms/step, matcher {r.ms_nn_kernel / max(1, r.n_nn_launches):.4f} ms/launch x {r.n_nn_launches}", flush=True)
timed("warm repeat", lambda: None)
timed("stateless", icp.forget_warm_start)
timed("stateless + schedule gone", lambda: icp.forget_warm_start(schedule=True))
timed("warm repeat again", lambda: None)
