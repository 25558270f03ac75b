#!/usr/bin/env python3
"""bench.py's sharded sequence replayed piece by piece (one rank, node-local mailbox attached): which step makes the FIRST timed stateless align
0.9 ms dearer than every later one?"""
import importlib, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import torch.distributed as dist
pkg = importlib.import_module("mola-fe-lidar_amd")
synth = importlib.import_module("mola-fe-lidar_amd.synth")
variant = sys.argv[1] if len(sys.argv) > 1 else "bench"
N = 1_000_000
g, l, _ = synth.make_pair(N, N, seed=42)
dev = torch.device("cuda", 0)
tg, tl = torch.from_numpy(g).to(dev), torch.from_numpy(np.ascontiguousarray(l)).to(dev)
icp = pkg.ICP(device=0)
icp.set_map(tg); icp.set_local(tl)
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29534")
os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1")
if variant != "noattach":
    dist.init_process_group("nccl", device_id=dev)
    lc = icp.comm_init_local()
p = pkg.Parameters()
p.matcher_threshold, p.fixed_iterations, p.skip_quality = 1.0, 1, 1
T0 = np.eye(4)
def barrier():
    if variant not in ("nobarrier", "noattach"):
        lc.allreduce(np.zeros(1))
    torch.cuda.synchronize()
def timed(tag):
    barrier()
    t0 = time.perf_counter()
    r = icp.align_resident(T0, p)
    torch.cuda.synchronize()
    print(f"{variant:10s} {tag:34s} {(time.perf_counter() - t0) / p.max_iterations * 1e3:.4f} ms/step (loop {r.ms_iterations / p.max_iterations:.4f})", flush=True)
p.max_iterations = 40
for k in range(60):
    if k == 59:
        icp.forget_warm_start()
    icp.align_resident(T0, p)
if variant != "nowarm3":
    p.max_iterations = 3
    icp.align_resident(T0, p)
p.max_iterations = 40
icp.forget_warm_start()
timed("first timed stateless")
timed("warm repeat")
icp.forget_warm_start()
timed("second stateless")
p.max_iterations = 3
icp.align_resident(T0, p)
p.max_iterations = 40
icp.forget_warm_start()
timed("stateless behind a 3-step align")
icp.forget_warm_start()
timed("third stateless")
