"""One rank of the two-rank GPU test (tests/test_sharded_gpu.py): started as a FRESH process per rank (never re-exec a
process that has touched the GPU), gloo process group, the product's HIP stages on this rank's shard + map slab, the
accumulator all-reduce through the host hook.  argv: rank world port outdir scenario"""
import importlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    rank, world, port, outdir, scenario = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), sys.argv[4], sys.argv[5]
    import torch.distributed as dist
    pkg = importlib.import_module("mola-fe-lidar_amd")
    synth = importlib.import_module("mola-fe-lidar_amd.sharded")
    sharded = synth
    synth = importlib.import_module("mola-fe-lidar_amd.synth")
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    try:
        g, l, Tgt = synth.make_pair(60_000, 90_000, seed=23)
        icp = pkg.ICP(device=0)
        s = sharded.ShardedICP(icp)
        if scenario == "p2p":
            p = pkg.Parameters()
            p.matcher_threshold, p.max_iterations, p.min_abs_step_trans, p.min_abs_step_rot = 1.0, 40, 5e-5, 1e-5
            margin = 4.5
        elif scenario == "p2pl":
            p = pkg.Parameters.load_from_file(os.path.join(ROOT, "params", "icp-settings-regular.yaml"))
            margin = 4.5
        else:  # "recut": a slab too tight for the pose correction -> every rank fails together, cuts again, succeeds
            p = pkg.Parameters()
            p.matcher_threshold, p.max_iterations, p.min_abs_step_trans, p.min_abs_step_rot = 1.0, 40, 5e-5, 1e-5
            margin = 1.05
        s.set_clouds(g, l, init_guess=np.eye(4), slab_margin=margin)
        m0 = s._slab_margin
        r = s.align(np.eye(4), p)
        np.savez(os.path.join(outdir, f"rank{rank}.npz"), T=r.optimal_tf, nit=r.nIterations, term=r.terminationReason,
                 quality=r.quality, n_pairs=r.n_pairs, n_shard=icp._shard_n, n_map_kept=s.n_map_kept, margin0=m0,
                 margin=s._slab_margin, shard_idx=icp.local_shard_indices())
        icp.close()
    finally:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
