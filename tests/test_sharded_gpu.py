"""SURVEY.md §8 row e on the hardware that is there (one GPU): the shard is cut on the device from the scan's Hilbert
order, a rank keeps only the part of the map its shard can reach, and two ranks -- fresh child processes, gloo, the real
HIP stages, the all-reduce hook -- reproduce the single-process pose."""
import importlib
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def test_device_shard_cut_and_map_slab_single_process(pkg, O, synth):
    g, l, _ = synth.make_pair(100_000, 150_000, seed=31)
    p = pkg.Parameters()
    p.matcher_threshold = 1.0
    full = pkg.ICP(device=0)
    full.set_map(g)
    full.set_local(l)
    idx_full, d2_full, n_full = full.match(np.eye(4), 1.0, l.shape[1])
    acc_full = full.accumulate(p, np.eye(4))
    W = 4
    icp = pkg.ICP(device=0)
    seen = np.zeros(l.shape[1], dtype=np.int32)
    tot = np.zeros(24)
    kept_total = 0
    for r in range(W):
        n = icp.set_local_shard(l, r, W)
        sidx = icp.local_shard_indices()
        assert n == len(sidx) and abs(n - l.shape[1] / W) <= 1
        seen[sidx] += 1
        lo, hi = icp.shard_reach_box(np.eye(4), 1.5)
        # compact: the shard's box is a fraction of the scene (Hilbert slices, not random subsamples)
        pts = l[:, sidx]
        assert np.all(pts.min(axis=1) >= lo + 1.49) and np.all(pts.max(axis=1) <= hi - 1.49)
        kept = icp.set_map_slab(g, lo, hi)
        inside = np.all((g >= lo[:, None].astype(np.float32)) & (g <= hi[:, None].astype(np.float32)), axis=0)
        assert abs(kept - int(inside.sum())) <= 8 and kept < g.shape[1]     # (box edges are rounded outwards in fp32)
        kept_total += kept
        i_r, d_r, n_r = icp.match(np.eye(4), 1.0, n)
        # same pairing as the full map's, bit for bit, in ORIGINAL map indices
        assert np.array_equal(i_r, idx_full[sidx]) and np.array_equal(d_r[i_r >= 0], d2_full[sidx][i_r >= 0])
        tot += icp.accumulate(p, np.eye(4))
    assert np.all(seen == 1)                                       # the shards partition the scan
    np.testing.assert_allclose(tot, acc_full, rtol=1e-12, atol=1e-6)
    assert kept_total < 2.5 * g.shape[1]                           # not W copies of the map
    # a pose that moves the shard out of its slab is refused (single process: the error comes straight back)
    with pytest.raises(pkg.IcpError, match="outside its map slab"):
        icp.match(synth.pose_from_xyzypr(5.0, 0, 0, 0, 0, 0), 1.0, n)
    # and the oracle agrees with a sample of the sharded pairing
    sel = sidx[::50]
    oidx, od2, _ = O.match(g, np.ascontiguousarray(l[:, sel]), np.eye(4), 1.0, O.KdTree(g))
    assert np.array_equal(i_r[::50], oidx)
    icp.close()
    full.close()


@pytest.mark.parametrize("scenario", ["p2p", "p2pl", "recut", "hook", "balance"])
def test_two_ranks_on_one_gpu_equal_single_process(pkg, synth, tmp_path, scenario):
    world = 2
    port = _free_port()
    env = dict(os.environ)
    env["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"
    procs = [subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "mp_sharded_gpu_worker.py"), str(r), str(world), str(port),
                               str(tmp_path), scenario], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
             for r in range(world)]
    outs = []
    for pr in procs:
        try:
            out, _ = pr.communicate(timeout=600)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
        outs.append(out)
    assert all(pr.returncode == 0 for pr in procs), "\n".join(outs)
    g, l, _ = synth.make_pair(60_000, 90_000, seed=23)
    if scenario == "p2pl":
        p = pkg.Parameters.load_from_file(os.path.join(ROOT, "params", "icp-settings-regular.yaml"))
    else:
        p = pkg.Parameters()
        p.matcher_threshold, p.max_iterations, p.min_abs_step_trans, p.min_abs_step_rot = 1.0, 40, 5e-5, 1e-5
    icp = pkg.ICP(device=0)
    ref = icp.align(g, l, np.eye(4), p)
    res = [np.load(os.path.join(tmp_path, f"rank{r}.npz")) for r in range(world)]
    assert np.array_equal(res[0]["T"], res[1]["T"])                # every rank: the identical solve
    for r in res:
        assert int(r["nit"]) == ref.nIterations and int(r["term"]) == ref.terminationReason
        np.testing.assert_allclose(r["T"], ref.optimal_tf, rtol=0, atol=1e-12)
        assert float(r["quality"]) == pytest.approx(ref.quality, abs=1e-12)
        assert int(r["n_map_kept"]) < g.shape[1]
    assert int(res[0]["n_shard"]) + int(res[1]["n_shard"]) == l.shape[1]
    assert len(np.intersect1d(res[0]["shard_idx"], res[1]["shard_idx"])) == 0
    if scenario == "recut":    # the 0.54 m / 2 deg correction does not fit a 1.05 m margin: both ranks cut again, together
        assert all(float(r["margin"]) > float(r["margin0"]) for r in res)
    if scenario == "balance":  # margin from the guess (1 m, 3 deg) and THIS shard's farthest corner; cuts by cost, the same on both ranks
        assert all(2.0 < float(r["margin"]) < 12.0 for r in res)
        assert np.array_equal(res[0]["cuts"], res[1]["cuts"]) and res[0]["cuts"][0] == 0 and res[0]["cuts"][-1] == l.shape[1]
        assert int(res[0]["n_shard"]) == int(res[0]["cuts"][1])
    icp.close()


def test_comm_init_argument_errors(pkg):
    import ctypes as C
    L = pkg._lib
    icp = pkg.ICP(device=0)
    ident = (C.c_uint8 * 128)()
    for nranks, rank in ((0, 0), (2, 2), (2, -1), (1, 1)):
        assert L.lib().mola_icp_comm_init(icp._h, ident, nranks, rank) == L.E_BADARG
        assert b"rank" in L.lib().mola_icp_last_error()
    assert L.lib().mola_icp_comm_init(icp._h, None, 1, 0) == L.E_BADARG
    assert L.lib().mola_icp_comm_init(None, ident, 1, 0) == L.E_BADARG
    assert L.lib().mola_icp_comm_unique_id(None) == L.E_BADARG
    assert L.lib().mola_icp_comm_destroy(icp._h) == 0          # nothing to destroy: fine
    icp.close()


@pytest.mark.gpu
def test_shard_by_range_is_the_rank_slice_and_any_other_slice(pkg, synth):
    """mola_icp_set_local_shard_range_*: [lo, hi) of the scan's Hilbert order -- rank r of W is the range of shard_bounds, and
    uneven cuts partition the scan"""
    sharded = importlib.import_module("mola-fe-lidar_amd.sharded")
    _, l, _ = synth.make_pair(30_011, 1000, seed=3)
    icp = pkg.ICP(device=0)
    n = l.shape[1]
    assert icp.set_local_shard(l, 1, 3) == sharded.shard_bounds(n, 1, 3)[1] - sharded.shard_bounds(n, 1, 3)[0]
    by_rank = icp.local_shard_indices().copy()
    lo, hi = sharded.shard_bounds(n, 1, 3)
    assert icp.set_local_shard_range(l, lo, hi) == hi - lo
    assert np.array_equal(icp.local_shard_indices(), by_rank)
    cuts = [0, 17, 17, 20_000, n]
    seen = np.concatenate([(icp.set_local_shard_range(l, cuts[k], cuts[k + 1]), icp.local_shard_indices()[:cuts[k + 1] - cuts[k]].copy())[1] for k in range(4)])
    assert np.array_equal(np.sort(seen), np.arange(n))
    with pytest.raises(pkg.IcpError):
        icp.set_local_shard_range(l, 5, n + 1)
    icp.close()
