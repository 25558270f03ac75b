"""`python bench.py --gpus N` launches its own ranks (SURVEY.md §8 row e: the driver's N>1 command has no launcher in front
of it on some paths).  CPU: the launcher starts N fresh children and reports a failing rank with a non-zero exit.
GPU (one device): two ranks sharing the GPU through the self-launch path, and the one-rank distributed path with the
native RCCL communicator on and off."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SMALL = ["--steps", "4", "--warmup", "1", "--device-warmup-aligns", "1", "--n-local", "60000", "--n-map", "70000",
         "--cpu-baseline-iters", "0", "--shipped-iters", "0", "--dense-iters", "0", "--e2e", "0", "--batch-pairs", "0",
         "--c5-map", "150000", "--c5-steps", "3"]


def _run(extra, timeout=600, overrides=()):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + extra + SMALL + list(overrides), env=env, capture_output=True, text=True,
                          timeout=timeout)


def _json_line(stdout):
    lines = [ln for ln in stdout.strip().splitlines() if ln.startswith("{")]
    assert len(lines) == 1, stdout
    return json.loads(lines[0])


def test_self_launch_reports_a_failing_rank(pkg):
    import ctypes
    n = ctypes.c_int(0)
    pkg._lib.lib().mola_icp_device_count(ctypes.byref(n))
    if n.value > 0:
        pytest.skip("a GPU is present: the ranks would run")
    r = _run(["--gpus", "2"], timeout=300)
    assert r.returncode != 0
    assert "exited with code" in r.stderr and "stopping the other ranks" in r.stderr
    assert not [ln for ln in r.stdout.splitlines() if ln.startswith("{")]   # no JSON line from a failed job


@pytest.mark.gpu
def test_two_self_launched_ranks_share_the_gpu():
    # fewer devices than ranks: the ranks share the GPU, said in the line; the node-local all-reduce.  The N > 1 line is COMPLETE
    # (VERDICT r5 item 2): configs[3] as replicas over the ranks and through the device pool, the CPU leg, both configs[4] regimes
    r = _run(["--gpus", "2"], overrides=["--batch-pairs", "4", "--cpu-baseline-iters", "4"])
    assert r.returncode == 0, r.stderr[-3000:]
    j = _json_line(r.stdout)
    b = j["config3_batch"]
    assert b["n_gpus"] == 2 and b["pairs_per_rank"] == [2, 2]
    for name in ("point_to_point", "shipped_loop_closure_yaml"):
        assert b[name]["replicas"]["n_gpus"] == 2 and b[name]["replicas"]["pairs_per_s"] > 0
        assert b[name]["device_pool"]["pairs_per_s"] > 0 and b[name]["device_pool"]["devices"] == [0, 0]
    assert j["cpu_baseline"]["value"] > 0 and j["cpu_baseline"]["cores"] == 1 and j["cpu_baseline"]["kind"] == "port"
    assert j["pose_err_vs_cpu"]["whole_timed_workload"] is True and j["pose_err_vs_cpu"]["rot_rad"] < 1e-4 and j["pose_err_vs_cpu"]["trans_m"] < 1e-3
    near = j["c5_sharded"]["near_converged"]
    assert near["ms_per_step"] > 0 and near["pose_err_vs_gt"]["trans_m"] < j["c5_sharded"]["pose_err_vs_gt"]["trans_m"]
    assert j["n_gpus"] == 2 and j["steps"] == 4 and j["value"] > 0
    assert j["config"]["ranks_share_gpus"] is True and "query-shard x2, local all-reduce" in j["config"]["parallelism"]
    assert j["config"]["comm_nranks"] == 2        # both ranks joined the mailbox
    # the default line times EVERY transport: both keys are there -- the mailbox with its step time and the two
    # ranks it saw; RCCL's entry is null here and says why (two ranks cannot share one device under RCCL) -- on a node it carries
    # ncclCommCount and its own step time, and `value` is the faster of the two
    ar = j["config"]["allreduce"]
    assert set(ar) == {"local", "rccl"}
    assert ar["local"]["nranks"] == 2 and ar["local"]["ms_per_step"] > 0
    assert ar["rccl"]["nranks"] is None and ar["rccl"]["ms_per_step"] is None and "one device" in ar["rccl"]["note"]
    assert set(j["c5_sharded"]["allreduce"]) == {"local", "rccl"} and j["c5_sharded"]["allreduce"]["local"]["nranks"] == 2
    assert j["value"] == pytest.approx(1e3 / ar["local"]["ms_per_step"], rel=1e-9)
    assert j["config"]["queries_per_gpu"] == 30000 and j["config"]["shard_balance"] is None      # (the headline: equal counts)
    assert j["c5_sharded"]["shard_balance"] is None            # (ranks sharing one device: no cost-balanced re-cut -- they would time each other)
    assert j["config"]["map_slab_rank0"]["margin_m"] > 2.0 and "recut" not in j["config"]["map_slab_rank0"]
    c5 = j["c5_sharded"]                          # configs[4]'s shape beside the headline, slab per rank
    assert c5["n_map"] == 150000 and c5["value"] > 0 and c5["map_slab_rank0"]["map_points_kept"] < 150000
    r1 = _run(["--gpus", "1"])
    j1 = _json_line(r1.stdout)
    # same job: the two-rank pose is the one-rank pose (fp64 sums in another order) -- "to 1e-12" on the new collective
    assert abs(j["pose_err_vs_gt"]["rot_rad"] - j1["pose_err_vs_gt"]["rot_rad"]) < 1e-12
    assert abs(j["pose_err_vs_gt"]["trans_m"] - j1["pose_err_vs_gt"]["trans_m"]) < 1e-12
    c51 = j1["c5_sharded"]
    assert abs(c5["pose_err_vs_gt"]["rot_rad"] - c51["pose_err_vs_gt"]["rot_rad"]) < 1e-12
    assert abs(c5["pose_err_vs_gt"]["trans_m"] - c51["pose_err_vs_gt"]["trans_m"]) < 1e-12


@pytest.mark.gpu
@pytest.mark.parametrize("allreduce", ["local", "rccl", "hook"])
def test_one_rank_distributed_path_over_every_collective(allreduce):
    r = _run(["--gpus", "1", "--force-dist", "--allreduce", allreduce])
    assert r.returncode == 0, r.stderr[-3000:]
    j = _json_line(r.stdout)
    assert j["n_gpus"] == 1
    assert allreduce in j["config"]["parallelism"]
    if allreduce in ("rccl", "local"):
        assert j["config"]["comm_nranks"] == 1     # what ncclCommCount / the mailbox's join count reports
    else:
        assert j["config"]["comm_nranks"] is None
    assert set(j["config"]["allreduce"]) == {allreduce}


@pytest.mark.gpu
def test_one_rank_default_line_carries_both_transports():
    """one rank through the distributed path with the default --allreduce both: the mailbox and RCCL are both timed, each reports one
    rank (ncclCommCount for RCCL), `value` is the faster one's and every entry produced the same pose"""
    r = _run(["--gpus", "1", "--force-dist"])
    assert r.returncode == 0, r.stderr[-3000:]
    j = _json_line(r.stdout)
    ar = j["config"]["allreduce"]
    assert set(ar) == {"local", "rccl"} and ar["local"]["nranks"] == 1 and ar["rccl"]["nranks"] == 1
    best = min(ar, key=lambda k: ar[k]["ms_per_step"])
    assert best in j["config"]["parallelism"] and j["ms_per_step"] == pytest.approx(ar[best]["ms_per_step"], rel=1e-9)
    assert set(j["c5_sharded"]["allreduce"]) == {"local", "rccl"}
