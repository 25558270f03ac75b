"""The node-local communicator (mola_icp_local_comm_*, csrc/local_comm.cpp): the per-iteration all-reduce
of the query-sharded path through a shared-memory mailbox on the host.  CPU, real processes: the sums, their bit-identity across
ranks, the error paths that must not hang (a rank that never comes, ranks that disagree), and the product's loop over it --
equal to the single-process result, like the gloo test beside it."""
import importlib
import multiprocessing as mp
import os
import sys
import time

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _name(tag):
    return f"mola_icp_test_{os.getpid()}_{tag}_{int.from_bytes(os.urandom(4), 'little'):x}"


def _sums_worker(name, rank, world, n_calls, q):
    sys.path.insert(0, ROOT)
    sharded = importlib.import_module("mola-fe-lidar_amd.sharded")
    try:
        c = sharded.LocalComm(name, world, rank, timeout_s=20.0)
        joined = c.nranks()
        out = []
        for k in range(n_calls):
            n = (24, 92, 1, 120)[k % 4]
            rng = np.random.default_rng(1000 * k + rank)
            a = rng.normal(size=n) * 10.0 ** rng.integers(-8, 8, size=n)
            c.allreduce(a)
            if k < 64:
                out.append(a.copy())
        # ... and its latency: back-to-back all-reduces of the 24-double block (every call waits for every rank: the rate of the
        # slowest exchange, the ctypes call included)
        b = np.zeros(24)
        t0 = time.perf_counter()
        for _ in range(20000):
            c.allreduce(b)
        dt = (time.perf_counter() - t0) / 20000
        c.close()
        q.put((rank, joined, out, dt))
    except Exception as e:  # noqa: BLE001
        q.put((rank, -1, repr(e), 0.0))


@pytest.mark.parametrize("world", [1, 2, 4])
def test_sums_in_rank_order_bit_identical_on_every_rank(pkg, world):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    name = _name("sums")
    n_calls = 400
    ps = [ctx.Process(target=_sums_worker, args=(name, r, world, n_calls, q)) for r in range(world)]
    for p in ps:
        p.start()
    got = sorted(q.get(timeout=120) for _ in ps)
    for p in ps:
        p.join(30)
    assert not os.path.exists("/dev/shm/" + name), "the segment's name must be gone once every rank has joined"
    for rank, joined, out, per_call in got:
        assert joined == world, out
    for k in range(64):
        n = (24, 92, 1, 120)[k % 4]
        want = np.zeros(n)
        for r in range(world):   # the contract: rows added in rank order, from zero
            rng = np.random.default_rng(1000 * k + r)
            want = want + rng.normal(size=n) * 10.0 ** rng.integers(-8, 8, size=n)
        for rank, _, out, _ in got:
            assert np.array_equal(out[k], want), (k, rank)
    print(f"[local_comm] world {world}: {max(g[3] for g in got) * 1e6:.2f} us per back-to-back all-reduce of 24 doubles (ctypes call included)")


def _lonely_worker(name, q):
    sys.path.insert(0, ROOT)
    sharded = importlib.import_module("mola-fe-lidar_amd.sharded")
    pkg = importlib.import_module("mola-fe-lidar_amd")
    t0 = time.perf_counter()
    try:
        sharded.LocalComm(name, 2, 0, timeout_s=1.0)
        q.put(("joined", time.perf_counter() - t0))
    except pkg.IcpError as e:
        q.put((str(e), time.perf_counter() - t0))


def test_a_rank_that_never_comes_is_a_comm_error_not_a_hang(pkg):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    name = _name("lonely")
    p = ctx.Process(target=_lonely_worker, args=(name, q))
    p.start()
    msg, dt = q.get(timeout=60)
    p.join(30)
    assert "1 of 2 ranks joined" in msg and dt < 10.0, (msg, dt)
    assert not os.path.exists("/dev/shm/" + name)


def _mismatch_worker(name, rank, mode, q):
    sys.path.insert(0, ROOT)
    sharded = importlib.import_module("mola-fe-lidar_amd.sharded")
    pkg = importlib.import_module("mola-fe-lidar_amd")
    c = sharded.LocalComm(name, 2, rank, timeout_s=3.0)
    try:
        a = np.ones(24)
        c.allreduce(a)
        if mode == "length":
            c.allreduce(np.ones(24 if rank == 0 else 92))
        elif mode == "missing" and rank == 0:
            c.allreduce(np.ones(24))     # rank 1 makes no second call
        elif mode == "abort":
            if rank == 1:
                time.sleep(0.5)          # (rank 0 has read the first all-reduce's rows by now: abort voids this rank's rows)
                c.abort()                # "I cannot go on": rank 0 must not wait for the time-out
            else:
                t0 = time.perf_counter()
                try:
                    c.allreduce(np.ones(24))
                finally:
                    q.put(("t", rank, time.perf_counter() - t0))
        if mode == "missing" and rank == 1:
            time.sleep(4.0)              # (stay alive past rank 0's time-out)
        q.put(("ok", rank, ""))
    except pkg.IcpError as e:
        again = ""
        t0 = time.perf_counter()
        try:                                 # the communicator is finished: the next call fails AT ONCE and says what to do
            c.allreduce(np.ones(24))
        except pkg.IcpError as e2:
            again = f" || again after {time.perf_counter() - t0:.3f} s: {e2}"
        q.put(("err", rank, str(e) + again))
    finally:
        c.close()


@pytest.mark.parametrize("mode", ["length", "missing", "abort"])
def test_ranks_that_disagree_fail_on_every_rank(pkg, mode):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    name = _name(mode)
    ps = [ctx.Process(target=_mismatch_worker, args=(name, r, mode, q)) for r in range(2)]
    for p in ps:
        p.start()
    msgs = [q.get(timeout=60) for _ in range(3 if mode == "abort" else 2)]
    for p in ps:
        p.join(60)
    res = {r: (kind, m) for kind, r, m in msgs if kind in ("ok", "err")}
    if mode == "length":
        assert res[0][0] == "err" and res[1][0] == "err", res
        assert "reduces" in res[0][1] or "gave up" in res[0][1]
    elif mode == "missing":
        assert res[0][0] == "err" and "did not reach all-reduce 2" in res[0][1], res
        assert "destroy the communicator" in res[0][1].split("||")[1] and float(res[0][1].split("again after ")[1].split(" s")[0]) < 0.5, res
    else:
        assert res[0][0] == "err" and "gave up" in res[0][1], res
        waited = [m for kind, r, m in msgs if kind == "t"][0]
        assert waited < 1.5, waited      # at once, not after the 3 s time-out


def _loop_worker(name, rank, world, outdir):
    sys.path.insert(0, ROOT)
    from oracle import oracle as O
    from tests.helpers import OracleStages, p2p_params
    pkg = importlib.import_module("mola-fe-lidar_amd")
    sharded = importlib.import_module("mola-fe-lidar_amd.sharded")
    c = sharded.LocalComm(name, world, rank, timeout_s=30.0)
    gold = np.load(os.path.join(ROOT, "tests", "golden", "icp_golden.npz"))
    g, l = gold["A_map"], gold["A_local"]
    lo, hi = sharded.shard_bounds(l.shape[1], rank, world)
    st = OracleStages(O, g, np.ascontiguousarray(l[:, lo:hi]))
    p = p2p_params(pkg, max_iterations=30)
    r = pkg.run_loop(st.match, st.accumulate, np.eye(4), p, l.shape[1], g.shape[1], c.allreduce)
    np.savez(os.path.join(outdir, f"rank{rank}.npz"), T=r.optimal_tf, nit=r.nIterations, quality=r.quality, n_pairs=r.n_pairs)
    c.close()


def test_product_loop_over_the_local_communicator_equals_single_process(pkg, O, golden, tmp_path):
    """three ranks (an uneven split), the product's loop + per-rank oracle stages: all ranks hold the SAME pose bit for bit (rank-ordered
    sums), equal to the single-process run up to the fp64 regrouping of the sums"""
    from tests.helpers import OracleStages, p2p_params
    world = 3
    ctx = mp.get_context("spawn")
    name = _name("loop")
    ps = [ctx.Process(target=_loop_worker, args=(name, r, world, str(tmp_path))) for r in range(world)]
    for p in ps:
        p.start()
    for p in ps:
        p.join(120)
        assert p.exitcode == 0
    g, l = golden["A_map"], golden["A_local"]
    st = OracleStages(O, g, l)
    single = pkg.run_loop(st.match, st.accumulate, np.eye(4), p2p_params(pkg, max_iterations=30), l.shape[1], g.shape[1], None)
    rs = [np.load(os.path.join(str(tmp_path), f"rank{r}.npz")) for r in range(world)]
    for r in rs[1:]:
        assert np.array_equal(r["T"], rs[0]["T"]) and int(r["nit"]) == int(rs[0]["nit"])
    assert int(rs[0]["nit"]) == single.nIterations and int(rs[0]["n_pairs"]) == single.n_pairs
    np.testing.assert_allclose(rs[0]["T"], single.optimal_tf, atol=1e-12)
    assert float(rs[0]["quality"]) == pytest.approx(single.quality, abs=1e-12)


def _leftover_worker(name, rank, delay, q):
    sys.path.insert(0, ROOT)
    sharded = importlib.import_module("mola-fe-lidar_amd.sharded")
    try:
        time.sleep(delay)
        c = sharded.LocalComm(name, 2, rank, timeout_s=15.0)
        a = np.full(24, float(rank + 1))
        c.allreduce(a)
        c.close()
        q.put((rank, a.tolist()))
    except Exception as e:  # noqa: BLE001
        q.put((rank, repr(e)))


def test_leftover_segment_of_a_crashed_init_is_not_joined(pkg):
    """the name still leads to the segment of a run that crashed inside create() -- right size, magic, rank count, even a
    complete stale `joined` count.  Rank 1 comes first and maps it; rank 0 then replaces it.  Rank 1 must notice (a leftover never
    carries rank 0's acknowledgement, and the name now leads elsewhere) and join the live segment: both ranks get the sum."""
    import struct
    name = _name("leftover")
    nranks = 2
    size = 64 + 1024 * 2 * nranks
    blob = bytearray(size)
    # Header: magic u64, nranks u32, joined u32, left u32, ack u32 (csrc/local_comm.cpp)
    struct.pack_into("<QIIII", blob, 0, 0x4d4f4c414c434f4d, nranks, nranks, 0, 0)
    with open("/dev/shm/" + name, "wb") as f:
        f.write(blob)
    try:
        ctx = mp.get_context("spawn")
        q = ctx.Queue()
        ps = [ctx.Process(target=_leftover_worker, args=(name, 1, 0.0, q)), ctx.Process(target=_leftover_worker, args=(name, 0, 1.0, q))]
        for p in ps:
            p.start()
        got = dict(q.get(timeout=60) for _ in ps)
        for p in ps:
            p.join(30)
        assert got[0] == [3.0] * 24 and got[1] == [3.0] * 24, got
        assert not os.path.exists("/dev/shm/" + name)
    finally:
        if os.path.exists("/dev/shm/" + name):
            os.unlink("/dev/shm/" + name)


def test_leftover_segment_of_another_rank_count_is_not_fatal(pkg):
    """the leftover is that of a crashed run with ANOTHER rank count (4 where this run has 2).  Rank 1, first to come, must not
    fail on the count -- only a live segment (rank 0's acknowledgement) can make that verdict -- but wait for rank 0 to replace it."""
    import struct
    name = _name("leftover4")
    blob = bytearray(64 + 1024 * 2 * 4)
    struct.pack_into("<QIIII", blob, 0, 0x4d4f4c414c434f4d, 4, 4, 0, 0)
    with open("/dev/shm/" + name, "wb") as f:
        f.write(blob)
    try:
        ctx = mp.get_context("spawn")
        q = ctx.Queue()
        ps = [ctx.Process(target=_leftover_worker, args=(name, 1, 0.0, q)), ctx.Process(target=_leftover_worker, args=(name, 0, 1.0, q))]
        for p in ps:
            p.start()
        got = dict(q.get(timeout=60) for _ in ps)
        for p in ps:
            p.join(30)
        assert got[0] == [3.0] * 24 and got[1] == [3.0] * 24, got
        assert not os.path.exists("/dev/shm/" + name)
    finally:
        if os.path.exists("/dev/shm/" + name):
            os.unlink("/dev/shm/" + name)


def test_a_bad_argument_does_not_finish_the_communicator(pkg):
    """(MOLA_ICP_E_COMM does: test_ranks_that_disagree_fail_on_every_rank[missing])"""
    sharded = importlib.import_module("mola-fe-lidar_amd.sharded")
    c = sharded.LocalComm(_name("broken"), 1, 0, timeout_s=1.0)
    c.allreduce(np.ones(24))
    with pytest.raises(pkg.IcpError):
        c.allreduce(np.ones(121))          # a bad argument is not a communication failure ...
    c.allreduce(np.ones(24))               # ... and leaves the communicator usable
    c.close()
