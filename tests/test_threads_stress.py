"""Host-side concurrency (SURVEY.md section 5 "Race detection / sanitizers"; the reference calls align() on ONE ICP object from
max(2, hw/2) pool threads and the odometry thread at once: src/LidarOdometry.cpp:94-96, 869).  Eight threads drive everything of
the C-ABI that runs without a GPU -- the loops over caller-supplied stages (mola_icp_run_loop, mola_icp_run_loop_batch: the
code behind align / align_batch / align_multi_init), the YAML loaders, validation, the host solvers, the front-end mirror with an
injected align -- at the same time; every thread must see its own results and its own error strings.  tools/sanitize.sh runs this
file (and the whole CPU suite) under AddressSanitizer + UBSan and under ThreadSanitizer builds of the host code.
(The handle's worker pool and the per-workspace lanes need a device and are exercised by the -m gpu tests; sanitizers are not
available on the GPU pool.)"""
import os
import threading

import numpy as np
import pytest

from tests.helpers import OracleStages, p2p_params

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
N_THREADS = 8


def _work(pkg, O, golden, tid, out, errors):
    try:
        rng = np.random.default_rng(100 + tid)
        g, l = golden["A_map"], golden["A_local"]
        sub = np.ascontiguousarray(l[:, tid::N_THREADS][:, :600])
        gs = np.ascontiguousarray(g[:, ::3])
        p = p2p_params(pkg, max_iterations=6 + tid % 3, matcher_threshold=0.8 + 0.05 * tid)
        for rep in range(3):
            # the stand-alone loop and the lockstep batch over oracle stages of this thread's own sub-problem
            st = OracleStages(O, gs, sub)
            r = pkg.run_loop(st.match, st.accumulate, np.eye(4), p, sub.shape[1], gs.shape[1])
            sts = [OracleStages(O, gs, np.ascontiguousarray(sub[:, k::2])) for k in range(2)]
            rb = pkg.run_loop_batch([(s.match, s.accumulate, s.l.shape[1], gs.shape[1]) for s in sts], [np.eye(4)] * 2, p)
            ref = O.align(gs, sub, np.eye(4), O.params_from_product(p))
            assert r.nIterations == ref["n_iterations"] and r.terminationReason == ref["termination"]
            np.testing.assert_allclose(r.optimal_tf, ref["T"], atol=1e-9)
            assert len(rb) == 2 and all(np.all(np.isfinite(x.optimal_tf)) for x in rb)
            # an error of this thread's own making: its message must be this thread's
            bad = p2p_params(pkg)
            bad.matcher_threshold = -1.0 - tid
            with pytest.raises(pkg.IcpError) as ex:
                pkg.run_loop(st.match, st.accumulate, np.eye(4), bad, 10, 10)
            assert "matcher threshold" in str(ex.value)
            with pytest.raises(pkg.IcpError) as ex:
                pkg.Parameters.load_from(open(os.path.join(ROOT, "params", "icp-settings-regular.yaml")).read()
                                         .replace("mp2p_icp::Solver_GaussNewton", "mp2p_icp::Solver_Thread%d" % tid))
            assert "Solver_Thread%d" % tid in str(ex.value)
            # loaders, host math
            q = pkg.Parameters.load_from_file(os.path.join(ROOT, "params", "icp-settings-regular.yaml"))
            assert q.knn == 6
            lp = pkg.LidarOdometryParams.load_from_file(os.path.join(ROOT, "params", "kitti-default.yaml"), ROOT)
            assert lp.min_icp_goodness == pytest.approx(0.5) and lp.loop_closure_montecarlo_samples == 10
            acc = np.zeros(24)
            pts = rng.normal(size=(3, 40))
            T = pkg.pose_from_xyzypr(list(rng.normal(0, 0.1, 6)))
            gg = T[:3, :3] @ pts + T[:3, 3:4]
            acc[0] = acc[16] = 40
            acc[1:4], acc[4:7], acc[7:16] = pts.sum(1), gg.sum(1), (pts @ gg.T).reshape(9)
            np.testing.assert_allclose(pkg.solve_horn(acc), T, atol=1e-10)
            # the front-end mirror with an injected align (no device): a short drive
            def align(frm, to, T0, params):
                return T0, 0.9, 3, pkg.TERM_STALLED
            lo = pkg.LidarOdometry(lp, align_fn=align)
            for k in range(5):
                lo.on_new_observation(10.0 + 0.1 * k, np.ascontiguousarray(sub[:, :50]))
            lo.close()
        out[tid] = r.nIterations
    except BaseException as e:   # noqa: BLE001 -- reported by the main thread
        errors.append((tid, repr(e)))


def test_eight_threads_through_the_host_side_of_the_c_abi(pkg, O, golden):
    out, errors = {}, []
    ths = [threading.Thread(target=_work, args=(pkg, O, golden, t, out, errors)) for t in range(N_THREADS)]
    for t in ths:
        t.start()
    for t in ths:
        t.join()
    assert not errors, errors
    assert len(out) == N_THREADS


def test_local_comm_ranks_as_threads(pkg):
    """the node-local communicator with its ranks as THREADS of one process (each its own handle and mapping of the segment):
    the collective create, 3000 all-reduces of both block lengths in lockstep, rank-ordered sums on every rank -- the form in which
    the sanitizer runs see both sides of the mailbox protocol (csrc/local_comm.cpp)"""
    import importlib
    import os
    import threading
    sharded = importlib.import_module("mola-fe-lidar_amd.sharded")
    world = 4
    name = f"mola_icp_thr_{os.getpid()}_{int.from_bytes(os.urandom(4), 'little'):x}"
    errs, outs = [], [None] * world

    def rank_main(r):
        try:
            c = sharded.LocalComm(name, world, r, timeout_s=30.0)
            acc = []
            for k in range(3000):
                n = 24 if k % 2 else 92
                a = np.full(n, float(r + 1) * (k + 1))
                c.allreduce(a)
                if k % 500 == 0:
                    acc.append(a.copy())
            c.close()
            outs[r] = acc
        except Exception as e:  # noqa: BLE001
            errs.append((r, repr(e)))

    th = [threading.Thread(target=rank_main, args=(r,)) for r in range(world)]
    [t.start() for t in th]
    [t.join(120) for t in th]
    assert not errs, errs
    for r in range(world):
        for i, k in enumerate(range(0, 3000, 500)):
            want = sum(float(q + 1) * (k + 1) for q in range(world))
            assert np.all(outs[r][i] == want)
