"""-m gpu: the host-side policies of round 6 change WHEN a thread learns of a result and WHICH stream carries its work, never the
result: wait policy (mola_icp_set_wait_policy: spin / yield / block) and the calling thread's priority class
(mola_icp_set_thread_priority; the reference's odometry thread beside its pool threads: src/LidarOdometry.cpp:94-96, 183-184, 869)."""
import ctypes as C
import os
import threading

import numpy as np
import pytest

from tests.helpers import p2p_params

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_wait_policies_give_the_same_bits(pkg, synth):
    g, l, _ = synth.make_pair(30_000, 30_000, seed=5)
    shipped = pkg.Parameters.load_from_file(os.path.join(ROOT, "params", "icp-settings-regular.yaml"))
    icp = pkg.ICP(device=0)
    ref = None
    try:
        for policy in ("spin", "yield", "block", "spin"):
            pkg.ICP.set_wait_policy(policy)
            got = C.c_int(-1)
            pkg._lib.check(pkg._lib.lib().mola_icp_get_wait_policy(C.byref(got)))
            assert got.value == {"spin": 0, "yield": 1, "block": 2}[policy]
            a = icp.align(g, l, np.eye(4), p2p_params(pkg, max_iterations=30))
            b = icp.align(g, l, np.eye(4), shipped)
            c = icp.align_batch([(g, l)] * 3, [np.eye(4)] * 3, shipped)
            cur = (a.optimal_tf, a.nIterations, b.optimal_tf, b.nIterations, b.quality, c[2].optimal_tf)
            if ref is None:
                ref = cur
            for x, y in zip(cur, ref):
                assert np.array_equal(x, y)
        with pytest.raises(pkg.IcpError):
            pkg.ICP.set_wait_policy(7)
    finally:
        pkg.ICP.set_wait_policy("spin")
    icp.close()


def test_thread_priority_class_changes_no_result(pkg, synth):
    """a high-priority thread (the odometry step) and normal-priority threads (checks) on ONE handle at the same time: every result
    equals the serial one; the priority is thread-local and the front-end restores the caller's"""
    g, l, _ = synth.make_pair(40_000, 40_000, seed=6)
    p = p2p_params(pkg, max_iterations=40)
    icp = pkg.ICP(device=0)
    serial = icp.align(g, l, np.eye(4), p)
    L = pkg._lib.lib()
    out, prio_seen = {}, {}

    def work(name, high):
        pkg._lib.check(L.mola_icp_set_thread_priority(1 if high else 0))
        rs = [icp.align(g, l, np.eye(4), p) for _ in range(4)]
        v = C.c_int(-1)
        pkg._lib.check(L.mola_icp_get_thread_priority(C.byref(v)))
        out[name], prio_seen[name] = rs, v.value
    th = [threading.Thread(target=work, args=("odo", True))] + [threading.Thread(target=work, args=(f"chk{i}", False)) for i in range(3)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    assert prio_seen["odo"] == 1 and all(prio_seen[f"chk{i}"] == 0 for i in range(3))
    for rs in out.values():
        for r in rs:
            assert np.array_equal(r.optimal_tf, serial.optimal_tf) and r.nIterations == serial.nIterations
    # the front-end raises the class for its own duration only
    lp = pkg.LidarOdometryParams.load_from_file(os.path.join(ROOT, "params", "kitti-default.yaml"), ROOT)
    lo = pkg.LidarOdometry(lp, icp=icp)
    v = C.c_int(-1)
    for k in range(3):
        lo.on_new_observation(10.0 + 0.1 * k, synth.lidar_scan(synth.pose_from_xyzypr(-5.0 + k, 0, 0, 0, 0, 0), n_rings=16, n_az=400, seed=9 + k))
        pkg._lib.check(L.mola_icp_get_thread_priority(C.byref(v)))
        assert v.value == 0
    lo.close()
    icp.close()
