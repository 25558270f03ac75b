"""-m gpu: device errors surface as status codes and leave the handle usable ("GPU errors (OOM, hipError)
must surface as exceptions, never abort"; the reference's task-level catch: src/LidarOdometry.cpp:510-513, 845-848)."""
import numpy as np
import pytest

from tests.helpers import p2p_params

pytestmark = pytest.mark.gpu


def _fill_device(torch, keep_free_bytes):
    """occupy the device's free memory down to < `keep_free_bytes`: one large torch block, then blocks of that size until the allocator
    refuses (what hipMemGetInfo calls free and what hipMalloc will actually hand out differ by a reserve).  Released by the caller."""
    torch.cuda.empty_cache()
    free, _ = torch.cuda.mem_get_info(0)
    hogs = [torch.empty(max(1, free - 8 * keep_free_bytes), dtype=torch.uint8, device="cuda:0")]
    for _ in range(4096):
        try:
            hogs.append(torch.empty(keep_free_bytes, dtype=torch.uint8, device="cuda:0"))
        except RuntimeError:   # torch.cuda.OutOfMemoryError
            break
    else:
        raise AssertionError("the device never ran out of memory")
    return hogs


def _small_pair_is_right(pkg, O, synth, icp):
    g, l, _ = synth.make_pair(6000, 5000, seed=3)
    p = p2p_params(pkg, max_iterations=30)
    r = icp.align(g, l, np.eye(4), p)
    ref = O.align(g, l, np.eye(4), O.params_from_product(p))
    rot, trans = O.pose_error(r.optimal_tf, ref["T"])
    assert r.nIterations == ref["n_iterations"] and r.terminationReason == ref["termination"]
    assert rot < 1e-7 and trans < 1e-7


def test_cloud_put_that_does_not_fit_returns_oom_and_the_handle_survives(pkg, O, synth):
    import torch
    icp = pkg.ICP(device=0)
    pkg.ICP.device_pool_trim(0, 0)                       # (parked blocks of earlier tests would serve the request)
    small = synth.make_pair(20_000, 20_000, seed=1)[0]
    icp.cloud_put(7, small)
    before = icp.cloud_count()
    big = synth.make_pair(1000, 3_000_000, seed=2)[0]    # 36 MB as given, ~100 MB prepared
    hog = _fill_device(torch, 16 << 20)
    try:
        with pytest.raises(pkg.IcpError) as e:
            icp.cloud_put(8, big)
        assert e.value.status == pkg._lib.E_OOM, e.value
        assert icp.cloud_count() == before               # nothing half-registered
        with pytest.raises(pkg.IcpError):
            icp.align_cached(8, 7, np.eye(4), p2p_params(pkg))   # the id that failed does not exist
    finally:
        del hog
        torch.cuda.empty_cache()
    # the same handle, memory back: the cloud goes in, a small pair aligns as the checker does
    icp.cloud_put(8, big)
    assert icp.cloud_count()[0] == before[0] + 1
    icp.cloud_drop(8)
    _small_pair_is_right(pkg, O, synth, icp)
    icp.close()


def test_set_map_that_does_not_fit_returns_oom_and_the_handle_survives(pkg, O, synth):
    import torch
    icp = pkg.ICP(device=0)
    pkg.ICP.device_pool_trim(0, 0)
    g, l, _ = synth.make_pair(50_000, 3_000_000, seed=4)
    hog = _fill_device(torch, 16 << 20)
    try:
        with pytest.raises(pkg.IcpError) as e:
            icp.align(g, l, np.eye(4), p2p_params(pkg, max_iterations=3))
        assert e.value.status == pkg._lib.E_OOM, e.value
    finally:
        del hog
        torch.cuda.empty_cache()
    r = icp.align(g, l, np.eye(4), p2p_params(pkg, max_iterations=3, fixed_iterations=1))
    assert r.nIterations == 3
    _small_pair_is_right(pkg, O, synth, icp)
    icp.close()
