"""The prepare chain (mola-fe-lidar_amd/csrc/map_sort.hip; the role of the kd-tree build in the reference: a new global cloud
every scan, src/LidarOdometry.cpp:215-234, 279): Hilbert keys, the hand-written stable sort (runs of 2048 through a bitonic
network + 8-way merges by ranking), the stable compaction, the fused box levels.  CPU: the library carries no rocPRIM / hipCUB
code any more and the sort's logic equals std::stable_sort (tests/hosts/sort_net_test.cpp walks the device functions thread by
thread).  GPU: the order the device produces equals a numpy restatement of the same fp32 key arithmetic + a stable argsort, at
sizes on both sides of every run / merge-level boundary."""
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "mola-fe-lidar_amd", "lib", "libmola_icp_amd.so")


def test_library_has_no_rocprim_or_hipcub_code(pkg):
    """the sort / select / scan of the prepare chain are hand-written -- neither host symbols nor device
    kernels of the CUB-compat libraries are left in the shared library (kernel names live in the embedded code object)."""
    blob = open(LIB, "rb").read()
    for needle in (b"rocprim", b"hipcub", b"ROCPRIM"):
        assert needle not in blob, needle


def test_sort_network_logic_on_cpu():
    """the device's own functions (sort_net.hpp: register-group schedule, compare-exchange stages, LDS slots, merge ranks),
    run thread by thread by a g++ host against std::stable_sort"""
    out = os.path.join(ROOT, "tests", "hosts", "_build")
    os.makedirs(out, exist_ok=True)
    exe = os.path.join(out, "sort_net_test")
    r = subprocess.run(["g++", "-std=c++17", "-O2", "-Wall", "-Wextra", "-Werror", "-I", os.path.join(ROOT, "mola-fe-lidar_amd", "csrc"),
                        os.path.join(ROOT, "tests", "hosts", "sort_net_test.cpp"), "-o", exe], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    r = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "PASSED" in r.stdout and "FAIL" not in r.stdout, r.stdout[-2000:]


# ---- numpy restatement of the keys (map_sort.hip: HilbertKeys::key, hilbert30) ------------------------------------------
def _spread10(v):
    v = v & np.uint32(0x3ff)
    v = (v | (v << np.uint32(16))) & np.uint32(0x030000ff)
    v = (v | (v << np.uint32(8))) & np.uint32(0x0300f00f)
    v = (v | (v << np.uint32(4))) & np.uint32(0x030c30c3)
    v = (v | (v << np.uint32(2))) & np.uint32(0x09249249)
    return v


def hilbert_keys(pc):
    """30-bit keys of a [3, n] fp32 cloud, the device's arithmetic step by step (fp32, truncation)"""
    pc = np.asarray(pc, np.float32)
    lo, hi = pc.min(1), pc.max(1)
    ext = np.float32(max(np.float32(hi[k] - lo[k]) for k in range(3)))
    scale = np.float32(1023.999) / ext if ext > 0 else np.float32(0)
    X = []
    for k in range(3):
        v = ((pc[k] - lo[k]).astype(np.float32) * scale).astype(np.float32)
        X.append(np.minimum(np.maximum(v, np.float32(0)), np.float32(1023)).astype(np.uint32))
    Q = np.uint32(1 << 9)
    while Q > 1:
        P = np.uint32(Q - 1)
        for i in range(3):
            hit = (X[i] & Q) != 0
            t = (X[0] ^ X[i]) & P
            x0_hit = X[0] ^ P
            x0_else = X[0] ^ t
            xi_else = X[i] ^ t
            X[i] = np.where(hit, X[i], xi_else) if i != 0 else X[i]
            X[0] = np.where(hit, x0_hit, x0_else)
        Q = np.uint32(Q >> 1)
    X[1] = X[1] ^ X[0]
    X[2] = X[2] ^ X[1]
    t = np.zeros_like(X[0])
    Q = np.uint32(1 << 9)
    while Q > 1:
        t = np.where((X[2] & Q) != 0, t ^ np.uint32(Q - 1), t)
        Q = np.uint32(Q >> 1)
    X = [x ^ t for x in X]
    return (_spread10(X[0]) << np.uint32(2)) | (_spread10(X[1]) << np.uint32(1)) | _spread10(X[2])


def test_hilbert_keys_restatement_is_a_bijection_on_a_small_grid():
    """sanity of the restatement itself: 8 x 8 x 8 cells spread over the 10-bit grid get 512 distinct keys, consecutive keys
    belong to neighbouring cells (the curve is continuous)"""
    g = np.stack(np.meshgrid(np.arange(8), np.arange(8), np.arange(8), indexing="ij")).reshape(3, -1).astype(np.float32)
    k = hilbert_keys(g)
    assert len(np.unique(k)) == 512
    order = np.argsort(k, kind="stable")
    steps = np.abs(np.diff(g[:, order], axis=1)).sum(0)
    assert np.all(steps == 1.0)


def _cloud(n, seed, dup=False):
    rng = np.random.default_rng(seed)
    pc = (rng.random((3, n), dtype=np.float32) * np.array([[120.0], [40.0], [8.0]], np.float32)
          - np.array([[60.0], [20.0], [1.0]], np.float32)).astype(np.float32)
    if dup and n > 16:   # duplicated points and one dense cell: long runs of EQUAL keys, where only a stable sort agrees
        pc[:, n // 2:] = pc[:, : n - n // 2]
        pc[:, : n // 8] = (pc[:, :1] + rng.random((3, n // 8), dtype=np.float32) * np.float32(1e-3)).astype(np.float32)
    return pc


@pytest.mark.gpu
@pytest.mark.parametrize("n,dup", [(1, False), (63, False), (4096, False), (4097, True), (50000, False), (120000, True),
                                   (131072, False), (131073, True), (300000, False), (1000003, True)])
def test_device_hilbert_order_equals_stable_argsort(pkg, n, dup):
    """rank 0 of 1 keeps the whole scan in the device's Hilbert order: its indices ARE the sort's permutation"""
    pc = _cloud(n, 1000 + n, dup)
    icp = pkg.ICP(device=0)
    assert icp.set_local_shard(pc, 0, 1) == n
    got = icp.local_shard_indices()
    icp.close()
    want = np.argsort(hilbert_keys(pc), kind="stable").astype(np.int32)
    assert np.array_equal(got, want)


@pytest.mark.gpu
def test_prepared_clouds_match_through_every_kernel(pkg, synth, O):
    """the sorted clouds + box levels the new chain builds serve the tiled matcher: pairings equal to the oracle's at a size with
    two merge levels and a padded last super-tile (any defect in the permutation, the padding or a box shows up as a wrong or
    missing neighbour)"""
    g, l, _ = synth.make_pair(150001, 140003, seed=77)
    icp = pkg.ICP(device=0)
    icp.set_map(g)
    icp.set_local(l)
    idx, d2, n = icp.match(np.eye(4), 1.0, l.shape[1], pkg.NN_TILED)
    icp.close()
    oidx, od2, on = O.match(g, l, np.eye(4), 1.0, O.KdTree(g))
    assert n == on and np.array_equal(idx, oidx) and np.array_equal(d2[idx >= 0], od2[oidx >= 0])


@pytest.mark.gpu
def test_voxel_filter_two_pass_sort_at_two_merge_levels(pkg, synth):
    """the voxel filter's 63-bit keys go through two stable passes of the 32-bit sort; 200k points = two merge levels"""
    g, _, _ = synth.make_pair(10, 200000, seed=19)
    icp = pkg.ICP(device=0)
    for voxel in (0.2, 2.0):
        out = icp.voxel_downsample(g, voxel)
        o = g.min(1)
        inv = np.float32(1.0) / np.float32(voxel)
        ijk = np.floor(((g - o[:, None]).astype(np.float32) * inv).astype(np.float32)).astype(np.int64)
        key = (ijk[0] << 42) | (ijk[1] << 21) | ijk[2]
        order = np.argsort(key, kind="stable")
        ks = key[order]
        heads = np.nonzero(np.r_[True, ks[1:] != ks[:-1]])[0]
        cnt = np.diff(np.r_[heads, len(ks)])
        ref = np.stack([np.add.reduceat(g[a][order].astype(np.float64), heads) / cnt for a in range(3)]).astype(np.float32)
        assert out.shape == ref.shape
        np.testing.assert_allclose(out, ref, rtol=0, atol=2e-6)
    icp.close()


@pytest.mark.gpu
def test_map_slab_compaction_is_stable(pkg):
    """row e: the points inside a box, in ascending original order (count / scan / scatter), at a size of many workgroups"""
    pc = _cloud(500000, 5)
    lo, hi = np.array([-10.0, -5.0, 0.0]), np.array([25.0, 12.0, 4.0])
    icp = pkg.ICP(device=0)
    kept = icp.set_map_slab(pc, lo, hi)
    flo, fhi = lo.astype(np.float32), hi.astype(np.float32)
    inside = np.all((pc >= flo[:, None]) & (pc <= fhi[:, None]), axis=0)
    assert kept == int(inside.sum())
    # the slab's points pair with themselves at distance 0, under their ORIGINAL indices (queries 1 m inside the box: the
    # library refuses a pose whose reach, gate included, leaves the slab)
    inner = np.all((pc >= (flo + 1)[:, None]) & (pc <= (fhi - 1)[:, None]), axis=0)
    sel = np.nonzero(inner)[0][::53]
    q = np.ascontiguousarray(pc[:, sel])
    icp.set_local(q)
    idx, d2, n = icp.match(np.eye(4), 0.5, q.shape[1], pkg.NN_TILED)
    icp.close()
    assert n == q.shape[1] and np.all(d2 == 0)
    assert np.array_equal(idx, sel.astype(np.int32))
