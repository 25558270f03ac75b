"""-m gpu: k_nn_q4 (four lanes per query, csrc/kernels_q4.hpp) on the paths the general suites reach rarely: the map's upper box levels
read from GLOBAL memory (a map too large for the LDS copy, several top boxes), lists that overflow (an unseeded launch with a gate many
tiles wide: the tile list is flushed and resumed, the super-tile list too), ragged tails (N not a multiple of 16 or 64), the exact-tie
redo at its own sizes, and the switch itself (MOLA_ICP_Q4=0|1: the same bits from the kernel it replaces)."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _reload(pkg, **env):
    for k, v in env.items():
        if v is None:
            os.environ.pop(k, None)
        else:
            os.environ[k] = v
    pkg._lib.lib().mola_icp_debug_reload_env()


def _sample_equals_oracle(O, g, l, T, gate, idx, d2, step):
    sel = np.arange(0, l.shape[1], step)
    oidx, od2, _ = O.match(g, np.ascontiguousarray(l[:, sel]), T, gate, O.KdTree(g))
    assert np.array_equal(idx[sel], oidx)
    k = oidx >= 0
    assert np.array_equal(d2[sel][k], od2[k])


def test_large_map_box_levels_from_global_memory(pkg, O, synth):
    """20 011 queries against a 1.2M-point map: ten top boxes, 586 super-tiles -- 14 KB of box levels, more than k_nn_q4 keeps in LDS"""
    g, l, _ = synth.make_pair(20_011, 1_200_000, seed=21)
    T = synth.pose_from_xyzypr(0.04, -0.03, 0.01, 0.004, 0.0, -0.001)
    icp = pkg.ICP(device=0)
    icp.set_map(g)
    icp.set_local(l)
    try:
        _reload(pkg, MOLA_ICP_Q4="1")
        idx0, d20, n0 = icp.match(T, 1.0, l.shape[1], pkg.NN_TILED)          # unseeded
        icp.match(np.eye(4), 1.0, l.shape[1], pkg.NN_TILED)
        idx1, d21, n1 = icp.match(T, 1.0, l.shape[1], pkg.NN_TILED)          # seeded from another pose
        _reload(pkg, MOLA_ICP_Q4="0")
        idx2, d22, n2 = icp.match(T, 1.0, l.shape[1], pkg.NN_TILED)          # the kernel it replaces
    finally:
        _reload(pkg, MOLA_ICP_Q4=None)
    assert n0 == n1 == n2 and np.array_equal(idx0, idx1) and np.array_equal(d20, d21)
    assert np.array_equal(idx0, idx2) and np.array_equal(d20, d22)
    _sample_equals_oracle(O, g, l, T, 1.0, idx0, d20, 7)
    icp.close()


@pytest.mark.parametrize("n", [8192, 8207, 9001, 12_345])
def test_wide_gate_overflows_the_lists_and_ragged_tails(pkg, O, synth, n):
    """an unseeded launch with a 6 m gate on a dense map: every 16-query wave reaches hundreds of tiles (its 64-entry tile list is
    flushed and resumed many times); N = 8207, 9001, 12 345: the last wave / workgroup is partly or wholly padding"""
    scene = synth.Scene(scene_seed=4, half=14.0, wall_y=5.0, wall_h=4.0, n_boxes=6)
    g, l, _ = synth.make_pair(n, 400_000, seed=5, scene=scene)
    icp = pkg.ICP(device=0)
    icp.set_map(g)
    icp.set_local(l)
    try:
        _reload(pkg, MOLA_ICP_Q4="1")
        idx, d2, cnt = icp.match(np.eye(4), 6.0, n, pkg.NN_TILED)
        idx_s, d2_s, cnt_s = icp.match(np.eye(4), 6.0, n, pkg.NN_TILED)     # seeded by itself: idempotent
    finally:
        _reload(pkg, MOLA_ICP_Q4=None)
    assert cnt == cnt_s and np.array_equal(idx, idx_s) and np.array_equal(d2, d2_s)
    _sample_equals_oracle(O, g, l, np.eye(4), 6.0, idx, d2, 3)
    # the accumulators of the fused rows: rel 1e-10 of the oracle's sums over the same pairing (the row of a ragged tail included)
    p = pkg.Parameters()
    p.matcher_threshold = 6.0
    acc = icp.accumulate(p, np.eye(4))
    oidx, od2, _ = O.match(g, l, np.eye(4), 6.0, O.KdTree(g))
    ref = O.accumulate(g, l, oidx, od2, O.params_from_product(p), np.eye(4), 0, None, None, np.zeros(n, np.uint8))
    np.testing.assert_allclose(acc, ref, rtol=1e-10, atol=1e-6)
    icp.close()


def test_exact_ties_at_matcher_sizes(pkg, O):
    """a lattice map with every point duplicated, queries at cell centres: 16-way exact distance ties in every wave -- the packed-key
    redo of k_nn_q4 must return the lowest ORIGINAL index, as the oracle does"""
    ax = np.arange(24, dtype=np.float32)
    cell = np.stack(np.meshgrid(ax, ax, ax, indexing="ij")).reshape(3, -1)
    g = np.ascontiguousarray(np.concatenate([cell, cell], axis=1))                       # 27 648 points, each twice
    l = np.ascontiguousarray((cell[:, :9000] + np.float32(0.5)).astype(np.float32))      # 9 000 queries, 8 nearest lattice points x 2
    icp = pkg.ICP(device=0)
    icp.set_map(g)
    icp.set_local(l)
    try:
        _reload(pkg, MOLA_ICP_Q4="1")
        idx, d2, n = icp.match(np.eye(4), 2.0, l.shape[1], pkg.NN_TILED)
        idx_s, d2_s, _ = icp.match(np.eye(4), 2.0, l.shape[1], pkg.NN_TILED)   # seeded: the tie is met again
    finally:
        _reload(pkg, MOLA_ICP_Q4=None)
    oidx, od2, on = O.match(g, l, np.eye(4), 2.0, O.KdTree(g))
    assert n == on and np.array_equal(idx, oidx) and np.array_equal(d2, od2)
    assert np.array_equal(idx_s, oidx) and np.array_equal(d2_s, od2)
    icp.close()


def test_the_switch_changes_no_align(pkg, O, synth):
    """point-to-point aligns (Horn, the weighted passes too) and a lockstep batch through k_nn_q4 and through the kernels it replaces:
    the same bits -- the rows of unit-weight sums are formed in the same order by every matcher"""
    g, l, _ = synth.make_pair(60_000, 50_000, seed=9)
    outs = {}
    try:
        for q in ("1", "0"):
            _reload(pkg, MOLA_ICP_Q4=q)
            icp = pkg.ICP(device=0)
            p = pkg.Parameters()
            p.matcher_threshold, p.max_iterations = 1.0, 40
            a = icp.align(g, l, np.eye(4), p)
            p.use_scale_outlier_detector, p.scale_outlier_threshold = 1, 1.1
            b = icp.align(g, l, np.eye(4), p)
            c = icp.align_batch([(g, l), (g[:, :30_000], l[:, :20_000])], [np.eye(4)] * 2, p)
            outs[q] = (a.optimal_tf, a.nIterations, a.quality, b.optimal_tf, b.nIterations, c[0].optimal_tf, c[1].optimal_tf, c[1].quality)
            icp.close()
    finally:
        _reload(pkg, MOLA_ICP_Q4=None)
    for x, y in zip(outs["1"], outs["0"]):
        assert np.array_equal(x, y)
