"""Shared test helpers: oracle-backed stages for the product's host loop."""
import numpy as np


class OracleStages:
    """match/accumulate stages computed by the CPU oracle over (a shard of) the local cloud.
    Plugged into the PRODUCT's loop (mola_icp_run_loop) to test its host logic without a GPU."""

    def __init__(self, O, map_pc, local_pc):
        self.O, self.g, self.l = O, map_pc, local_pc
        self.tree = O.KdTree(map_pc) if map_pc.shape[1] else None
        self.idx = self.d2 = self.outlier = None
        self.n_match = 0

    def match(self, T, thr):
        self.n_match += 1
        if self.l.shape[1] == 0 or self.g.shape[1] == 0:
            self.idx = np.full(self.l.shape[1], -1, np.int32)
            self.d2 = np.zeros(self.l.shape[1], np.float32)
            return 0
        self.idx, self.d2, n = self.O.match(self.g, self.l, T, thr, self.tree)
        return n

    def accumulate(self, p, T, stage, cl, cg, reset):
        if reset or self.outlier is None:
            self.outlier = np.zeros(max(1, self.l.shape[1]), np.uint8)
        if self.l.shape[1] == 0:
            return np.zeros(24)
        return self.O.accumulate(self.g, self.l, self.idx, self.d2, self.O.params_from_product(p), T, stage, cl, cg,
                                 self.outlier)


def p2p_params(pkg, **kw):
    p = pkg.Parameters()
    p.max_iterations = 40
    p.min_abs_step_trans = 5e-5
    p.min_abs_step_rot = 1e-5
    p.matcher_threshold = 1.0
    p.quality_threshold = 0.10
    for k, v in kw.items():
        setattr(p, k, v)
    return p
