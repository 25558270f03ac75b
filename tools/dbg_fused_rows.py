"""round 3 debugging aid: k_nn_tiled's fused item rows (item_row_mfma) against k_accumulate over the same pairing"""
import importlib, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pkg = importlib.import_module("mola-fe-lidar_amd")
synth = importlib.import_module("mola-fe-lidar_amd.synth")
os.environ["MOLA_ICP_COOP"] = "0"
p = pkg.Parameters(); p.matcher_threshold = 1.0
for N, M in [(64, 5000), (100, 5000), (1000, 9000), (20000, 30000), (200000, 200000), (1000000, 1000000)]:
    g, l, _ = synth.make_pair(N, M, seed=5)
    icp = pkg.ICP(device=0); icp.set_map(g); icp.set_local(l)
    accs = {}
    for fused in (1, 0):
        if fused: os.environ.pop("MOLA_ICP_NO_FUSED_ROWS", None)
        else: os.environ["MOLA_ICP_NO_FUSED_ROWS"] = "1"
        pkg._lib.lib().mola_icp_debug_reload_env()
        for rep in range(2):
            idx, d2, n = icp.match(np.eye(4), 1.0, N, pkg.NN_TILED)
            accs[(fused, rep)] = icp.accumulate(p, np.eye(4))
    a, b = accs[(1, 1)], accs[(0, 1)]
    rel = np.abs(a - b) / np.maximum(1e-300, np.abs(b))
    print(N, M, "n", n, "fused n", a[16], a[0], "unfused n", b[16], "max rel", rel.max(), "argmax", rel.argmax(), "first-launch same:", np.array_equal(accs[(1, 0)], accs[(1, 1)]))
    if rel.max() > 1e-11:
        print("  fused  ", a); print("  unfused", b)
    icp.close()
