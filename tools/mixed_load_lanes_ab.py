import importlib, os, sys
sys.path.insert(0, os.getcwd())
import bench
pkg = importlib.import_module("mola-fe-lidar_amd"); synth = importlib.import_module("mola-fe-lidar_amd.synth"); lib = importlib.import_module("mola-fe-lidar_amd._lib")
for rep in range(3):
    for lpq in (None, "2"):
        os.environ.pop("MOLA_ICP_KNN_Q4_LPQ", None)
        if lpq: os.environ["MOLA_ICP_KNN_Q4_LPQ"] = lpq
        lib.lib().mola_icp_debug_reload_env()
        r = bench.mixed_load_leg(pkg, synth)
        o = r["odometry_ms_c_call"]
        print("lanes %s: quiet p50 %.3f loaded p50 %.3f (x%.2f) p99 %.1f max %.1f, %.0f checks/s" % (lpq or "auto", o["quiet_p50"], o["loaded_p50"], r["p50_loaded_over_quiet"], o["loaded_p99"], o["loaded_max"], r["checks_per_s_under_odometry"]), flush=True)
