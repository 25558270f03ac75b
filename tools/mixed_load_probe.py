#!/usr/bin/env python3
"""bench.py's mixed_load leg (the odometry stream at 10 Hz beside threads looping loop-closure checks on ONE handle) with the odometry
path's stream priority on (the product) and off (MOLA_ICP_NO_STREAM_PRIORITY), for 2 / 4 / 8 load threads: what the priority buys."""
import importlib, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
pkg = importlib.import_module("mola-fe-lidar_amd")
synth = importlib.import_module("mola-fe-lidar_amd.synth")
lib = importlib.import_module("mola-fe-lidar_amd._lib")
for n_thr in (2, 4, 8):
    for prio in (True, False, True, False):
        os.environ.pop("MOLA_ICP_NO_STREAM_PRIORITY", None)
        if not prio:
            os.environ["MOLA_ICP_NO_STREAM_PRIORITY"] = "1"
        lib.lib().mola_icp_debug_reload_env()
        r = bench.mixed_load_leg(pkg, synth, n_threads=n_thr)
        o = r["odometry_ms_c_call"]
        print("%d load threads, odometry streams at %-17s: odometry C call p50 %.3f ms quiet -> %.3f ms loaded (x%.2f), p99 %.2f -> %.2f, max loaded %.2f; %.1f checks/s" %
              (n_thr, "greatest priority" if prio else "default priority", o["quiet_p50"], o["loaded_p50"], r["p50_loaded_over_quiet"], o["quiet_p99"], o["loaded_p99"], o["loaded_max"],
               r["checks_per_s_under_odometry"]), flush=True)
os.environ.pop("MOLA_ICP_NO_STREAM_PRIORITY", None)
