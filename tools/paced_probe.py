#!/usr/bin/env python3
"""Why does a scan cost more when the scans arrive every 100 ms than back to back?  The paced leg of bench.py three ways: as it is;
with the GPU kept busy between scans by a background thread (a tiny kernel every millisecond); with the HOST kept busy instead (the
same thread spinning on the CPU).  Prints the median per-scan time (whole call / C call only) of each."""
import importlib, os, sys, threading, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
pkg = importlib.import_module("mola-fe-lidar_amd")
synth = importlib.import_module("mola-fe-lidar_amd.synth")
stop = threading.Event()


def gpu_warm():
    x = torch.zeros(1 << 16, device="cuda")
    while not stop.is_set():
        x.add_(1.0)
        time.sleep(0.001)


def cpu_warm():
    while not stop.is_set():
        t = time.perf_counter()
        while time.perf_counter() - t < 0.0005:
            pass
        time.sleep(0.0005)


for name, fn in (("as it is", None), ("GPU kept busy between scans", gpu_warm), ("host thread kept busy between scans", cpu_warm), ("as it is, again", None)):
    stop.clear()
    th = None
    if fn is not None:
        th = threading.Thread(target=fn, daemon=True)
        th.start()
    r = bench.odometry_stream_leg(pkg, synth, period_s=0.1, passes=2)
    stop.set()
    if th is not None:
        th.join()
    print("%-40s median %.3f ms per scan (C call %.3f), min %.3f" % (name, r["ms_per_scan_median"], r["ms_per_scan_median_c_call_only"], r["ms_per_scan_min"]), flush=True)
r = bench.odometry_stream_leg(pkg, synth)
print("%-40s median %.3f ms per scan (C call %.3f)" % ("back to back", r["ms_per_scan_median"], r["ms_per_scan_median_c_call_only"]), flush=True)
