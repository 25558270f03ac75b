#!/usr/bin/env python3
"""the bench line's figures, one per row (tools/line_summary.py <bench.json>)"""
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
for k in ("value", "ms_per_step", "value_repeat_on_warm_state", "ms_per_step_repeat_on_warm_state", "value_with_cloud_schedule_kept", "ms_per_step_with_cloud_schedule_kept"):
    print(k, d.get(k))
r = d["roofline"]
print("roofline", json.dumps({a: b for a, b in r.items() if a not in ("flop_view", "pmc", "binding")})[:700])
print("binding", json.dumps(r.get("binding"))[:500])
for k in ("shipped_point2plane_gn", "time_to_pose", "cold_start", "c5_sharded", "odometry_stream", "odometry_stream_10hz", "odometry_stream_10hz_pinned", "odometry_stream_small",
          "odometry_stream_small_10hz", "config3_batch", "config3_batch_shipped", "loop_closure_montecarlo", "dense_mfma", "mixed_load"):
    v = d.get(k)
    if isinstance(v, dict):
        v = {a: b for a, b in v.items() if a not in ("ms_per_scan", "pmc", "flop_view", "workload", "note", "roofline", "cpu", "state", "regime", "keyframes")}
    print(k, json.dumps(v)[:520])
print("align_e2e", json.dumps({k: (round(v["gpu"]["ms"], 3), round(v["gpu"].get("iterations_ms", 0), 3)) for k, v in d.get("align_e2e", {}).items()}))
print("cpu_baseline", json.dumps(d.get("cpu_baseline"))[:300])
print("pose_err_vs_cpu", d.get("pose_err_vs_cpu"))
for k in ("odometry_stream", "odometry_stream_small"):
    print(k, "trajectory", json.dumps((d.get(k) or {}).get("trajectory"))[:700])
print("c5 near_converged", json.dumps((d.get("c5_sharded") or {}).get("near_converged"))[:500])
