#!/bin/bash
set -o pipefail
O=gpurun_out/r04j; mkdir -p $O
MOLA_ICP_COOP=1 timeout -k 10 900 python tools/shard_step.py --config c5 --worlds 1,2,4 --balance-rounds 1 > $O/shard_step_c5_coop.jsonl 2> $O/shard_step_c5_coop.err || { tail -c 1500 $O/shard_step_c5_coop.err; exit 1; }
python - <<'PY'
import json
for ln in open("gpurun_out/r04j/shard_step_c5_coop.jsonl"):
    j = json.loads(ln)
    print("COOP=1 c5 world", j["world"], j["cuts"], "slowest", round(j["step_ms_slowest_rank"], 4), "mean", round(j["step_ms_mean_rank"], 4),
          "ms", [round(r["ms_per_iteration"], 3) for r in j["ranks"]], "q", [r["queries"] for r in j["ranks"]])
PY
