#!/bin/bash
# round 6: everything the committed profiles/r06 and the docs quote (run from the repository root on the GPU box), in parts that each fit
# one lease:   PART=a bash tools/gpu_final_r06.sh   (suite + the full bench line)        PART=b ...   (N > 1 rehearsal, transports, shard steps)
#              PART=c ...   (rocprofv3: bench.py itself, headline / odometry-size counters, timelines)        PART=d ... (probes of the round)
set -o pipefail
O=gpurun_out/final_r06; mkdir -p $O
LEAN="--cpu-baseline-iters 0 --e2e 0 --batch-pairs 0 --dense-iters 0 --shipped-iters 0"
case "${PART:-a}" in
a)
  if [ -z "$SKIP_SUITE" ]; then echo "== GPU suite"; timeout -k 10 1000 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; rc=$?; tail -3 $O/pytest.log; [ $rc -ne 0 ] && exit $rc; fi
  echo "== bench (all legs)"; (time timeout -k 10 1000 python bench.py) > $O/bench.log 2> $O/bench.err || { tail -30 $O/bench.err; exit 1; }
  tail -n1 $O/bench.log > $O/bench_line.json
  python tools/line_summary.py $O/bench_line.json 2>/dev/null | head -60
  ;;
b)
  echo "== bench --gpus 2 (self-launched ranks sharing this GPU: replicas of configs[3], cpu_baseline, both configs[4] regimes)"
  timeout -k 10 900 python bench.py --gpus 2 --batch-pairs 16 --e2e 0 --dense-iters 0 --shipped-iters 0 > $O/bench_gpus2.json 2> $O/bench_gpus2.err || { tail -20 $O/bench_gpus2.err; exit 1; }
  echo "== one rank through every collective"
  for ar in plain both local rccl; do
    if [ $ar = plain ]; then X=""; elif [ $ar = both ]; then X="--force-dist"; else X="--force-dist --allreduce $ar"; fi
    timeout -k 10 300 python bench.py $LEAN --c5-map 0 $X > $O/bench_one_rank_$ar.json 2> $O/bench_one_rank_$ar.err || { tail -20 $O/bench_one_rank_$ar.err; exit 1; }
  done
  echo "== shard steps"; timeout -k 10 600 python tools/shard_step.py --config c3 > $O/shard_step_c3.jsonl 2> $O/shard_step_c3.err; timeout -k 10 900 python tools/shard_step.py --config c5 > $O/shard_step_c5.jsonl 2> $O/shard_step_c5.err
  cat $O/shard_step_c3.jsonl | head -12
  ;;
c)
  echo "== bench.py under rocprofv3 --kernel-trace --stats"; bash tools/rocprof_bench.sh > $O/rocprof_bench.txt 2>&1; tail -8 $O/rocprof_bench.txt
  echo "== headline counters"; bash tools/rocprof_headline.sh r06_headline > $O/rocprof_headline.log 2>&1; tail -5 $O/rocprof_headline.log
  echo "== odometry-size matcher counters (k_nn_q4)"; bash tools/rocprof_small.sh r06_small > $O/rocprof_small.log 2>&1; tail -5 $O/rocprof_small.log
  echo "== odometry stream timeline"; bash tools/rocprof_odometry.sh > $O/rocprof_odometry.log 2>&1; tail -3 gpurun_out/prof_odometry/timeline.txt
  echo "== plane matcher at 120k x 120k (k_knn_q4 + the far launches' k_knn_coop): counters"; bash tools/rocprof_planes.sh r06_knn_q4 120000 > $O/rocprof_planes.log 2>&1; tail -3 $O/rocprof_planes.log
  echo "== plane matcher at 1M x 1M (k_knn_q4, one lane per query: the shipped pipeline's matcher at configs[2]): counters"; bash tools/rocprof_planes.sh r06_knn_q4_1m 1000000 > $O/rocprof_planes_1m.log 2>&1; tail -3 $O/rocprof_planes_1m.log
  echo "== 100k x 100k iteration timeline"; bash tools/rocprof_timeline_any.sh 100000 p2p 12 > $O/p2p_100k_timeline.txt 2>&1; tail -8 $O/p2p_100k_timeline.txt
  ;;
d)
  echo "== k_nn_q4 against the matchers it replaces"; timeout -k 10 400 python tools/q4_ab.py 12000x12000 50000x50000 100000x100000 120000x120000 125000x1000000 200000x200000 260000x260000 > $O/q4_ab.txt 2>&1; cat $O/q4_ab.txt
  echo "== the plane matcher's launches per odometry scan, device time: k_knn_coop, then k_knn_q4"; KNN_Q4=0 bash tools/gpu_kq4_trace.sh final_r06/kq4_trace_coop product; bash tools/gpu_kq4_trace.sh final_r06/kq4_trace_q4 product
  echo "== shipped-pipeline batch"; for q in 0 1; do MOLA_ICP_KNN_Q4=$q timeout -k 10 200 python tools/batch_shipped_probe.py 2>&1 | tail -1; done
  echo "== wait policies"; timeout -k 10 400 python tools/wait_policy_probe.py > $O/wait_policy.txt 2>&1; cat $O/wait_policy.txt
  ;;
esac
