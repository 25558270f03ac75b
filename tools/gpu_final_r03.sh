#!/bin/bash
# round 3: everything the committed profiles/r03 and the docs quote, in one lease (run from the repository root on the GPU box):
#   bash tools/gpu_final_r03.sh          then, in the repository:
#   python tools/pmc_record.py gpurun_out/prof_r03_headline profiles/r03
#   python tools/pmc_record.py gpurun_out/prof_r03_small profiles/r03 --kernel k_nn_coop
#   python tools/pmc_record_planes.py gpurun_out/prof_r03_planes_c3 profiles/r03 ; ... prof_r03_planes_120k ...
O=gpurun_out/final_r03; mkdir -p $O
echo "== bench (all legs)"; (time timeout 900 python bench.py) > $O/bench.log 2> $O/bench.err || { tail -30 $O/bench.err; exit 1; }
tail -n1 $O/bench.log > $O/bench_line.json
echo "== bench --gpus 2 (self-launched ranks sharing this GPU)"; timeout 600 python bench.py --gpus 2 --cpu-baseline-iters 0 --e2e 0 --batch-pairs 0 --dense-iters 0 --shipped-iters 0 > $O/bench_gpus2.json 2> $O/bench_gpus2.err; tail -c 400 $O/bench_gpus2.json
echo "== one-rank RCCL path"; timeout 600 python bench.py --gpus 1 --force-dist --cpu-baseline-iters 0 --e2e 0 --batch-pairs 0 --dense-iters 0 --shipped-iters 0 > $O/bench_rccl1.json 2> $O/bench_rccl1.err; tail -c 300 $O/bench_rccl1.json
echo "== shard step"; timeout 600 python tools/shard_step.py > $O/shard_step.txt 2>&1; cat $O/shard_step.txt
echo "== headline counters"; bash tools/rocprof_headline.sh r03_headline > $O/rocprof_headline.log 2>&1; tail -5 $O/rocprof_headline.log
echo "== odometry-size matcher counters"; bash tools/rocprof_small.sh r03_small > $O/rocprof_small.log 2>&1; tail -5 $O/rocprof_small.log
echo "== plane matcher counters"; bash tools/rocprof_planes.sh r03_planes_c3 1000000 > $O/planes_c3.log 2>&1; bash tools/rocprof_planes.sh r03_planes_120k 120000 > $O/planes_120k.log 2>&1; tail -3 $O/planes_120k.log
echo "== config 0 timeline"; bash tools/rocprof_config0.sh > $O/config0_kernel_trace.txt 2>&1; tail -3 $O/config0_kernel_trace.txt; cp gpurun_out/prof_config0/trace.log $O/config0_prof.log
echo "== Monte-Carlo batch timelines"; bash tools/rocprof_mc.sh p2pl > $O/mc_p2pl.txt 2>&1; bash tools/rocprof_mc.sh p2p > $O/mc_p2p.txt 2>&1; head -12 $O/mc_p2pl.txt
echo "== headline iteration timeline"; bash tools/rocprof_timeline_any.sh 1000000 p2p 12 > $O/c3_timeline.txt 2>&1; tail -6 $O/c3_timeline.txt
