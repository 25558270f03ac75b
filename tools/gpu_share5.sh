O=gpurun_out/r02_share; mkdir -p $O
timeout 800 python -m torch.distributed.run --nnodes=1 --nproc-per-node 5 --master-addr 127.0.0.1 --master-port 29521 bench.py --gpus 5 --share-gpu --steps 40 --warmup 2 > $O/bench5.log 2>&1; tail -n 2 $O/bench5.log | cut -c1-900
