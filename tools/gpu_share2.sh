O=gpurun_out/r02_share; mkdir -p $O
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --share-gpu --steps 20 --warmup 2 > $O/bench2.log 2>&1; tail -n 3 $O/bench2.log | cut -c1-1500
