
python bench.py --steps 5 --warmup 1 --cpu-seconds 4 > gpurun_out/bench_r03a.json 2> gpurun_out/bench_r03a.err; echo rc=$?; tail -c 3000 gpurun_out/bench_r03a.err
python bench.py --workload timeshard --log2-samples 26 --steps 3 --warmup 1 --no-cpu-baseline 2>&1 | tail -3
python bench.py --workload fanout --force-dist --steps 3 --warmup 1 --no-cpu-baseline 2>&1 | tail -3
python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --workload timeshard --log2-samples 26 --steps 3 --warmup 1 --dist-backend gloo --same-device 2>&1 | tail -3
