mkdir -p gpurun_out
timeout 900 python tools/extra_bench.py > gpurun_out/extra.json 2> gpurun_out/extra.err; tail -5 gpurun_out/extra.err; cat gpurun_out/extra.json
echo "--- torchrun 1 rank"
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --steps 5 --warmup 1 --no-large 2>&1 | tail -2 | cut -c1-400
