mkdir -p gpurun_out/r2b
python bench.py --steps 20 --warmup 5 > gpurun_out/r2b/bench1.log 2>&1; echo "rc=$?" >> gpurun_out/r2b/bench1.log
QRK_BENCH_BACKEND=gloo python bench.py --gpus 2 --steps 20 --warmup 5 --strong-blocks 10000,100000 > gpurun_out/r2b/bench2.log 2>&1; echo "rc=$?" >> gpurun_out/r2b/bench2.log
python -m pytest tests/test_sharding_gloo.py -q -m gpu > gpurun_out/r2b/shard.log 2>&1; echo "rc=$?" >> gpurun_out/r2b/shard.log
tail -3 gpurun_out/r2b/bench1.log | cut -c1-3000; tail -3 gpurun_out/r2b/bench2.log | cut -c1-3000; tail -3 gpurun_out/r2b/shard.log
