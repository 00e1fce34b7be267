mkdir -p gpurun_out/r2a
python -m pytest tests/test_ties_gpu.py -q -m gpu > gpurun_out/r2a/ties.log 2>&1; echo "ties rc=$?" >> gpurun_out/r2a/ties.log
python -m pytest tests/ -q -m gpu --deselect tests/test_ties_gpu.py > gpurun_out/r2a/all.log 2>&1; echo "all rc=$?" >> gpurun_out/r2a/all.log
python bench.py --steps 20 --warmup 5 > gpurun_out/r2a/bench.log 2>&1
tail -15 gpurun_out/r2a/ties.log; tail -8 gpurun_out/r2a/all.log; tail -1 gpurun_out/r2a/bench.log | cut -c1-900
