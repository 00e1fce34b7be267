mkdir -p gpurun_out/r2a
python tools/ab.py run 10000 > gpurun_out/r2a/ab.log 2>&1
python -m pytest tests/test_ties_gpu.py tests/test_bd_gpu.py tests/test_small_tiles_gpu.py -q -m gpu > gpurun_out/r2a/ties.log 2>&1; echo "ties rc=$?" >> gpurun_out/r2a/ties.log
cat gpurun_out/r2a/ab.log; tail -6 gpurun_out/r2a/ties.log
