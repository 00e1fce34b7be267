mkdir -p gpurun_out/r2d
python -m pytest tests/test_bd_gpu.py tests/test_ties_gpu.py tests/test_golden_gpu.py tests/test_small_tiles_gpu.py -q -m gpu -x > gpurun_out/r2d/tests.log 2>&1; echo "rc=$?" >> gpurun_out/r2d/tests.log
tail -4 gpurun_out/r2d/tests.log
python tools/ab.py run 10000 > gpurun_out/r2d/ab.log 2>&1; cat gpurun_out/r2d/ab.log
python tools/ab.py run 160000 > gpurun_out/r2d/ab160.log 2>&1; cat gpurun_out/r2d/ab160.log
