mkdir -p gpurun_out/r2a
python -m pytest tests/test_banded.py tests/test_cpp_facade_gpu.py -q -m gpu > gpurun_out/r2a/bb.log 2>&1; echo "rc=$?" >> gpurun_out/r2a/bb.log
tail -25 gpurun_out/r2a/bb.log
