mkdir -p gpurun_out/r2a
python -m pytest tests/test_dense_gpu.py tests/test_angular.py tests/test_cpp_facade_gpu.py tests/test_qproduct_gpu.py tests/test_lm_gpu.py -q -m gpu > gpurun_out/r2a/dense.log 2>&1; echo "rc=$?" >> gpurun_out/r2a/dense.log
tail -30 gpurun_out/r2a/dense.log
