mkdir -p gpurun_out/r2a
python -m pytest tests/test_thin_gpu.py -q -m gpu > gpurun_out/r2a/thin.log 2>&1; echo "rc=$?" >> gpurun_out/r2a/thin.log
tail -40 gpurun_out/r2a/thin.log
