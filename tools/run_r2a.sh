mkdir -p gpurun_out/r2a
python -m pytest tests/test_angular.py -q -m gpu -k configs3 > gpurun_out/r2a/cfg3.log 2>&1; echo "rc=$?" >> gpurun_out/r2a/cfg3.log
tail -30 gpurun_out/r2a/cfg3.log
