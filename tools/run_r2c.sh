# round-2 baseline: GPU test suite, bench line, kernel-trace stats
mkdir -p gpurun_out/r2c
R=$(pwd)
python -m pytest tests -q -m gpu -x > gpurun_out/r2c/gpu_tests.log 2>&1; echo "rc=$?" >> gpurun_out/r2c/gpu_tests.log
tail -5 gpurun_out/r2c/gpu_tests.log
python bench.py --steps 20 --warmup 5 > gpurun_out/r2c/bench.log 2>&1; echo "rc=$?" >> gpurun_out/r2c/bench.log
tail -2 gpurun_out/r2c/bench.log | cut -c1-2500
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r2c/prof -- python3 $R/bench.py --steps 50 --warmup 5 --no-cpu-baseline > $R/gpurun_out/r2c/prof.log 2>&1
echo "prof rc=$?"
