#!/bin/bash
# copies the summaries of an end-of-round evidence run (tools/run_r6_end.sh OUT) into profiles/ under the round's names
OUT=${1:-gpurun_out/r6end}
cp $OUT/bench.json profiles/r06_end_bench.json
cp $OUT/kernel_stats.csv profiles/r06_kernel_stats.csv
cp $OUT/k1_pmc_summary.txt profiles/r06_k1_pmc_summary.txt
{ echo "# python -m pytest tests -q -m gpu on the GPU box, final build of round 6 ($OUT)"; tail -4 $OUT/gpu_tests.txt; } > profiles/r06_gpu_tests.txt
{ echo "# end-of-round probes, final build of round 6 ($OUT; tools/run_r6_end.sh)"; for f in mixed_probe k2_sizes angular_probe caqr_probe strips_probe; do echo "## $f"; cat $OUT/$f.txt; done; echo "## bench line under rocprofv3 (truncated)"; cat $OUT/bench_profiled_line.txt; } > profiles/r06_end_probes.txt
python3 - "$OUT" <<'PY'
import json, re, sys
out = sys.argv[1]
txt = open(f"{out}/k1_pmc_summary.txt").read()
f = float(re.search(r"FETCH_SIZE\s+n=\s*\d+ avg=([0-9.e+]+)", txt).group(1))
w = float(re.search(r"WRITE_SIZE\s+n=\s*\d+ avg=([0-9.e+]+)", txt).group(1))
n = int(re.search(r"FETCH_SIZE\s+n=\s*(\d+)", txt).group(1))
hb = float(re.search(r"hbm_bytes_per_launch ([0-9.e+]+)", txt).group(1))
old = json.load(open("profiles/pmc_traffic.json"))
old.update({"fetch_size_kb_raw": f, "write_size_kb": w, "hbm_bytes_per_launch": hb,
            "command": re.sub(r"\d+ dispatches", f"{n} dispatches", old["command"])})
json.dump(old, open("profiles/pmc_traffic.json", "w"), indent=1)
print(old)
PY
