"""Summarise rocprofv3 --pmc CSV passes: per-counter average per dispatch of the factorisation kernel."""
import csv
import glob
import os
import sys
from collections import defaultdict

out = sys.argv[1]
kern = sys.argv[2] if len(sys.argv) > 2 else "bdqr"
acc = defaultdict(list)
for f in sorted(glob.glob(os.path.join(out, "p*", "**", "*counter_collection.csv"), recursive=True)):
    for row in csv.DictReader(open(f)):
        if kern in row.get("Kernel_Name", ""):
            acc[row["Counter_Name"]].append(float(row["Counter_Value"]))
print(f"# per-dispatch averages for kernels matching '{kern}'")
for k in sorted(acc):
    v = acc[k]
    print(f"{k:32s} n={len(v):4d} avg={sum(v)/len(v):.6g} min={min(v):.6g} max={max(v):.6g}")

# HBM traffic per launch for bench.py's roofline block (MI355X_MICROARCH.md: FETCH_SIZE / WRITE_SIZE are in
# kilobytes; on gfx950 FETCH_SIZE under-reports by 2x -> doubled).
if "FETCH_SIZE" in acc and "WRITE_SIZE" in acc:
    import json
    fetch = sum(acc["FETCH_SIZE"]) / len(acc["FETCH_SIZE"])
    write = sum(acc["WRITE_SIZE"]) / len(acc["WRITE_SIZE"])
    t = {"kernel": kern, "fetch_size_kb_raw": fetch, "write_size_kb": write,
         "hbm_bytes_per_launch": (2.0 * fetch + write) * 1024.0,
         "note": "FETCH_SIZE x2 (gfx950 correction) + WRITE_SIZE, KB -> bytes; per-dispatch average"}
    with open(os.path.join(out, "pmc_traffic.json"), "w") as f:
        json.dump(t, f, indent=1)
    print("hbm_bytes_per_launch", t["hbm_bytes_per_launch"])
