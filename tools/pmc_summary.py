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
