"""Kernel statistics from a rocprofv3 result database (rocpd sqlite): name, calls, total ms, average us.
Usage: python tools/rocpd_stats.py path/to/results.db [divide_by]"""
import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
div = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
cur = db.cursor()
tabs = [r[0] for r in cur.execute("select name from sqlite_master where type in ('table','view')")]
kd = [t for t in tabs if t.startswith('rocpd_kernel_dispatch')][0]
ks = [t for t in tabs if t.startswith('rocpd_info_kernel_symbol')][0]
q = (f"select s.kernel_name, count(*), sum(d.end-d.start)/1e6, avg(d.end-d.start)/1e3 from {kd} d join {ks} s "
     f"on d.kernel_id=s.id group by s.kernel_name order by 3 desc")
print(f"{'kernel':100s} {'calls':>8s} {'total ms':>10s} {'avg us':>10s}")
for name, n, tot, avg in cur.execute(q):
    print(f"{name[:100]:100s} {n / div:8.0f} {tot / div:10.3f} {avg:10.2f}")
