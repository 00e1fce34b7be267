"""Timeline of the last N kernel dispatches of a rocprofv3 result database (rocpd sqlite): start and end in ms relative to the
first of them, duration, grid, name.  Usage: python tools/rocpd_timeline.py path/to/results.db [N]"""
import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
N = int(sys.argv[2]) if len(sys.argv) > 2 else 40
cur = db.cursor()
tabs = [r[0] for r in cur.execute("select name from sqlite_master where type in ('table','view')")]
kd = [t for t in tabs if t.startswith('rocpd_kernel_dispatch')][0]
ks = [t for t in tabs if t.startswith('rocpd_info_kernel_symbol')][0]
cols = [r[1] for r in cur.execute(f"pragma table_info({kd})")]
gx = "d.grid_size_x" if "grid_size_x" in cols else ("d.grid_x" if "grid_x" in cols else "0")
wx = "d.workgroup_size_x" if "workgroup_size_x" in cols else ("d.workgroup_x" if "workgroup_x" in cols else "1")
rows = list(cur.execute(f"select d.start, d.end, {gx}, {wx}, s.kernel_name from {kd} d join {ks} s on d.kernel_id=s.id order by d.start"))[-N:]
t0 = rows[0][0]
for st, en, g, w, name in rows:
    print(f"{(st - t0) / 1e6:9.3f} {(en - t0) / 1e6:9.3f}  {(en - st) / 1e3:10.1f} us  wgs={int(g) // max(int(w), 1):6d} x{int(w):5d}  {name[:90]}")
