"""Timeline of ONE two-stage factorisation (40 000 x 2 000 pivoted) from a rocprofv3 --kernel-trace csv (tools/caqr_probe.py runs three; the
last one is taken: it starts at the last caqr_panel_kernel<false> that follows a cols_step kernel).  Per kernel name: calls, summed
duration; stage spans; and the kernels of one panel period in the middle of stage 1 with their queues -- what runs beside what.
Usage: python tools/k3_timeline.py path/to/*_kernel_trace.csv [panel index]"""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
ks = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r["Queue_Id"], int(r["Grid_Size_X"]) // max(1, int(r["Workgroup_Size_X"]))) for r in rows]
ks = sorted(k for k in ks if "caqr" in k[2] or "cols_step" in k[2])
start = 0
for i in range(1, len(ks)):
    if "caqr_panel_kernel<false>" in ks[i][2] and "cols_step" in ks[i - 1][2]: start = i
run = ks[start:]
t0 = run[0][0]
short = lambda n: n.split("(")[0].replace("void qrk::caqr::", "").replace("void qrk::cols::", "")
print(f"kernels of the last factorisation: {len(run)}, span {(run[-1][1] - t0) / 1e6:.2f} ms")
agg = collections.OrderedDict()
for s, e, n, q, g in run:
    a = agg.setdefault(short(n), [0, 0]); a[0] += 1; a[1] += e - s
for k, (c, d) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print(f"  {k:40s} calls {c:5d}  sum {d / 1e6:8.2f} ms  avg {d / c / 1e3:8.1f} us")
def union(iv):
    iv = sorted(iv); tot = 0; cs, ce = iv[0]
    for s, e in iv[1:]:
        if s > ce: tot += ce - cs; cs, ce = s, e
        else: ce = max(ce, e)
    return tot + ce - cs
c1 = [(s, e) for s, e, n, q, g in run if "caqr" in n]
c2 = [(s, e) for s, e, n, q, g in run if "cols_step" in n]
s1 = (min(s for s, _ in c1), max(e for _, e in c1))
print(f"stage 1: {(s1[0] - t0) / 1e6:.2f} .. {(s1[1] - t0) / 1e6:.2f} ms; some caqr kernel running {union(c1) / 1e6:.2f} ms")
if c2: print(f"stage 2: {(min(s for s, _ in c2) - t0) / 1e6:.2f} .. {(max(e for _, e in c2) - t0) / 1e6:.2f} ms; kernels {len(c2)}, running {union(c2) / 1e6:.2f} ms")
for name in ("caqr_panel", "caqr_apply_kernel", "caqr_apply_narrow"):
    iv = [(s, e) for s, e, n, q, g in run if name in n]
    print(f"  {name:20s} {len(iv):5d} kernels, running (union) {union(iv) / 1e6:7.2f} ms = {union(iv) / (s1[1] - s1[0]) * 100:5.1f} % of stage 1")
p0 = [i for i, k in enumerate(run) if "caqr_panel_kernel<false>" in k[2]]
pi = int(sys.argv[2]) if len(sys.argv) > 2 else len(p0) // 2
a, b = p0[pi], p0[pi + 1]
print(f"--- panel {pi} of {len(p0)}: period {(run[b][0] - run[a][0]) / 1e3:.1f} us; start / end (us from the panel's level-0 kernel), duration, queue, workgroups")
base = run[a][0]
for s, e, n, q, g in run[a:b + 1]:
    print(f"  {(s - base) / 1e3:8.1f} {(e - base) / 1e3:8.1f}  {(e - s) / 1e3:7.1f}  q{q:>3s}  wgs {g:5d}  {short(n)}")
