"""GPU busy / idle accounting from a rocprofv3 --kernel-trace CSV (development helper): union of kernel intervals over the traced
window, idle gaps above a threshold with the kernels around them, and the busy time per kernel name.

    python tools/gpu_gaps.py DIR [min_gap_us] [skip_first_n_kernels]
"""
import csv
import glob
import sys
from collections import defaultdict

f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
thr = float(sys.argv[2]) if len(sys.argv) > 2 else 20.0
skip = int(sys.argv[3]) if len(sys.argv) > 3 else 0
rows = []
for r in csv.DictReader(open(f)):
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
rows = rows[skip:]
t0, t1 = rows[0][0], max(r[1] for r in rows)
busy, cur_s, cur_e = 0, rows[0][0], rows[0][1]
gaps = []
prev = rows[0]
for s, e, n in rows[1:]:
    if s > cur_e:
        busy += cur_e - cur_s
        if (s - cur_e) / 1e3 >= thr:
            gaps.append(((s - cur_e) / 1e3, prev[2][:60], n[:60], (cur_e - t0) / 1e6))
        cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
    if e >= prev[1]:
        prev = (s, e, n)
busy += cur_e - cur_s
print(f"window {(t1 - t0) / 1e6:.3f} ms, busy {busy / 1e6:.3f} ms, idle {(t1 - t0 - busy) / 1e6:.3f} ms, {len(rows)} kernels")
tot = defaultdict(float)
for g in gaps:
    tot[(g[1], g[2])] += g[0]
print(f"{len(gaps)} gaps >= {thr} us, total {sum(g[0] for g in gaps) / 1e3:.3f} ms; by (before -> after):")
for (a, b), v in sorted(tot.items(), key=lambda kv: -kv[1])[:15]:
    print(f"  {v / 1e3:8.3f} ms   {a}  ->  {b}")
