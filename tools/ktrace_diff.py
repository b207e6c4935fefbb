"""Per-step kernel time of two rocprofv3 --kernel-trace runs side by side (development helper): whole steps only (between the
sgd_ema launches), aggregated by full kernel name, sorted by the difference.   python tools/ktrace_diff.py DIR_A DIR_B [top]"""
import collections
import csv
import glob
import re
import sys


def load(d):
    f = glob.glob(d + "/**/*kernel_trace.csv", recursive=True)[0]
    rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
    idx = [i for i, r in enumerate(rows) if "sgd_ema" in r["Kernel_Name"]]
    a, b = idx[1], idx[-1]
    steps = len(idx) - 2
    agg = collections.defaultdict(lambda: [0, 0.0])
    for r in rows[a + 1:b + 1]:
        n = re.sub(r"\(anonymous namespace\)::|^void |ustrun_b::", "", r["Kernel_Name"])
        n = re.sub(r"\(.*$", "", n)
        agg[n][0] += 1
        agg[n][1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    wall = (int(rows[b]["End_Timestamp"]) - int(rows[a]["End_Timestamp"])) / 1e3 / steps
    return {k: (v[0] / steps, v[1] / steps) for k, v in agg.items()}, steps, wall


A, sa, wa = load(sys.argv[1])
B, sb, wb = load(sys.argv[2])
top = int(sys.argv[3]) if len(sys.argv) > 3 else 30
print(f"A: {sa} steps, {wa:.1f} us wall per step, kernel sum {sum(v[1] for v in A.values()):.1f};  "
      f"B: {sb} steps, {wb:.1f} us wall, kernel sum {sum(v[1] for v in B.values()):.1f}")
out = []
for k in set(A) | set(B):
    a, b = A.get(k, (0, 0.0)), B.get(k, (0, 0.0))
    out.append((b[1] - a[1], k, a, b))
out.sort()
for d, k, a, b in out[:top] + [(0, "...", (0, 0), (0, 0))] + out[-top:]:
    print(f"{d:9.1f} us   A {a[0]:6.1f}x {a[1]:9.1f}   B {b[0]:6.1f}x {b[1]:9.1f}   {k[:120]}")
