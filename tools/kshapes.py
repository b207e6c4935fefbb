"""Group a rocprofv3 kernel trace by (kernel, grid, LDS size): calls, total and mean duration (development helper).
usage: kshapes.py DIR name-substring"""
import collections, csv, glob, re, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
sub = sys.argv[2] if len(sys.argv) > 2 else ""
agg = collections.defaultdict(lambda: [0, 0.0])
for r in csv.DictReader(open(f)):
    if sub in r["Kernel_Name"]:
        m = re.search(r"([A-Za-z_0-9]+_kernel(<[^>]*>)?)", r["Kernel_Name"])
        name = m.group(1) if m else r["Kernel_Name"][:60]
        k = (name, int(r["Grid_Size_X"]) // max(int(r["Workgroup_Size_X"]), 1), int(r["Grid_Size_Y"]), int(r["LDS_Block_Size"]))
        agg[k][0] += 1
        agg[k][1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
for k, (n, t) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print(f"{k[0]:60s} blocks={k[1]:6d} y={k[2]:4d} lds={k[3]:6d} calls={n:5d} total={t/1e3:8.3f} ms mean={t/n:8.1f} us")
