"""Sum rocprofv3 --pmc counters per kernel name (development helper).  usage: pmc_summary.py DIR [name-substring]"""
import collections, csv, glob, sys
sub = sys.argv[2] if len(sys.argv) > 2 else ""
agg = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.Counter()
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if sub in k:
            agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
            cnt[(k, r["Counter_Name"])] += 1
for k, d in agg.items():
    print(k[:110])
    for c, v in sorted(d.items()):
        n = cnt[(k, c)]
        print(f"    {c:32s} {v / n:16.1f} per launch ({n} launches)")
