"""Print the top rows of a rocprofv3 kernel-stats CSV (development helper).  usage: kstats.py DIR [N]"""
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True)[0]
n = int(sys.argv[2]) if len(sys.argv) > 2 else 20
for r in list(csv.DictReader(open(f)))[:n]:
    print(f'{r["Name"][:78]:78s} {int(r["Calls"]):6d} {int(r["TotalDurationNs"])/1e6:9.3f} ms {float(r["AverageNs"])/1e3:9.1f} us {r["Percentage"]:>6s}%')
