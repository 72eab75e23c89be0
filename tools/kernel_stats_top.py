"""Print the top rows of a rocprofv3 --stats kernel_stats CSV (argument: a directory or the CSV)."""
import csv, glob, os, sys
p = sys.argv[1]
if os.path.isdir(p):
    p = sorted(glob.glob(os.path.join(p, "**", "*kernel_stats.csv"), recursive=True))[0]
for r in list(csv.DictReader(open(p)))[: int(sys.argv[2]) if len(sys.argv) > 2 else 12]:
    print(f"{r['Name'][:100]:100s} {int(r['Calls']):6d} {float(r['AverageNs']) / 1e3:9.1f} us avg {float(r['Percentage']):6.2f} %")
