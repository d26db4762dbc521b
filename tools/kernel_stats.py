"""print per-kernel averages from a rocprofv3 --kernel-trace --stats output directory: kernel_stats.py DIR [name-substring ...]"""
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True)[0]
pats = sys.argv[2:] or ["mee::"]
for r in csv.DictReader(open(f)):
    if any(p in r["Name"] for p in pats):
        print("%-72s calls %5s avg %8.1f us  min %8.1f  max %8.1f" % (r["Name"][:72], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3, float(r["MaxNs"]) / 1e3))
