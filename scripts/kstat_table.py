"""rocprofv3 --stats directory -> a small per-kernel table (calls, average us, total per iteration) on stdout: python scripts/kstat_table.py <dir> [iterations]"""
import csv, glob, os, sys
d, it = sys.argv[1], float(sys.argv[2]) if len(sys.argv) > 2 else None
for f in glob.glob(os.path.join(d, "**", "*kernel_stats.csv"), recursive=True):
    rows = sorted(csv.DictReader(open(f)), key=lambda r: -float(r["TotalDurationNs"]))
    tot = sum(float(r["TotalDurationNs"]) for r in rows)
    print("%-60s %8s %9s %9s" % ("kernel", "calls", "avg us", "% time"))
    for r in rows:
        if float(r["Percentage"]) < 0.05: continue
        n = r["Name"].split("(")[0].replace("void ", "").replace("gsr::", "")[:60]
        print("%-60s %8s %9.2f %9.2f" % (n, r["Calls"], float(r["AverageNs"]) / 1e3, float(r["Percentage"])))
    print("sum of kernel time %.1f us%s" % (tot / 1e3, (" = %.1f us per iteration over %g iterations" % (tot / 1e3 / it, it)) if it else ""))
