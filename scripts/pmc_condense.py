"""Shrinks a rocprofv3 output directory before gpurun merges it back (64 MiB limit): every *counter_collection.csv is rewritten with
one row per (kernel, counter) holding the AVERAGE over the kernel's launches (what scripts/profile_report.py forms anyway), and the
per-dispatch *kernel_trace.csv files are dropped (the *kernel_stats.csv summary stays)."""
import collections, csv, glob, os, sys
for d in sys.argv[1:]:
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        s = collections.defaultdict(float); n = collections.defaultdict(int)
        for row in csv.DictReader(open(f)):
            k = (row["Kernel_Name"], row["Counter_Name"]); s[k] += float(row["Counter_Value"]); n[k] += 1
        with open(f, "w", newline="") as o:
            w = csv.writer(o); w.writerow(["Kernel_Name", "Counter_Name", "Counter_Value", "Launches"])
            for k in s: w.writerow([k[0], k[1], s[k] / n[k], n[k]])
    for f in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
        os.remove(f)
