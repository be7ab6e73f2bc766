"""Print a compact per-kernel table from a rocprofv3 --kernel-trace --stats csv directory."""
import csv, glob, sys, re
d = sys.argv[1]
for f in glob.glob(d + "/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        m = re.search(r"(k\d*_\w+(<[\w, ]+>)?|__amd\w+)", r["Name"]); name = m.group(1) if m else r["Name"][:40]
        print("%-28s calls %4s avg %10.1f us  total %10.1f us  %5s%%" % (name[:28], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e3, r["Percentage"]))
