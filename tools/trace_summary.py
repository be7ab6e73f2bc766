"""Start / end / duration of every kernel dispatch of a rocprofv3 --kernel-trace csv directory, relative to the first start (ms)."""
import csv, glob, re, sys
d = sys.argv[1]
pat = sys.argv[2] if len(sys.argv) > 2 else "k_"
rows = []
for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if pat in r["Kernel_Name"]:
            m = re.search(r"(k\d*_\w+(<[\w, ]+>)?)", r["Kernel_Name"])
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), m.group(1) if m else r["Kernel_Name"][:30], r.get("Queue_Id", "?")))
rows.sort()
t0 = rows[0][0] if rows else 0
for s, e, n, q in rows:
    print("%-24s queue %-4s start %10.2f  end %10.2f  dur %10.2f ms" % (n, q, (s - t0) / 1e6, (e - t0) / 1e6, (e - s) / 1e6))
