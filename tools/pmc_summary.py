"""Aggregate rocprofv3 --pmc csv output per kernel: sum of each counter over dispatches / dispatch count."""
import csv, glob, sys, re, collections
d = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.defaultdict(set)
for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        m = re.search(r"(k\d*_\w+(<[\w, ]+>)?|__amd\w+)", r["Kernel_Name"]); k = m.group(1) if m else r["Kernel_Name"][:40]
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
        cnt[k].add(r["Dispatch_Id"])
for k in acc:
    n = len(cnt[k])
    print(k, "dispatches", n)
    for c, v in sorted(acc[k].items()):
        print("   %-28s %16.0f per dispatch" % (c, v / n))
