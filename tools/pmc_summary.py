"""Per-kernel mean of every counter in rocprofv3 counter_collection.csv files.  usage: pmc_summary.py dir [kernel-substr...]"""
import collections
import csv
import glob
import sys

d = sys.argv[1]
want = sys.argv[2:]
agg = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.defaultdict(lambda: collections.Counter())
for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "")
        if want and not any(w in k for w in want):
            continue
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
        cnt[k][r["Counter_Name"]] += 1
for k in agg:
    print(k[:40], " ".join(f"{c}={agg[k][c] / cnt[k][c]:.4g}" for c in sorted(agg[k])))
