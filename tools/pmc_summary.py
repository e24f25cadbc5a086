"""Average rocprofv3 --pmc counters per kernel name.  Usage: python tools/pmc_summary.py <dir> [name filter]"""
import collections
import csv
import glob
import sys

d = sys.argv[1]
flt = sys.argv[2] if len(sys.argv) > 2 else ""
acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
    with open(f) as fh:
        for r in csv.DictReader(fh):
            k = r["Kernel_Name"]
            if flt and flt not in k:
                continue
            a = acc[k[:70]][r["Counter_Name"]]
            a[0] += float(r["Counter_Value"])
            a[1] += 1
for k, cs in acc.items():
    print(k)
    for c, (s, n) in sorted(cs.items()):
        print(f"    {c:32s} {s / n:16.0f}   (n={n})")
