"""Per-kernel averages of whatever counters a rocprofv3 --pmc pass collected.  Usage: python tools/pmc_any.py <dir> [substr]"""
import collections, csv, glob, sys
d = sys.argv[1]
sub = sys.argv[2] if len(sys.argv) > 2 else ""
acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
    with open(f) as fh:
        for r in csv.DictReader(fh):
            if sub in r["Kernel_Name"]:
                a = acc[r["Kernel_Name"][:70]][r["Counter_Name"]]
                a[0] += float(r["Counter_Value"]); a[1] += 1
for k, cs in acc.items():
    print(k, {n: round(v[0] / v[1], 1) for n, v in cs.items()}, "launches", max(v[1] for v in cs.values()))
