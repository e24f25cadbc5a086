"""Condense a rocprofv3 --kernel-trace --stats output directory into a small text summary (top kernels by
total time) suitable for committing under profiles/."""
import csv
import glob
import sys

d, out = sys.argv[1], sys.argv[2]
files = glob.glob(d + "/**/*kernel_stats.csv", recursive=True)
if not files:
    sys.exit("no *kernel_stats.csv under " + d)
rows = []
for f in files:
    with open(f) as fh:
        rows += list(csv.DictReader(fh))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
with open(out, "w") as fh:
    fh.write(f"# rocprofv3 --kernel-trace --stats summary ({len(rows)} kernels, total {tot/1e6:.2f} ms of GPU kernel time)\n")
    fh.write(f"{'calls':>7} {'total_ms':>10} {'avg_us':>10} {'pct':>6}  name\n")
    for r in rows[:60]:
        fh.write(f"{int(r['Calls']):7d} {float(r['TotalDurationNs'])/1e6:10.3f} {float(r['AverageNs'])/1e3:10.2f} "
                 f"{float(r['Percentage']):6.2f}  {r['Name'][:150]}\n")
print(open(out).read())
