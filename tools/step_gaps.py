"""Developer tool: inter-kernel gaps of the replayed train step from a `rocprofv3 --kernel-trace` directory.
    python tools/step_gaps.py <trace dir> [out.txt]
A step = lamb_stage1 start -> next lamb_stage1 start; prints wall, busy (sum of kernel durations), idle, and which kernels
are followed by the longest gaps."""
import collections, csv, glob, os, re, sys
import numpy as np

d = sys.argv[1]
files = glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)
rows = []
for f in files:
    rows += list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
marks = [i for i, r in enumerate(rows) if "lamb_stage1" in r["Kernel_Name"]]
out = []
steps = []
for a, b in zip(marks[:-1], marks[1:]):
    st = rows[a:b]
    wall = int(rows[b]["Start_Timestamp"]) - int(st[0]["Start_Timestamp"])
    busy = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in st)
    steps.append((wall, busy, len(st), a, b))
walls = np.array([s[0] for s in steps]) / 1e6
out.append(f"{len(steps)} steps; wall ms: " + " ".join(f"{w:.2f}" for w in walls))
# the replayed (graph) steps are the fastest ones: take the median of the fastest half
order = np.argsort(walls)
pick = steps[order[len(order) // 4]]
wall, busy, n, a, b = pick
st = rows[a:b + 1]
gaps = [(int(st[i + 1]["Start_Timestamp"]) - int(st[i]["End_Timestamp"]), re.sub(r"\(.*", "", st[i]["Kernel_Name"])[:70]) for i in range(len(st) - 1)]
g = np.array([x[0] for x in gaps])
out.append(f"picked step: wall {wall / 1e6:.3f} ms, busy {busy / 1e6:.3f} ms, kernels {n}, gaps sum {g[g > 0].sum() / 1e6:.3f} ms "
           f"(overlaps {(-g[g < 0]).sum() / 1e6:.3f} ms), gap median {np.median(g) / 1e3:.2f} us, p90 {np.percentile(g, 90) / 1e3:.2f} us, max {g.max() / 1e3:.1f} us")
per = collections.defaultdict(lambda: [0, 0])
for gap, nm in gaps:
    per[nm][0] += 1
    per[nm][1] += gap
out.append("gap after kernel (sum us, count, mean us):")
for nm, (c, t) in sorted(per.items(), key=lambda kv: -kv[1][1])[:25]:
    out.append(f"  {t / 1e3:9.1f} {c:5d} {t / c / 1e3:8.2f}  {nm}")
text = "\n".join(out)
print(text)
if len(sys.argv) > 2:
    open(sys.argv[2], "w").write(text + "\n")
