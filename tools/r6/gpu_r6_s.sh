#!/bin/bash
set -u
export TMPDIR=/tmp
O=gpurun_out; mkdir -p $O
cd /tmp
CLS2=2 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/prof_cls2 -- python3 $GRAFT_REPO_ROOT/tools/ab_dense_image.py > $GRAFT_REPO_ROOT/$O/s_ab.txt 2>&1
cd $GRAFT_REPO_ROOT
python3 - <<'PY'
import csv, glob, collections
rows = []
for f in glob.glob("gpurun_out/prof_cls2/**/*kernel_trace.csv", recursive=True):
    rows += list(csv.DictReader(open(f)))
d = collections.defaultdict(list)
for r in rows:
    n = r["Kernel_Name"]
    if "dense_cls" in n:
        key = (n.split("(")[0][-40:], r.get("Grid_Size") or r.get("Grid_Size_X"), r.get("Workgroup_Size") or r.get("Workgroup_Size_X"))
        d[key].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
for k, v in sorted(d.items()):
    v.sort()
    print(k, len(v), "median us", v[len(v)//2] / 1e3)
PY
rm -rf $O/prof_cls2
