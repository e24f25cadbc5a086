#!/bin/bash
# round 6, call H: the other configurations on one GPU (records under profiles/)
set -u
O=gpurun_out; mkdir -p $O
S="--no-cpu-baseline --no-ssl-side --no-forward-only --no-step-variants"
timeout 900 python bench.py $S --model d8_inv_early_deit_huge_patch14 > $O/bench_r6_inv_n1.json 2> $O/h_inv.err; echo inv rc=$?
timeout 900 python bench.py $S --model hybrid_deit_large_patch16 > $O/bench_r6_vitl_n1.json 2> $O/h_vitl.err; echo vitl rc=$?
timeout 900 python bench.py $S --accum 4 --steps 6 --warmup 2 > $O/bench_r6_accum4_n1.json 2> $O/h_acc.err; echo accum rc=$?
timeout 900 python bench.py $S --force-ddp --accum 4 --steps 6 --warmup 2 > $O/bench_r6_accum4_ddp1.json 2> $O/h_accd.err; echo accum-ddp rc=$?
timeout 900 python bench.py $S --force-ddp > $O/bench_r6_ddp1_graph.json 2> $O/h_ddp.err; echo ddp rc=$?
timeout 900 python bench.py $S --force-ddp --torch-ddp > $O/bench_r6_ddp1_torchddp.json 2> $O/h_tddp.err; echo torch-ddp rc=$?
python3 - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/bench_r6_*.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        print(f, d["value"], d["ms_per_step"], d["host_issue_ms_per_step"], d["config"]["launch"], "|", d["config"].get("gradient_reduction"))
    except Exception as e:
        print(f, "unreadable", e)
PY
