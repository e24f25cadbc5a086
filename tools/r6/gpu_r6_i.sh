#!/bin/bash
set -u
O=gpurun_out; mkdir -p $O
timeout 1500 python -m pytest tests/test_train_gpu.py -x -q -k "captured or accum or ddp or two_rank or segment" 2>&1 | tail -8
S="--no-cpu-baseline --no-ssl-side --no-forward-only --no-step-variants --no-kernel-timing"
timeout 900 python bench.py $S --accum 4 --steps 6 --warmup 2 > $O/i_accum4.json 2> $O/i_acc.err; echo accum rc=$?
timeout 900 python bench.py $S --force-ddp --accum 4 --steps 6 --warmup 2 > $O/i_accum4_ddp.json 2> $O/i_accd.err; echo accum-ddp rc=$?
python3 - <<'PY'
import json, glob
for f in ("gpurun_out/i_accum4.json", "gpurun_out/i_accum4_ddp.json"):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        print(f, d["value"], d["ms_per_step"], d["host_issue_ms_per_step"], d["config"]["launch"], "|", d["config"].get("gradient_reduction"))
    except Exception as e:
        print(f, "unreadable", e)
PY
tail -3 $O/i_acc.err
