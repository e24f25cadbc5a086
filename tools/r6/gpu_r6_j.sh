#!/bin/bash
set -u
O=gpurun_out; mkdir -p $O
timeout 1500 python -m pytest tests/test_train_gpu.py tests/test_kernels_gpu.py -x -q -k "captured or accum or ddp or two_rank or deferred or finish" 2>&1 | tail -4
S="--no-cpu-baseline --no-ssl-side --no-forward-only --no-step-variants --no-kernel-timing"
for i in 1 2; do
timeout 900 python bench.py $S --force-ddp > $O/j_ddp.json 2> $O/j_ddp.err; echo ddp rc=$?
python3 - <<'PY'
import json
d = json.loads(open("gpurun_out/j_ddp.json").read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"], d["host_issue_ms_per_step"], d["config"]["launch"], "|", d["config"].get("gradient_reduction"))
PY
done
timeout 900 python bench.py $S > $O/j_plain.json 2> $O/j_plain.err
python3 -c "
import json
d = json.loads(open('gpurun_out/j_plain.json').read().strip().splitlines()[-1]); print('plain', d['value'], d['ms_per_step'])"
