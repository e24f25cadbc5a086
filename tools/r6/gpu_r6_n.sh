#!/bin/bash
# round 6: the bench flows of the data-parallel step on a one-rank RCCL group (captured step x3, the watchdog via OCTIC_BENCH_FAKE_HANG, the default line with its child-process side figures)
set -u
O=gpurun_out; mkdir -p $O
timeout 900 python -m pytest tests/test_train_gpu.py -x -q -k "captured_data_parallel or captured_accumulated or ddp_wrapped" 2>&1 | tail -3
S="--no-cpu-baseline --no-ssl-side --no-forward-only --no-step-variants"
for i in 1 2 3; do
timeout 900 python bench.py $S --force-ddp > $O/n_ddp$i.json 2> $O/n_ddp$i.err; echo ddp rc=$?
done
OCTIC_BENCH_FAKE_HANG=1 OCTIC_BENCH_WATCHDOG_S=45 timeout 900 python bench.py $S --force-ddp > $O/n_hang.json 2> $O/n_hang.err; echo hang rc=$?
( time python bench.py --no-cpu-baseline --no-ssl-side --no-forward-only > $O/n_full.json 2> $O/n_full.err ) 2>&1 | tail -3
python3 - <<'PY'
import json
for f in ("gpurun_out/n_ddp1.json", "gpurun_out/n_ddp2.json", "gpurun_out/n_ddp3.json", "gpurun_out/n_hang.json", "gpurun_out/n_full.json"):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        print(f, d["value"], d["ms_per_step"], d["host_issue_ms_per_step"], d["config"]["launch"], d.get("eager_reducer_ms_per_step"), "roofline" in d,
              {k: v for k, v in d.items() if k.startswith("ddp_") and "note" not in k and "reduction" not in k})
    except Exception as e:
        print(f, "unreadable", e)
PY
grep -i "watchdog\|error" $O/n_hang.err | tail -3
