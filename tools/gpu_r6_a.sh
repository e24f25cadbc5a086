#!/bin/bash
# round 6, call A: the captured data-parallel step on a one-rank RCCL group + baseline bench
set -u
O=gpurun_out
mkdir -p $O
timeout 900 python -m pytest tests/test_train_gpu.py -x -q -k "captured_data_parallel or ddp_wrapped or two_rank or captured_step_replays" > $O/a_tests.log 2>&1
echo "tests rc=$?" >> $O/a_tests.log
tail -5 $O/a_tests.log
timeout 600 python bench.py --force-ddp --steps 20 --warmup 5 --no-cpu-baseline --no-forward-only --no-ssl-side --no-step-variants > $O/a_ddp.json 2> $O/a_ddp.err
echo "ddp bench rc=$?"; tail -3 $O/a_ddp.err; cut -c1-1500 $O/a_ddp.json
timeout 900 python bench.py --no-cpu-baseline --no-ssl-side --no-forward-only > $O/a_base.json 2> $O/a_base.err
echo "base bench rc=$?"; tail -3 $O/a_base.err
python - <<'PY'
import json
for f in ("gpurun_out/a_ddp.json", "gpurun_out/a_base.json"):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        keep = {k: v for k, v in d.items() if k in ("value", "ms_per_step", "host_issue_ms_per_step", "loss") or k.startswith(("ddp_", "eager_", "segment_"))}
        print(f, keep, d["config"].get("launch"), d["config"].get("gradient_reduction"))
    except Exception as e:
        print(f, "unreadable", e)
PY
