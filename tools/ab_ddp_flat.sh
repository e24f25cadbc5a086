#!/bin/bash
# Developer A/B (GPU box, repo root): the bench's eager-DDP side figures with the small gradients as one flat all-reduce
# (train.DDP_FLAT_SMALL_NUMEL, default) and with everything inside DistributedDataParallel's buckets (0).
python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-forward-only --no-kernel-timing --no-ssl-side 2>/dev/null > gpurun_out/ab_on.json
python - > gpurun_out/ab_off.json 2>/dev/null <<PY
import octic_vits_amd.train as TR
TR.DDP_FLAT_SMALL_NUMEL = 0
import runpy, sys
sys.argv = ["bench.py", "--steps", "10", "--warmup", "3", "--no-cpu-baseline", "--no-forward-only", "--no-kernel-timing", "--no-ssl-side"]
try:
    runpy.run_path("bench.py", run_name="__main__")
except SystemExit:
    pass
PY
python - <<PY
import json
for f in ("gpurun_out/ab_on.json", "gpurun_out/ab_off.json"):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        print(f, {k: d.get(k) for k in ("value", "eager_ms_per_step", "ddp_hooks_ms_per_step", "ddp_proxy_ms_per_step")})
    except Exception as e:
        print(f, "error", e)
PY
