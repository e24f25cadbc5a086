#!/bin/bash
set -u
O=gpurun_out; mkdir -p $O
( time python bench.py > $O/f_bench_full.json 2> $O/f_bench_full.err ) 2>&1 | tail -3
python3 - <<'PY'
import json
d = json.loads(open('gpurun_out/f_bench_full.json').read().strip().splitlines()[-1])
print({k: v for k, v in d.items() if k not in ("kernels", "config", "extra", "roofline", "roofline_hbm_kernel", "roofline_top_handwritten")})
print(d["roofline"])
print(list(d["kernels"].items())[:6])
PY
bash tools/round_profiles.sh > $O/f_profiles.log 2>&1
tail -5 $O/f_profiles.log
