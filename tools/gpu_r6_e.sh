#!/bin/bash
set -u
O=gpurun_out; mkdir -p $O
timeout 1800 python -m pytest tests/test_dense_gemm_gpu.py tests/test_dense_gpu.py tests/test_vith_gpu.py -x -q 2>&1 | tail -5
for i in 1 2; do
timeout 900 python bench.py --no-cpu-baseline --no-ssl-side --no-forward-only --no-step-variants --no-kernel-timing > $O/e_bench.json 2> $O/e_bench.err
python3 -c "
import json
d = json.loads(open('$O/e_bench.json').read().strip().splitlines()[-1]); print('bench', d['value'], d['ms_per_step'], d['loss'])"
OCTIC_NO_IMAGE=1 timeout 900 python bench.py --no-cpu-baseline --no-ssl-side --no-forward-only --no-step-variants --no-kernel-timing > $O/e_bench0.json 2> $O/e_bench0.err
python3 -c "
import json
d = json.loads(open('$O/e_bench0.json').read().strip().splitlines()[-1]); print('bench classic panels', d['value'], d['ms_per_step'], d['loss'])"
done
