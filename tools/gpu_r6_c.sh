#!/bin/bash
set -u
O=gpurun_out; mkdir -p $O
timeout 900 python -m pytest tests/test_dense_gemm_gpu.py -x -q -k "per_image or launch_model" 2>&1 | tail -15
timeout 600 python tools/ab_dense_image.py 2>&1 | tail -12
timeout 900 python bench.py --no-cpu-baseline --no-ssl-side --no-forward-only --no-step-variants > $O/c_bench.json 2> $O/c_bench.err
python3 -c "
import json
d = json.loads(open('$O/c_bench.json').read().strip().splitlines()[-1]); print('bench', d['value'], d['ms_per_step'], d['loss'])"
