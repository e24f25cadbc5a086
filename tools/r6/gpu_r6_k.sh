#!/bin/bash
# A/B/C in alternating processes: class-token rows of the K = 3840 / 5120 launches as two launches (default) | single launch
# (OCTIC_CLS2=1: K = 5120 then stays on classic panels) | no per-image panels at all (OCTIC_NO_IMAGE=2)
set -u
O=gpurun_out; mkdir -p $O
S="--no-cpu-baseline --no-ssl-side --no-forward-only --no-step-variants --no-kernel-timing"
for i in 1 2 3; do
timeout 900 python bench.py $S > $O/k_a.json 2> $O/k_a.err
OCTIC_CLS2=1 timeout 900 python bench.py $S > $O/k_c.json 2> $O/k_c.err
OCTIC_NO_IMAGE=2 timeout 900 python bench.py $S > $O/k_b.json 2> $O/k_b.err
python3 -c "
import json
r = lambda f: json.loads(open(f).read().strip().splitlines()[-1])['ms_per_step']
print('two-launch', r('gpurun_out/k_a.json'), 'single', r('gpurun_out/k_c.json'), 'classic', r('gpurun_out/k_b.json'))"
done
