#!/bin/bash
set -u
O=gpurun_out; mkdir -p $O
timeout 900 python -m pytest tests/test_train_gpu.py -x -q -k "captured or accum or ddp or two_rank" 2>&1 | tail -3
S="--no-cpu-baseline --no-ssl-side --no-forward-only --no-step-variants"
ok=0; bad=0
for i in 1 2 3 4 5 6 7 8 9 10 11 12 13 14 15 16; do
timeout 600 python bench.py $S --force-ddp > $O/o_run.json 2> $O/o_run.err
rc=$?
if [ $rc -eq 0 ]; then ok=$((ok+1)); else bad=$((bad+1)); echo "run $i rc=$rc: $(grep -o 'watchdog thread terminated[^:]*' $O/o_run.err | head -1) $(grep '^\[bench' $O/o_run.err | tail -1)"; fi
done
echo "ok $ok bad $bad"
python3 -c "
import json
d = json.loads(open('gpurun_out/o_run.json').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['config']['launch'], d['config']['gradient_reduction'], d.get('eager_reducer_ms_per_step'))"
