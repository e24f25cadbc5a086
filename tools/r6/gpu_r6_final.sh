#!/bin/bash
# round 6: validation of the final tree - smoke, the whole GPU suite, the default bench line, the round's profiles
set -u
O=gpurun_out; mkdir -p $O
python -c "import __graft_entry__ as g; g.build(); g.smoke()" 2>&1 | tail -2
timeout 3000 python -m pytest tests -m gpu -q 2>&1 | tail -6 > $O/gputest_r6.txt; tail -3 $O/gputest_r6.txt
( time python bench.py > $O/bench_r6_n1.json 2> $O/bench_r6_n1.err ) 2>&1 | tail -3
python3 -c "
import json
d = json.loads(open('gpurun_out/bench_r6_n1.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['step_breakdown']['timed_kernels_ms'], d['step_breakdown']['other_ms'], d.get('ddp_graph_ms_per_step'), d.get('ddp_eager_ms_per_step'), d['cpu_baseline']['value'])"
bash tools/round_profiles.sh > $O/final_profiles.log 2>&1
tail -3 $O/final_profiles.log
