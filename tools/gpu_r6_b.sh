#!/bin/bash
# round 6, call B: bucket-size A/B of the captured data-parallel step + kernel trace of it
set -u
export TMPDIR=/tmp
O=gpurun_out
mkdir -p $O
timeout 600 python -m pytest tests/test_train_gpu.py -x -q -k "captured_data_parallel" 2>&1 | tail -3
B="python3 bench.py --force-ddp --steps 20 --warmup 4 --no-cpu-baseline --no-forward-only --no-ssl-side --no-step-variants --no-kernel-timing"
for mb in 48 96 200 1500; do
  timeout 600 $B --bucket-mb $mb > $O/b_ddp_$mb.json 2> $O/b_ddp_$mb.err
  python3 - $O/b_ddp_$mb.json $mb <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("bucket_mb", sys.argv[2], d["ms_per_step"], d["host_issue_ms_per_step"], d["config"]["gradient_reduction"])
PY
done
timeout 600 $B --no-graph > $O/b_ddp_eager.json 2> $O/b_ddp_eager.err
python3 -c "
import json
d = json.loads(open('$O/b_ddp_eager.json').read().strip().splitlines()[-1]); print('eager reducer', d['ms_per_step'], d['host_issue_ms_per_step'])"
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/prof_ddp -- python3 $GRAFT_REPO_ROOT/bench.py --force-ddp --steps 10 --warmup 3 --no-cpu-baseline --no-forward-only --no-ssl-side --no-step-variants --no-kernel-timing > $GRAFT_REPO_ROOT/$O/b_prof.json 2> $GRAFT_REPO_ROOT/$O/b_prof.log
cd $GRAFT_REPO_ROOT
python3 tools/rocprof_summary.py $O/prof_ddp $O/rocprof_r6_ddp.txt
head -40 $O/rocprof_r6_ddp.txt
python3 tools/step_gaps.py $O/prof_ddp $O/step_gaps_r6_ddp.txt | tail -20
rm -rf $O/prof_ddp
