#!/bin/bash
set -u
export TMPDIR=/tmp
O=gpurun_out; mkdir -p $O
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/prof_lamb -- python3 $GRAFT_REPO_ROOT/tools/bench_lamb.py > $GRAFT_REPO_ROOT/$O/m_lamb.txt 2>&1
cd $GRAFT_REPO_ROOT
python3 tools/rocprof_summary.py $O/prof_lamb $O/m_prof.txt
head -16 $O/m_prof.txt
rm -rf $O/prof_lamb
