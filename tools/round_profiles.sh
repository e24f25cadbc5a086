#!/bin/bash
# Developer recipe: the measurements behind profiles/*_r6* (run on the GPU box from the repo root).
set -u
export TMPDIR=/tmp
O=gpurun_out
B="python3 bench.py --no-cpu-baseline --no-forward-only --no-kernel-timing --no-graph"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_trace -- python3 bench.py --no-cpu-baseline --no-forward-only --no-step-variants --steps 10 --warmup 3 > $O/prof_trace.json 2> $O/prof_trace.log
python3 tools/rocprof_summary.py $O/prof_trace $O/rocprof_r6_bench.txt
python3 tools/step_gaps.py $O/prof_trace $O/step_gaps_r6.txt > /dev/null
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -- $B --steps 3 --warmup 2 > /dev/null 2> $O/pmc_fetch.log
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -- $B --steps 3 --warmup 2 > /dev/null 2> $O/pmc_write.log
python3 tools/traffic_from_pmc.py $O/pmc_fetch $O/pmc_write $O/traffic_r6.json > $O/traffic_r6.txt
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $O/pmc_sq -- $B --steps 3 --warmup 2 > /dev/null 2> $O/pmc_sq.log
python3 tools/pmc_sq_summary.py $O/pmc_sq $O/pmc_r6_sq.txt > /dev/null
# BASELINE configs[4] on one GPU: the DINOv2 student / teacher step (kernel list + two timings)
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_ssl -- python3 tools/bench_ssl.py 32 4 > $O/bench_r6_ssl.txt 2> $O/prof_ssl.log
python3 tools/rocprof_summary.py $O/prof_ssl $O/rocprof_r6_ssl.txt
python3 tools/bench_ssl.py 32 8 2> /dev/null | tail -1 >> $O/bench_r6_ssl.txt
rm -rf $O/prof_trace $O/pmc_fetch $O/pmc_write $O/pmc_sq $O/prof_ssl
ls -la $O | tail -15
