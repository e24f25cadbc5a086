#!/bin/bash
set -u
O=gpurun_out; mkdir -p $O
timeout 1500 python -m pytest tests/test_dispatch_gpu.py -x -q 2>&1 | tail -12
timeout 1500 python tools/compile_bench.py 2>&1 | grep -v Warning | tail -6
