#!/bin/bash
set -u
O=gpurun_out; mkdir -p $O
OCTIC_LAMB_DEBUG=1 timeout 600 python -m pytest tests/test_train_gpu.py -x -q -k "lamb" 2>&1 | grep -v "^$" | tail -25
OCTIC_LAMB_DEBUG=1 timeout 600 python tools/bench_lamb.py 2>&1 | tail -7
