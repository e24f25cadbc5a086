#!/bin/bash
# round 6: SSL trainer with the own reducer - its GPU tests and the SSL bench figure
set -u
timeout 1800 python -m pytest tests/test_ssl_gpu.py tests/test_ssl_loss_gpu.py -x -q 2>&1 | tail -4
timeout 600 python tools/bench_ssl.py 32 6 2>/dev/null | tail -1
