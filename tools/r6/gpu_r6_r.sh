#!/bin/bash
# round 6: class-token rows as two launches - parity tests + A/B (CLS2=1 never, 2 wherever legal, unset: by shape)
set -u
timeout 900 python -m pytest tests/test_dense_gemm_gpu.py -x -q -k "per_image or launch_model" 2>&1 | tail -3
for c in 1 2; do echo "CLS2=$c"; CLS2=$c timeout 600 python tools/ab_dense_image.py 2>&1 | grep "mode 0"; done
