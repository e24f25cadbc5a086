# developer A/B (run on the GPU box): residual tail of proj / fc2 fused into the GEMM epilogue vs the separate row kernel
for cfg in "" "--resid-fused" "" "--resid-fused" ""; do
  python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-forward-only $cfg 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('[$cfg]', d['value'], d['ms_per_step'])"
done
