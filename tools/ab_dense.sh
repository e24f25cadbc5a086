for cfg in default qkv,dqkv,fc1,proj,dproj qkv,dqkv,fc1,fc2,dfc1 qkv,dqkv,fc1,proj,dproj,fc2,dfc1 qkv,dqkv,fc1,proj,dproj,fc2,dfc1,dfc2 default; do
  if [ $cfg = default ]; then A=""; else A="--dense-hip $cfg"; fi
  python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-forward-only $A 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$cfg', d['value'], d['ms_per_step'])"
done
