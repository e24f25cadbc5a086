#!/bin/bash
# Developer A/B of bench.py flags on ONE box: alternating runs, value and ms/step of each.  usage: ab_bench.sh "<flags A>" "<flags B>" [rounds]
A="$1"; B="$2"; R="${3:-2}"
Q="--no-cpu-baseline --no-forward-only --no-step-variants --no-kernel-timing"
for i in $(seq 1 $R); do
  for F in "$A" "$B"; do
    python bench.py $Q $F 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('[$F]', d['value'], 'img/s', d['ms_per_step'], 'ms')"
  done
done
