#!/bin/bash
# usage: ab_env.sh "ENV=1" "ENV2=x" ...   -- headline ms per chunk for each setting, twice, alternating (same box)
cd "$(dirname "$0")/../.."
# tuning switches exist only in the tuning build (the product reads none of them)
export RVC_TUNING=1 RVC_LIB_OVERRIDE=$(python tests/tools/build_tuning.py | tail -1)
for i in 1 2; do
for v in "$@"; do
  echo -n "$v: "; env "$v" RVC_BENCH_SOAK=0 timeout 150 python bench.py --only-headline --no-cpu --steps 300 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print(d['ms_per_step'], d['latency_ms']['p50'], d['latency_ms']['p99'])"
done; done
