#!/bin/bash
# usage: ab_streams.sh "<streams...>" "ENV=1" ...  -- ms per step of bench.py --only-headline --streams S for each setting
cd "$(dirname "$0")/../.."
# tuning switches exist only in the tuning build (the product reads none of them)
export RVC_TUNING=1 RVC_LIB_OVERRIDE=$(python tests/tools/build_tuning.py | tail -1)
S="$1"; shift
for st in $S; do for v in "$@"; do
  echo -n "streams=$st $v: "; env "$v" RVC_BENCH_SOAK=0 timeout 200 python bench.py --only-headline --no-cpu --streams $st --steps 60 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print(d['ms_per_step'])"
done; done
