#!/bin/bash
# Tuning aid: the product's step time at every stream count, in ONE call on one box (figures from different boxes differ by +-1.5 %).
# usage (on the GPU box): bash tests/tools/stream_sweep.sh [counts...]
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
for S in ${@:-1 2 4 8 16 32 64}; do
  python $R/bench.py --only-headline --no-cpu --steps 30 --warmup 8 --streams $S 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$S streams: %.4f ms/step (p50 %.4f, gpu %.4f)' % (d['ms_per_step'], d['latency_ms']['p50'], d.get('gpu_ms_last_chunk', 0)))"
done
