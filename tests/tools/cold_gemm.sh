#!/bin/bash
# GEMM launch time by where its weights live: same weights every launch (L2 / MALL warm), a 150 MB rotation (memory-side cache, not L2),
# a 600 MB rotation (HBM): what a weight prefetch into the memory-side cache could buy the one-stream chain.
cd "$(dirname "$0")/../.."
for cold in "" 150 600; do
  if [ -n "$cold" ]; then export RVC_BENCH_COLD=$cold; else unset RVC_BENCH_COLD; fi
  echo -n "rotation=${cold:-0} MB: "; RVC_FORCE_MFAST=0 python tests/gemm_microbench.py child cv_qkv,cv_o,cv_ff1,cv_ff2,enc_ff1,d128_k11,d64_k11,rm_l5x64 2>&1 | tail -1
done
