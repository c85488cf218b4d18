#!/bin/bash
# Would L2-warm weights make the one-stream launches faster?  Each launch reads weights that were last touched 600 MB ago (HBM), timed by its own
# dispatch events: as is (mode 3 = no touch), after a touch kernel that reads every line from any XCD (1), after one that reads every m-tile from the
# XCD that will consume it (2).
cd "$(dirname "$0")/../.."
export RVC_BENCH_COLD=600 RVC_FORCE_MFAST=${RVC_FORCE_MFAST:-1}
for t in 3 1 2; do
  echo -n "touch=$t: "; RVC_BENCH_TOUCH=$t python tests/gemm_microbench.py child cv_qkv,cv_o,cv_ff1,cv_ff2,enc_ff1,rm_l5x64 2>&1 | tail -1
done
