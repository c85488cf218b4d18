#!/bin/bash
# ContentVec stem at one stream: thresholds of the workgroup-tiled kernels (the stem's 512 x 1536 convolutions are the only layers large enough): cv.feat = end of the stem
cd "$(dirname "$0")/../.."
# tuning switches exist only in the tuning build (the product reads none of them)
export RVC_TUNING=1 RVC_LIB_OVERRIDE=$(python tests/tools/build_tuning.py | tail -1)
run() { echo -n "$*: "; env "$@" timeout 100 python tests/tools/timeline.py 1 2>&1 | tail -1 | tr " " "\n" | grep -A1 "cv.feat#0\|cv.out#0\|rm.sal#0\|sy.audio" | grep -v "^--" | tr "\n" " "; echo; }
run RVC_X=1
run RVC_G32_NARROW=200
run RVC_G32_NARROW=100
run RVC_GEMM32_MIN=100
run RVC_GEMM32_MIN=200 RVC_G32_NARROW=100
run RVC_X=1
