#!/bin/bash
# In-chain tile sweep for ContentVec's LayerNorm-consumer projections (qkv 2304 x 768, ff1 3072 x 768): end of the ContentVec branch (cv.out) per setting
cd "$(dirname "$0")/../.."
# tuning switches exist only in the tuning build (the product reads none of them)
export RVC_TUNING=1 RVC_LIB_OVERRIDE=$(python tests/tools/build_tuning.py | tail -1)
for t in "" "2304,768:0,4;3072,768:0,4" "2304,768:0,8;3072,768:0,8" "2304,768:1,4;3072,768:1,4" "2304,768:1,8;3072,768:1,8" "2304,768:2,4;3072,768:2,4" "2304,768:3,8;3072,768:3,8" "2304,768:4,4;3072,768:4,4" "2304,768:3,4;3072,768:4,4" "2304,768:1,8;3072,768:3,4"; do
  echo -n "RVC_TUNE=$t: "
  RVC_TUNE="$t" timeout 100 python tests/tools/timeline.py 1 2>&1 | tail -1 | tr " " "\n" | grep -A1 "cv.out#0\|rm.sal#0\|cv.pos#0" | grep -v "^--" | tr "\n" " "; echo
done
