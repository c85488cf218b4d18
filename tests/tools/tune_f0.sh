#!/bin/bash
# In-chain sweep of the in-workgroup K split for RMVPE's layers (the f0 branch owns 32 CUs: 1024-thread workgroups queue there): end of the f0 branch (rm.sal)
cd "$(dirname "$0")/../.."
# tuning switches exist only in the tuning build (the product reads none of them)
export RVC_TUNING=1 RVC_LIB_OVERRIDE=$(python tests/tools/build_tuning.py | tail -1)
run() { echo -n "RVC_TUNE=$1: "; RVC_TUNE="$1" timeout 100 python tests/tools/timeline.py 1 2>&1 | tail -1 | tr " " "\n" | grep -A1 "rm.int#0\|rm.sal#0\|cv.out#0\|sy.audio" | grep -v "^--" | tr "\n" " "; echo; }
run ""
run "64,1152:0,8;128,2304:0,8;256,4608:0,8"
run "64,1152:0,4;128,2304:0,8;256,4608:0,8;32,576:0,4"
run "128,1152:0,8;256,2304:0,8;512,1536:0,8"
run "128,1152:0,4;256,2304:0,8;512,1536:0,8;64,576:0,4"
run "64,1152:0,8;128,2304:0,8;256,4608:0,8;128,1152:0,8;256,2304:0,8;512,1536:0,8"
run "512,1536:0,4"
run "512,1536:1,16"
run ""
