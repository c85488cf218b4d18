#!/bin/bash
# Tuning aid: in-situ tile sweep of the four ContentVec GEMM shapes at one stream (RVC_TUNE), plus the f0 partition size.
out=gpurun_out/${1:-sweep}; mkdir -p $out
run() { # name, env...
  name=$1; shift
  env "$@" python tests/tools/op_profile.py 1 > $out/$name.txt 2>&1
  echo "== $name $*"; grep -E "wall ms|M=3072 N=111 K=768|M=2304 N=111 K=768|M=768 N=111 K=3072|M=768 N=111 K=768" $out/$name.txt | awk '{k=$NF; $1=$1; print}' | sort | uniq -c | sort -rn | head -8
}
run base X=1
i=0
for v in "3,8;3,8;0,8;0,8" "4,4;4,4;0,16;1,4" "4,8;4,8;1,8;1,8" "1,4;1,4;1,16;2,4" "2,4;2,4;2,8;2,8" "2,8;2,8;2,16;3,4" "0,4;0,4;3,8;3,8" "3,16;3,16;3,16;0,16" "4,16;4,16;4,16;1,16"; do
  IFS=';' read a b c d <<< "$v"
  run tune$i RVC_TUNE="3072,768:$a;2304,768:$b;768,3072:$c;768,768:$d"
  i=$((i+1))
done
for n in 48 64 96; do run f0cus$n RVC_F0_CUS=$n; done
run nomask RVC_NO_CUMASK=1
