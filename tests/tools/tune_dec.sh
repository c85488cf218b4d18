#!/bin/bash
# Tuning aid: tile sweep of the fused HiFiGAN ResBlock launches at one stream (RVC_TUNE matches M and the LONGEST phase's K)
out=gpurun_out/${1:-sweepdec}; mkdir -p $out
run() { name=$1; shift; env "$@" python tests/tools/op_profile.py 1 > $out/$name.txt 2>&1; echo "== $name $*"; grep -E "wall ms" $out/$name.txt; grep "nph=3" $out/$name.txt | awk '{print $1, $(NF-9), $(NF-8), $(NF-7), $(NF-6), $(NF-5)}' | sort | uniq -c | sort -k2 -n | awk '{s+=$1*$2; print} END {print "sum_us", s}' | tail -5; }
run base X=1
for v in "4,4" "4,1" "3,8" "4,8" "3,1" "2,4" "1,4"; do
  run t_$v RVC_TUNE="256,2816:$v;128,1408:$v;64,704:$v;32,352:$v"
done
