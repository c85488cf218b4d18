#!/bin/bash
# Round 6: HBM / fabric read traffic of a 64-stream step with the panel order of igemm32l_kernel on (default) and off (test hook RVC_G32L_PANEL=0), the plans
# rule-based in both (no autotune: the same kernels in both arms).  rocprofv3 --pmc FETCH_SIZE passes, --kernel-trace only (as gpurun requires).
# usage (gpurun): bash tests/tools/traffic_ab.sh   -> gpurun_out/traffic_ab/{panel1,panel0}.json
R=$GRAFT_REPO_ROOT; raw=/tmp/prof_tab; out=$R/gpurun_out/traffic_ab; mkdir -p $raw $out; cd /tmp; export TMPDIR=/tmp
for v in 1 0; do
  rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $raw/f$v -- python $R/bench.py --only-headline --no-cpu --no-calibration --no-autotune --steps 6 --warmup 2 --streams 64 --hook RVC_G32L_PANEL=$v > $out/bench_$v.json 2> $out/bench_$v.err
  python $R/tests/tools/pmc_traffic.py $raw/f$v/*/*counter_collection.csv - $out/panel$v.json "bench.py --only-headline --streams 64 --no-autotune --hook RVC_G32L_PANEL=$v:" '{"streams": 64, "index": false, "version": 2, "preset": "full", "hook": "RVC_G32L_PANEL='$v'"}' 13.9e9 > /dev/null
done
python - <<EOF
import json
for v in (1, 0):
    d = json.load(open("$out/panel%d.json" % v))
    ig = d["igemm_all_instantiations"]
    print("RVC_G32L_PANEL=%d: %.2f GB read per step (implicit-GEMM class)" % (v, ig["hbm_read_bytes_per_chunk"] / 1e9))
    for k in d["by_kernel"][:8]:
        print("   %-48s wgs %6d x%5.1f  %8.1f MB/launch" % (k["kernel"][:48], k["workgroups"], k["launches_per_chunk"], k["read_mb_per_launch"]))
EOF
