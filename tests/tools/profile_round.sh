#!/bin/bash
# Round profiles: rocprofv3 kernel stats of the three single-GPU configurations + the FETCH_SIZE / WRITE_SIZE passes (separate runs,
# --kernel-trace only, as gpurun requires).  usage: profile_round.sh <tag>   -> gpurun_out/<tag>/...
tag=${1:-r02}; R=$GRAFT_REPO_ROOT; out=$R/gpurun_out/$tag; mkdir -p $out
cd /tmp; export TMPDIR=/tmp
prof() { name=$1; shift; rocprofv3 --kernel-trace --stats --output-format csv -d $out/ks_$name -- python $R/bench.py --only-headline --no-cpu "$@" > $out/bench_$name.json 2> $out/bench_$name.err; }
prof 1stream --steps 100
prof index100k --steps 100 --index
prof 64streams --steps 15 --warmup 3 --streams 64
pmc() { name=$1; ctr=$2; shift 2; rocprofv3 --kernel-trace --pmc $ctr --output-format csv -d $out/pmc_${name}_$ctr -- python $R/bench.py --only-headline --no-cpu "$@" > /dev/null 2>&1; }
pmc index100k FETCH_SIZE --steps 40 --index
pmc index100k WRITE_SIZE --steps 40 --index
pmc 64streams FETCH_SIZE --steps 6 --warmup 2 --streams 64
pmc 64streams WRITE_SIZE --steps 6 --warmup 2 --streams 64
ls $out
