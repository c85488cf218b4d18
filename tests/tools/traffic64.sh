#!/bin/bash
# One-off of round 5: re-take ONLY the 64-stream FETCH_SIZE / WRITE_SIZE passes (after a kernel class list changed) and the default bench line; the full
# evidence set is tests/tools/profile_round.sh.
R=$GRAFT_REPO_ROOT; raw=/tmp/prof_t64; out=$R/gpurun_out/r05j; mkdir -p $raw $out; cd /tmp; export TMPDIR=/tmp
pmc() { name=$1; ctr=$2; shift 2; rocprofv3 --kernel-trace --pmc $ctr --output-format csv -d $raw/pmc_${name}_$ctr -- python $R/bench.py --only-headline --no-cpu "$@" > /dev/null 2>&1; }
pmc 64streams FETCH_SIZE --steps 6 --warmup 2 --streams 64
pmc 64streams WRITE_SIZE --steps 6 --warmup 2 --streams 64
python $R/tests/tools/pmc_traffic.py $raw/pmc_64streams_FETCH_SIZE/*/*counter_collection.csv $raw/pmc_64streams_WRITE_SIZE/*/*counter_collection.csv $out/r05_pmc_traffic_64streams.json "bench.py --only-headline --streams 64:" '{"streams": 64, "index": false, "version": 2, "preset": "full"}' 13.9e9 > /dev/null
cp $out/r05_pmc_traffic_64streams.json $R/profiles/
cd $R; python bench.py > $out/bench_default.json 2> $out/bench_default.err; tail -c 200 $out/bench_default.json
