#!/bin/bash
# Round profiles: rocprofv3 kernel stats of the three single-GPU configurations, the FETCH_SIZE / WRITE_SIZE passes and the SQ / TCC
# passes (separate runs, --kernel-trace only, as gpurun requires).  Raw traces stay in /tmp on the GPU box (hundreds of MB); the
# summaries the repository keeps are written to gpurun_out/<tag>/ under their profiles/ names.
# usage: profile_round.sh <tag> [round]      e.g. profile_round.sh r02d r02
tag=${1:-r06}; rnd=${2:-r06}; R=$GRAFT_REPO_ROOT; out=$R/gpurun_out/$tag; raw=/tmp/prof_$tag; mkdir -p $out $raw
cd /tmp; export TMPDIR=/tmp
prof() { name=$1; shift; rocprofv3 --kernel-trace --stats --output-format csv -d $raw/ks_$name -- python $R/bench.py --only-headline --no-cpu "$@" > $out/bench_$name.json 2> $out/bench_$name.err
         cp $raw/ks_$name/*/*kernel_stats.csv $out/${rnd}_kernel_stats_bench_$name.csv
         python $R/tests/tools/trace_stats.py $raw/ks_$name/*/*kernel_trace.csv $out/${rnd}_kernel_stats_bench_${name}_steps.csv; }
# (two tables per pass since round 6: rocprofv3's own --stats table of the whole process, which includes the plan build's autotune trials, and the same trace
#  restricted to whole steps -- tests/tools/trace_stats.py; the serial passes and every per-step figure use the second)
if [ -z "$RVC_PROFILE_ONLY_SERIAL" ]; then
prof 1stream --steps 100
prof index100k --steps 100 --index
prof 64streams --steps 15 --warmup 3 --streams 64
# the two front branches issued one after the other (bench.py --serial-branches = the RVC_SERIAL_BRANCHES test hook through rvc_debug_option; the product
# build reads no such variable from the environment): every kernel's own duration at many streams.  serial_pass.py writes the sidecar that bench.py checks
# next to its own in-run roofline.frac of a many-stream configuration (events / rocprof: the validation of the event method).
rocprofv3 --kernel-trace --stats --output-format csv -d $raw/ks_64serial -- python $R/bench.py --only-headline --no-cpu --steps 15 --warmup 3 --streams 64 --serial-branches > $out/bench_64streams_serial.json 2> $out/bench_64streams_serial.err
cp $raw/ks_64serial/*/*kernel_stats.csv $out/${rnd}_kernel_stats_bench_64streams_serial_branches.csv
python $R/tests/tools/trace_stats.py $raw/ks_64serial/*/*kernel_trace.csv $out/${rnd}_kernel_stats_bench_64streams_serial_branches_steps.csv
python $R/tests/tools/serial_pass.py $out/${rnd}_kernel_stats_bench_64streams_serial_branches_steps.csv 64 $out/${rnd}_serial_64streams.json ${rnd}_kernel_stats_bench_64streams_serial_branches_steps.csv $out/bench_64streams_serial.json > /dev/null
fi
# the same pass at the stream counts between the two regimes (validation of the event method at those counts); RVC_PROFILE_ONLY_SERIAL=1 stops here
for S in 8 16 32; do
    rocprofv3 --kernel-trace --stats --output-format csv -d $raw/ks_${S}serial -- python $R/bench.py --only-headline --no-cpu --steps 20 --warmup 3 --streams $S --serial-branches > $out/bench_${S}streams_serial.json 2> $out/bench_${S}streams_serial.err
    cp $raw/ks_${S}serial/*/*kernel_stats.csv $out/${rnd}_kernel_stats_bench_${S}streams_serial_branches.csv
    python $R/tests/tools/trace_stats.py $raw/ks_${S}serial/*/*kernel_trace.csv $out/${rnd}_kernel_stats_bench_${S}streams_serial_branches_steps.csv
    python $R/tests/tools/serial_pass.py $out/${rnd}_kernel_stats_bench_${S}streams_serial_branches_steps.csv $S $out/${rnd}_serial_${S}streams.json ${rnd}_kernel_stats_bench_${S}streams_serial_branches_steps.csv $out/bench_${S}streams_serial.json > /dev/null
done
if [ -n "$RVC_PROFILE_ONLY_SERIAL" ]; then ls -la $out; exit 0; fi
pmc() { name=$1; ctr=$2; shift 2; rocprofv3 --kernel-trace --pmc $ctr --output-format csv -d $raw/pmc_${name}_$ctr -- python $R/bench.py --only-headline --no-cpu "$@" > /dev/null 2>&1; }
pmc 1stream FETCH_SIZE --steps 40
pmc 1stream WRITE_SIZE --steps 40
pmc index100k FETCH_SIZE --steps 40 --index
pmc index100k WRITE_SIZE --steps 40 --index
pmc 64streams FETCH_SIZE --steps 6 --warmup 2 --streams 64
pmc 64streams WRITE_SIZE --steps 6 --warmup 2 --streams 64
# one file per configuration; bench.py takes roofline.traffic only from the pass of exactly its own configuration and build
python $R/tests/tools/pmc_traffic.py $raw/pmc_1stream_FETCH_SIZE/*/*counter_collection.csv $raw/pmc_1stream_WRITE_SIZE/*/*counter_collection.csv $out/${rnd}_pmc_traffic_1stream.json "bench.py --only-headline (1 stream, retrieval off):" '{"streams": 1, "index": false, "version": 2, "preset": "full"}' 852778176 > /dev/null
python $R/tests/tools/pmc_traffic.py $raw/pmc_index100k_FETCH_SIZE/*/*counter_collection.csv $raw/pmc_index100k_WRITE_SIZE/*/*counter_collection.csv $out/${rnd}_pmc_traffic.json "bench.py --only-headline --index (1 stream + 100k x 768 index):" '{"streams": 1, "index": true, "version": 2, "preset": "full"}' 852778176 > /dev/null
python $R/tests/tools/pmc_traffic.py $raw/pmc_64streams_FETCH_SIZE/*/*counter_collection.csv $raw/pmc_64streams_WRITE_SIZE/*/*counter_collection.csv $out/${rnd}_pmc_traffic_64streams.json "bench.py --only-headline --streams 64:" '{"streams": 64, "index": false, "version": 2, "preset": "full"}' 13.9e9 > /dev/null
# per-layer roofline table (HIP events of the dispatches themselves, branches serialised): M, N, K, kernel, us, TF/s, fraction of the fp32 MFMA peak
python $R/tests/tools/op_profile.py 1 full --json $out/${rnd}_layers_1streams.json > $out/op_profile_1.txt 2>&1
python $R/tests/tools/op_profile.py 8 full --json $out/${rnd}_layers_8streams.json > $out/op_profile_8.txt 2>&1
python $R/tests/tools/op_profile.py 64 full --json $out/${rnd}_layers_64streams.json > $out/op_profile_64.txt 2>&1
# SQ / TCC counters per kernel instantiation (two passes each: 8 SQ counters, then TCC + GRBM) -> tests/tools/pmc_summary.py
SQ="SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA"
pmcs() { name=$1; shift; rocprofv3 --kernel-trace --pmc $SQ --output-format csv -d $raw/pmcs_${name}_a -- python $R/bench.py --only-headline --no-cpu "$@" > /dev/null 2>&1
         rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum GRBM_GUI_ACTIVE --output-format csv -d $raw/pmcs_${name}_b -- python $R/bench.py --only-headline --no-cpu "$@" > /dev/null 2>&1; }
pmcs 1stream --steps 40
pmcs 64streams --steps 6 --warmup 2 --streams 64
python $R/tests/tools/pmc_summary.py $out/${rnd}_pmc_igemm_summary.json "bench_1stream=$(ls $raw/pmcs_1stream_a/*/*counter_collection.csv),$(ls $raw/pmcs_1stream_b/*/*counter_collection.csv)" \
    "bench_64streams=$(ls $raw/pmcs_64streams_a/*/*counter_collection.csv),$(ls $raw/pmcs_64streams_b/*/*counter_collection.csv)" > $out/pmc_summary.txt
ls -la $out
