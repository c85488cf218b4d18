cd "$(dirname "$0")/../.."
# tuning switches exist only in the tuning build (the product reads none of them)
export RVC_TUNING=1 RVC_LIB_OVERRIDE=$(python tests/tools/build_tuning.py | tail -1)
for i in 1 2; do
for v in "RVC_X=1" "RVC_TUNE=768,3072:1,8" "RVC_TUNE=768,3072:1,8;768,768:1,8" "RVC_TUNE=768,3072:1,8;768,768:1,8;2304,768:0,4" "RVC_TUNE=768,3072:1,16;768,768:1,8" "RVC_TUNE=768,3072:2,8;768,768:1,8"; do
  echo -n "$v: "; env "$v" RVC_BENCH_SOAK=0 python bench.py --only-headline --no-cpu --steps 300 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print(d['ms_per_step'], d['latency_ms']['p50'])"
done; done
