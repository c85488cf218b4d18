cd /root/repo
for i in 1 2 3; do
for v in 0 1; do
  echo -n "RVC_CONV_TILE=$v: "; RVC_CONV_TILE=$v RVC_BENCH_SOAK=0 python bench.py --only-headline --no-cpu --steps 300 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print(d['ms_per_step'], d['latency_ms']['p50'], d['latency_ms']['p99'])"
done; done
