cd /root/repo
python tests/tools/timeline.py 1 2>&1 | tail -30
for c in 24 32 40 48; do echo -n "RVC_F0_CUS=$c: "; RVC_F0_CUS=$c RVC_BENCH_SOAK=0 python bench.py --only-headline --no-cpu --steps 300 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print(d['ms_per_step'], d['latency_ms']['p50'])"; done
