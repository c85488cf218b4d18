cd /root/repo
for cold in "" 600; do
for cfg in auto 3,4 3,8 4,4 0,4 0,8 0,16 1,4 1,8 2,4; do
  if [ "$cfg" = auto ]; then unset RVC_FORCE_CFG; else export RVC_FORCE_CFG=$cfg; fi
  if [ -n "$cold" ]; then export RVC_BENCH_COLD=$cold; else unset RVC_BENCH_COLD; fi
  echo -n "cold=${cold:-0} "; RVC_FORCE_MFAST=0 python tests/gemm_microbench.py child cv_qkv,cv_o,cv_ff1,cv_ff2,enc_ff1 2>&1 | tail -1
done; done
