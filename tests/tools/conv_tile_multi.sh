cd "$(dirname "$0")/../.."
export RVC_BENCH_B=${RVC_BENCH_B:-64}
S=d128_k3,d128_k7,d128_k11,d128_k11d5,d64_k3,d64_k7,d64_k11,d64_k11d5,d32_k3,d32_k7,d32_k11,d32_k11d5
for v in "RVC_CONV_TILE_MULTI=0" "RVC_CONV_TILE_MULTI=2 RVC_CONV_TILE_KS=1" "RVC_CONV_TILE_MULTI=2 RVC_CONV_TILE_KS=2" "RVC_CONV_TILE_MULTI=2 RVC_CONV_TILE_KS=1 RVC_CONV_TILE_W128=7" "RVC_CONV_TILE_MULTI=2 RVC_CONV_TILE_KS=2 RVC_CONV_TILE_W128=7"; do
  echo "== $v"
  env $v python tests/gemm_microbench.py child $S 2>&1 | tail -1 | tr '|' '\n'
done
