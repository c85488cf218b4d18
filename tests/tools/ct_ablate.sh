#!/bin/bash
# Ablation of conv_tile_kernel's main loop: variants of the library that differ in conv_tile_inst.hip only (-DRVC_CT_DBG: 1 no weight reloads,
# 2 no B reads, 8 no MFMAs), timed on the decoder's 128-channel shape through the conv bench.  build: ct_ablate.sh build ; run on the GPU: ct_ablate.sh
cd "$(dirname "$0")/../.."
if [ "$1" = build ]; then
for dbg in 0 1 2 8 3 9 10; do
python - <<PY
import os, sys
sys.path.insert(0, ".")
from obs_rvc_amd import _native as N
N.UNITS[:] = [u for u in N.UNITS if u[0] != "conv_tile_inst.hip"] + [("conv_tile_inst.hip", ["-DRVC_CT_DBG=$dbg"], ("conv_tile_inst.hip", "conv_tile.hip.h") + N._IGEMM_DEPS)]
N.link_library(N.compile_units(), os.path.join(N.CSRC, "librvc_ctdbg$dbg.so"), N.source_hash())
PY
done; exit 0; fi
export RVC_TUNING=1 RVC_CONV_TILE=2
for ks in 1 2; do for dbg in 0 1 2 8 3 9 10; do
  echo -n "KS=$ks dbg=$dbg: "; RVC_CONV_TILE_KS=$ks RVC_LIB_OVERRIDE=$PWD/obs_rvc_amd/csrc/librvc_ctdbg$dbg.so timeout 100 python tests/tools/conv_tile_check.py child 2>&1 | grep "M=128 Cin=128 KW=11 dil=1" | awk '{print $9, $10}'
done; done
