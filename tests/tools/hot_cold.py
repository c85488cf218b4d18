#!/usr/bin/env python
"""Measurement aid (round 6): what a weight-streaming launch of RMVPE's deep levels costs with its weights HOT (the same layer repeated: the panel stays in the
L2s) -- against the 6.5-8.5 us the per-layer table shows for it in the chain, where every panel is cold (853 MB of weights pass between two uses).  Needs the
tuning library (rvc_debug_conv_bench).  usage: RVC_TUNING=1 RVC_LIB_OVERRIDE=obs_rvc_amd/csrc/librvc_tuning.so python tests/tools/hot_cold.py"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
from obs_rvc_amd import _native
L = _native.lib()
L.rvc_debug_conv_bench.restype = C.c_double
L.rvc_debug_conv_bench.argtypes = [C.c_void_p] + [C.c_int] * 9
h = C.c_void_p()
assert L.rvc_create(b"/tmp", 0, C.byref(h)) == 0
# (M, Cin, KW, N): the GEMM shapes of the deep levels at one stream (K = Cin * KW)
for label, M, Cin, KW, N in [("bottleneck 512 x 1536, 4 px", 512, 512, 3, 4), ("level 4: 256 x 2304, 16 px", 256, 256, 9, 16), ("level 3: 128 x 1152, 64 px", 128, 128, 9, 64),
                             ("level 4 first: 256 x 1152, 16 px", 256, 128, 9, 16), ("decoder 4: 256 x 4608, 16 px", 256, 512, 9, 16)]:
    us = [L.rvc_debug_conv_bench(h, M, Cin, KW, 1, N, 40, 0, 1, 1) for _ in range(3)]
    print("%-36s hot: %s us per launch (%.2f MB of weights)" % (label, " ".join("%.2f" % u for u in us), M * Cin * KW * 4 / 1e6), flush=True)
