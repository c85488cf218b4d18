#!/usr/bin/env python
"""Tuning aid (needs the tuning library: RVC_TUNING=1 RVC_LIB_OVERRIDE=obs_rvc_amd/csrc/librvc_tuning.so): isolated timings of the many-stream
GEMM shapes under the planner's own choice and under every wide-register-tile instantiation of igemm32_kernel (RVC_G32W = 9 .. 12).

usage: g32w_sweep.py [streams ...]        default: 64 32 16 8"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
from obs_rvc_amd import _native
L = _native.lib()
L.rvc_debug_conv_bench.restype = C.c_double
L.rvc_debug_conv_bench.argtypes = [C.c_void_p] + [C.c_int] * 9
h = C.c_void_p()
assert L.rvc_create(b"/tmp", 0, C.byref(h)) == 0
# (label, M, Cin, KW, dil, N per stream, act)
SHAPES = [("cv ffn1 3072x768", 3072, 768, 1, 1, 111, 3), ("cv qkv 2304x768", 2304, 768, 1, 1, 111, 0), ("cv ffn2 768x3072", 768, 3072, 1, 1, 111, 0),
          ("cv out 768x768", 768, 768, 1, 1, 111, 0), ("cv conv1 512 k3 N=3583", 512, 512, 3, 1, 3583, 3), ("cv conv3 512 k3 N=895", 512, 512, 3, 1, 895, 3),
          ("dec 256 k11 N=210", 256, 256, 11, 1, 210, 0), ("dec 128 k11 N=2520", 128, 128, 11, 1, 2520, 0), ("dec 128 k3 N=2520", 128, 128, 3, 1, 2520, 0),
          ("dec 64 k11 N=5040", 64, 64, 11, 1, 5040, 0)]
VARIANTS = [("auto", None), ("256x128", "9,1"), ("256x256", "10,1"), ("128x256", "11,1"), ("128x256b", "12,1")]
for S in [int(a) for a in sys.argv[1:]] or [64, 32, 16, 8]:
    print("streams %d: us (TF/s) per variant" % S)
    print("%-26s" % "layer" + "".join("%-20s" % v[0] for v in VARIANTS))
    for label, M, Cin, KW, dil, N, act in SHAPES:
        cells = []
        for name, val in VARIANTS:
            if val is None:
                os.environ.pop("RVC_G32W", None)
            else:
                os.environ["RVC_G32W"] = val
            us = L.rvc_debug_conv_bench(h, M, Cin, KW, dil, N, 12 if S >= 32 else 30, 0, S, act)
            cells.append("%8.1f (%5.1f)     " % (us, 2.0 * M * Cin * KW * N * S / us / 1e6))
        print("%-26s" % label + "".join(cells), flush=True)
