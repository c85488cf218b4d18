#!/usr/bin/env python
"""Tuning aid: the register-direct GEMM on the 64-stream layers it still serves (A/B with RVC_LIB_OVERRIDE)."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
os.environ["RVC_BENCH_B"] = "64"
from obs_rvc_amd import _native
L = _native.lib()
L.rvc_debug_conv_bench.restype = C.c_double
L.rvc_debug_conv_bench.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]
h = C.c_void_p()
assert L.rvc_create(b"/tmp", 0, C.byref(h)) == 0
out = []
for name, (M, Cin, KW, N) in {"ffn2": (768, 3072, 1, 111), "o_proj": (768, 768, 1, 111), "dec256k11": (256, 256, 11, 252), "dec256k3": (256, 256, 3, 252), "cv_conv5": (512, 512, 2, 223)}.items():
    us = L.rvc_debug_conv_bench(h, M, Cin, KW, 1, N, 20, 0)
    out.append("%s %.1fus %.1fTF" % (name, us, 2.0 * M * Cin * KW * N * 64 / us / 1e6))
print(os.environ.get("RVC_LIB_OVERRIDE", "product")[-16:], " | ".join(out))
