#!/usr/bin/env python
"""Tuning aid: the throughput-mode GEMM (igemm32) on the 64-stream layer shapes, K swept so that the per-launch time separates into
a slope (main loop, us per K) and an intercept (prologue + epilogue + tail).   usage: g32_sweep.py [act]"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
from obs_rvc_amd import _native
L = _native.lib()
L.rvc_debug_conv_bench.restype = C.c_double
L.rvc_debug_conv_bench.argtypes = [C.c_void_p] + [C.c_int] * 9
h = C.c_void_p()
assert L.rvc_create(b"/tmp", 0, C.byref(h)) == 0
# (label, M, KW, N per stream, Cin list)
for label, M, KW, N, cins in (("cv conv (M=512, k=3, N=64x3583)", 512, 3, 3583, (128, 256, 512)), ("ffn1 (M=3072, k=1, N=64x111)", 3072, 1, 111, (256, 768, 1536)),
                              ("dec (M=128, k=11, N=64x2520)", 128, 11, 2520, (32, 64, 128)), ("dec (M=64, k=7, N=64x5040)", 64, 7, 5040, (32, 64, 128))):
    pts = []
    for cin in cins:
        us = L.rvc_debug_conv_bench(h, M, cin, KW, 1, N, 10, 0, 64, int(sys.argv[1]) if len(sys.argv) > 1 else 0)
        K = cin * KW
        pts.append((K, us))
        print("%-34s K=%5d %9.1f us %6.1f TF/s" % (label, K, us, 2.0 * M * K * N * 64 / us / 1e6))
    (k0, t0), (k1, t1) = pts[0], pts[-1]
    slope = (t1 - t0) / (k1 - k0)
    print("%-34s slope %.4f us/K = %.1f TF/s (%.0f%% of peak), intercept %.1f us" % ("", slope, 2.0 * M * N * 64 / slope / 1e6, 2.0 * M * N * 64 / slope / 1e6 / 157.3 * 100, t0 - slope * k0))
