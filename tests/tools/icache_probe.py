#!/usr/bin/env python
"""Tuning aid: what a launch pays when its code is not in the instruction cache.  Each target layer is timed by its own dispatch events
back to back with itself, and again with launches of other kernel instantiations (small, L2-resident data) in between."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
os.environ["RVC_BENCH_EVICT"] = sys.argv[1] if len(sys.argv) > 1 else "6"
from obs_rvc_amd import _native
L = _native.lib()
L.rvc_debug_conv_bench.restype = C.c_double
L.rvc_debug_conv_bench.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]
h = C.c_void_p()
assert L.rvc_create(b"/tmp", 0, C.byref(h)) == 0
for name, (M, Cin, KW, dil, N) in {"cv_qkv": (2304, 768, 1, 1, 111), "cv_o": (768, 768, 1, 1, 111), "cv_ff2": (768, 3072, 1, 1, 111), "hg3_k11": (32, 32, 11, 1, 10080),
                                    "hg1_k7": (128, 128, 7, 3, 2520), "enc_ff1": (768, 192, 3, 1, 21), "rm_deep": (512, 512, 3, 1, 256)}.items():
    sys.stdout.write("%-8s " % name); sys.stdout.flush()
    L.rvc_debug_conv_bench(h, M, Cin, KW, dil, N, 30, 0)
