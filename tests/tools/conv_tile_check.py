#!/usr/bin/env python
"""Tuning aid: conv_tile_kernel against the register-direct kernel on the decoder's shapes -- error vs a double-precision host evaluation
(rvc_debug_conv_check) and back-to-back launch time (rvc_debug_conv_bench), RVC_CONV_TILE=0 / 2 in child processes."""
import ctypes as C, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
SHAPES = [(128, 128, 11, 1, 2520, 1), (128, 128, 7, 3, 2520, 1), (128, 128, 3, 5, 2520, 1), (128, 128, 11, 5, 2520, 1), (64, 64, 11, 1, 5040, 1), (64, 64, 3, 1, 5040, 0),
          (32, 32, 11, 1, 10080, 1), (32, 32, 7, 5, 10080, 1), (33, 16, 11, 1, 130, 1), (40, 32, 7, 3, 300, 1), (100, 48, 5, 2, 1000, 0), (256, 256, 11, 1, 252, 1)]
if len(sys.argv) > 1 and sys.argv[1] == "child":
    sys.path.insert(0, ROOT)
    from obs_rvc_amd import _native
    L = _native.lib()
    L.rvc_debug_conv_check.restype = C.c_double; L.rvc_debug_conv_check.argtypes = [C.c_void_p] + [C.c_int] * 7
    L.rvc_debug_conv_bench.restype = C.c_double; L.rvc_debug_conv_bench.argtypes = [C.c_void_p] + [C.c_int] * 7
    h = C.c_void_p(); assert L.rvc_create(b"/tmp", 0, C.byref(h)) == 0
    for (M, Cin, KW, dil, N, pre) in SHAPES:
        err = L.rvc_debug_conv_check(h, M, Cin, KW, dil, N, 1, pre)
        us = L.rvc_debug_conv_bench(h, M, Cin, KW, dil, N, 100, pre)
        print("  M=%d Cin=%d KW=%d dil=%d N=%d pre=%d: err %.2e  %.1f us  %.1f TF/s" % (M, Cin, KW, dil, N, pre, err, us, 2.0 * M * Cin * KW * N / us / 1e6), flush=True)
else:
    for mode in ("0", "2"):
        print("RVC_CONV_TILE=" + mode, flush=True)
        subprocess.run([sys.executable, __file__, "child"], env=dict(os.environ, RVC_CONV_TILE=mode))
