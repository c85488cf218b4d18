#!/usr/bin/env python
"""conv32s_kernel forced onto single convolutions (every tile x shapes x streams) against the fp64 host evaluation (rvc_debug_conv_check)."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from common import set_opt
from obs_rvc_amd import _native
L = _native.lib()
L.rvc_debug_conv_check.restype = C.c_double
L.rvc_debug_conv_check.argtypes = [C.c_void_p] + [C.c_int] * 7
h = C.c_void_p()
assert L.rvc_create(b"/tmp", 0, C.byref(h)) == 0
bad = 0
set_opt("RVC_CONV32S", "2")
for tile in range(3):
    set_opt("RVC_CONV32S_TILE", str(tile))
    for streams in (1, 3, 8):
        for (M, Cin, KW, dil, N, pre) in [(32, 32, 11, 1, 1000, 1), (64, 64, 7, 3, 520, 0), (128, 128, 11, 5, 700, 1), (40, 32, 7, 3, 300, 1), (256, 64, 3, 1, 97, 1),
                                          (100, 96, 5, 2, 333, 0), (32, 32, 1, 1, 256, 0), (256, 256, 3, 1, 252, 1)]:
            e = L.rvc_debug_conv_check(h, M, Cin, KW, dil, N, streams, pre)
            ok = 0 <= e < 2e-5
            bad += not ok
            print("tile %d streams %d M=%d Cin=%d KW=%d dil=%d N=%d pre=%d: %.3e %s" % (tile, streams, M, Cin, KW, dil, N, pre, e, "" if ok else "FAIL"), flush=True)
print("FAILURES:", bad)
