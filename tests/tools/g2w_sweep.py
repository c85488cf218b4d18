#!/usr/bin/env python
"""Tuning aid: the table-free 1x1 layers of ContentVec at a few streams -- the planner's own choice against igemm2w_kernel (register-direct 32x32x2)
forced with every wave tile / K split (test hook RVC_FORCE_G2W = "tile,ks": works on the product library).  Isolated launches, one process.
usage: g2w_sweep.py [streams ...]"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
from obs_rvc_amd import _native
L = _native.lib()
L.rvc_debug_option.argtypes = [C.c_char_p, C.c_char_p]
if not hasattr(L, "rvc_debug_conv_bench"):
    raise SystemExit("needs the tuning library (rvc_debug_conv_bench): RVC_TUNING=1 RVC_LIB_OVERRIDE=obs_rvc_amd/csrc/librvc_tuning.so")
L.rvc_debug_conv_bench.restype = C.c_double
L.rvc_debug_conv_bench.argtypes = [C.c_void_p] + [C.c_int] * 9
h = C.c_void_p()
assert L.rvc_create(b"/tmp", 0, C.byref(h)) == 0
SHAPES = [("ffn1 3072x768", 3072, 768, 111, 3), ("qkv 2304x768", 2304, 768, 111, 0), ("ffn2 768x3072", 768, 3072, 111, 0), ("out 768x768", 768, 768, 111, 0),
          ("proj 768x512", 768, 512, 111, 0)]
VAR = [("auto", None)] + [("%dx%d k%d" % (32 * (1, 2, 2)[t], 32 * (1, 1, 2)[t], k), "%d,%d" % (t, k)) for t in (0, 1, 2) for k in ((1, 2, 3, 4, 6, 8, 12, 16) if t == 0 else (1, 2, 3, 4, 6, 8))]
for S in [int(a) for a in sys.argv[1:]] or [1, 2, 4, 8, 16, 32]:
    print("streams %d: us per variant (best marked *)" % S)
    print("%-16s" % "layer" + "".join("%-10s" % v[0] for v in VAR))
    for label, M, Cin, N, act in SHAPES:
        res = []
        for name, val in VAR:
            L.rvc_debug_option(b"RVC_FORCE_G2W", val.encode() if val else None)
            res.append(L.rvc_debug_conv_bench(h, M, Cin, 1, 1, N, 10 if S >= 32 else 30, 0, S, act))
        L.rvc_debug_option(b"RVC_FORCE_G2W", None)
        best = min(r for r in res if r > 0)
        print("%-16s" % label + "".join(("%8.1f%s " % (r, "*" if r == best else " ")) for r in res) + " best/auto %.2f (%s)" % (best / res[0], VAR[res.index(best)][0]), flush=True)
