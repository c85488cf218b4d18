#!/usr/bin/env python
"""Tuning aid (tuning library): isolated timings of the ContentVec transformer / stem / decoder GEMM shapes at a given stream count under EVERY tile the
engine has -- the 32x32x2 LDS-staged tiles (RVC_G32W = 3 4 5 7 8) and the register-direct 16x16x4 tiles with in-workgroup K split
(RVC_G32W = -1 + RVC_FORCE_CFG = cfg,ks) -- next to the planner's own choice.  One process, one box.
usage: tile_sweep.py [streams ...]"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
from obs_rvc_amd import _native
L = _native.lib()
L.rvc_debug_conv_bench.restype = C.c_double
L.rvc_debug_conv_bench.argtypes = [C.c_void_p] + [C.c_int] * 9
L.rvc_debug_option.argtypes = [C.c_char_p, C.c_char_p]
h = C.c_void_p()
assert L.rvc_create(b"/tmp", 0, C.byref(h)) == 0
SHAPES = [("ffn1 3072x768", 3072, 768, 1, 1, 111, 3), ("qkv 2304x768", 2304, 768, 1, 1, 111, 0), ("ffn2 768x3072", 768, 3072, 1, 1, 111, 0),
          ("out 768x768", 768, 768, 1, 1, 111, 0), ("conv1 512k3 N=3583", 512, 512, 3, 1, 3583, 3), ("conv2 512k3 N=1791", 512, 512, 3, 1, 1791, 3), ("conv3 512k3 N=895", 512, 512, 3, 1, 895, 3), ("conv4 512k3 N=447", 512, 512, 3, 1, 447, 3),
          ("dec 256k11 N=210", 256, 256, 11, 1, 210, 0), ("dec 128k11 N=2520", 128, 128, 11, 1, 2520, 0)]
G32 = [("128x128", "3"), ("64x256", "4"), ("32x256", "5"), ("128x64", "7"), ("64x64", "8")]
REG = [("r32x64k%d" % k, "4,%d" % k) for k in (1, 4, 8)] + [("r32x32k%d" % k, "3,%d" % k) for k in (1, 4, 8)] + [("r16x64k4", "2,4"), ("r16x16k8", "0,8")]
for S in [int(a) for a in sys.argv[1:]] or [8, 16, 32]:
    print("streams %d: us per variant (best marked *)" % S)
    names = ["auto"] + [g[0] for g in G32] + [r[0] for r in REG]
    print("%-20s" % "layer" + "".join("%-10s" % n for n in names))
    for label, M, Cin, KW, dil, N, act in SHAPES:
        res = []
        for kind, val in [("auto", None)] + [("g", g[1]) for g in G32] + [("r", r[1]) for r in REG]:
            os.environ.pop("RVC_G32W", None); L.rvc_debug_option(b"RVC_FORCE_CFG", None)
            if kind == "g":
                os.environ["RVC_G32W"] = val
            elif kind == "r":
                os.environ["RVC_G32W"] = "-1"; L.rvc_debug_option(b"RVC_FORCE_CFG", val.encode())
            res.append(L.rvc_debug_conv_bench(h, M, Cin, KW, dil, N, 10 if S >= 32 else 25, 0, S, act))
        best = min(r for r in res if r > 0)
        print("%-20s" % label + "".join(("%8.1f%s " % (r, "*" if r == best else " ")) for r in res) + "  best/auto %.2f" % (best / res[0]), flush=True)
