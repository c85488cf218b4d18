#!/usr/bin/env python
"""Tuning aid: where the time of ONE implicit-GEMM launch goes.  Builds a probe variant of the library (-DRVC_KPROBE: lane 0 of every
wave stamps the device wall clock at phase boundaries) next to the product library and prints, per phase, when the first / median /
last wave got there, relative to the earliest wave entry.

usage: kprobe.py M Cin KW dil N [RVC_FORCE_CFG=cfg,ks in the environment]
phases: 0 entry | 8 koff copy queued | 9 epilogue operands requested | 10 addresses ready | 11 first weight loads issued | 1 koff slice published (first weight loads in flight) | 2 first activation gathers issued | 3 main loop done |
        4 partials written to LDS | 5 barrier passed | 6 epilogue done
"""
import ctypes as C, os, subprocess, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
CSRC = os.path.join(ROOT, "obs_rvc_amd", "csrc")
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import build_tuning
SO = build_tuning.build(os.environ.get("KPROBE_FLAGS", "").split())
if len(sys.argv) < 6:
    raise SystemExit("built " + SO)
M, Cin, KW, dil, N = [int(v) for v in sys.argv[1:6]]
L = C.CDLL(SO)
h = C.c_void_p()
assert L.rvc_create(b"/tmp", 0, C.byref(h)) == 0
cap = 1 << 18
buf = np.zeros((cap, 16), np.uint64)
ev = C.c_double(); wpw = C.c_int()
L.rvc_debug_conv_probe.restype = C.c_int
L.rvc_debug_conv_probe.argtypes = [C.c_void_p] + [C.c_int] * 5 + [C.c_void_p, C.c_size_t, C.POINTER(C.c_double), C.POINTER(C.c_int)]
nw = L.rvc_debug_conv_probe(h, M, Cin, KW, dil, N, buf.ctypes.data, cap, C.byref(ev), C.byref(wpw))
assert nw > 0, nw
t = buf[:nw].astype(np.int64)
live = t[:, 3] > 0
t0 = t[:, 0][t[:, 0] > 0].min()
print("M=%d Cin=%d KW=%d N=%d  cfg=%s: %d waves (%d per workgroup, %d ran the main loop), dispatch begin..end %.2f us" %
      (M, Cin, KW, N, os.environ.get("RVC_FORCE_CFG", "auto"), nw, wpw.value, int(live.sum()), ev.value))
print("phase   first    median   last   (us after the earliest wave entry);   median duration since the previous phase")
prev = None
for ph in (0, 8, 9, 10, 11, 1, 2, 3, 4, 5, 6):
    v = t[:, ph]; ok = v > 0
    if not ok.any():
        continue
    r = (v[ok] - t0) / 100.0
    dur = ""
    if prev is not None:
        both = ok & (t[:, prev] > 0)
        dur = "%.2f" % float(np.median((t[both, ph] - t[both, prev]) / 100.0))
    print("  %d   %7.2f  %7.2f  %7.2f    %s" % (ph, r.min(), np.median(r), r.max(), dur))
    prev = ph
# share of the summed wave lifetime spent before each stamp (what a wave is doing while it holds its slot)
done = t[:, 6] > 0
if done.any():
    life = (t[done, 6] - t[done, 0]).sum()
    prev = 0
    parts = []
    for ph in (8, 9, 10, 11, 1, 2, 3, 4, 5, 6):
        ok = done & (t[:, ph] > 0) & (t[:, prev] > 0)
        if not (t[done, ph] > 0).all():
            continue
        parts.append("->%d %.1f%%" % (ph, 100.0 * (t[ok, ph] - t[ok, prev]).sum() / life))
        prev = ph
    print("share of the waves' lifetime: " + "  ".join(parts) + "   (mean lifetime %.2f us)" % (life / done.sum() / 100.0))
