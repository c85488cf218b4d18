"""Tuning aid: per-step timeline of the persistent synthesizer front end (csrc/synth_front.hip), from workgroup 0's wall-clock stamps.
usage: RVC_FRONT_STAMPS=1 python tests/tools/front_probe.py"""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ.setdefault("RVC_FRONT_STAMPS", "1")
os.environ.setdefault("RVC_SYNTH_FRONT", "1")
from common import BASELINE_160MS as g, voice_signal, zoo  # noqa: E402
from obs_rvc_amd import _native  # noqa: E402
from obs_rvc_amd.rvc import RvcInfer  # noqa: E402

z = zoo("full")
eng = RvcInfer(z["data"]); eng.load_contentvec(2); eng.load_f0(1); eng.load_model(z["model"]); eng.set_noise_seed(1, 0)
x = voice_signal(g.input_buffer_16k_size, seed=3)
acc = None
N = 20
for i in range(N + 5):
    eng.infer(x, g.sample_frame_16k, 12, g.skip_head, g.model_return_length)
    out = (C.c_double * 512)()
    n = _native.lib().rvc_debug_front_stamps(eng._h, out, 512)
    t = np.array(out[:n])
    sub = np.array(out[128:512])
    if i >= 5:
        acc = t if acc is None else acc + t
        sacc = sub if i == 5 else sacc + sub
t = acc / N
names = ["phone"]
for l in range(6):
    names += ["l%d.qkv" % l, "l%d.attn+o" % l, "l%d.ff1" % l, "l%d.ff2" % l]
names += ["proj+prior"]
for f in range(3, -1, -1):
    names += ["f%d.pre" % f] + sum([["f%d.in%d" % (f, j), "f%d.rs%d" % (f, j)] for j in range(3)], []) + ["f%d.post" % f]
d = np.diff(t)
for i, dt in enumerate(d):
    sb = sacc[(i + 1) * 4:(i + 1) * 4 + 4] / N
    print("%-12s start %8.2f us  %6.2f us   staged %5.2f  mfma %5.2f  barrier %5.2f  epilogue %5.2f" % (names[i] if i < len(names) else "?", t[i], dt, sb[0], sb[1], sb[2], sb[3]))
print("total %.2f us over %d steps" % (t[-1], len(d)))
