#!/usr/bin/env python
"""Tuning aid: section timeline of one chunk (test hook RVC_STAMPS: device timestamps) for a given stream count / model version.
usage: python tests/tools/timeline.py [streams] [version] [HOOK=VALUE ...]"""
import ctypes
import os
import sys

import numpy as np

os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: E402
from common import BASELINE_160MS as g, set_opt, voice_signal, zoo  # noqa: E402
from obs_rvc_amd.rvc import RvcInfer  # noqa: E402

set_opt("RVC_STAMPS", "1")          # test hook (rvc_debug_option): device timestamps at the section boundaries
S = int(sys.argv[1]) if len(sys.argv) > 1 else 1
ver = int(sys.argv[2]) if len(sys.argv) > 2 else 2
for a in sys.argv[3:]:              # further test hooks: NAME=VALUE (set before the models are loaded: RVC_NO_LN_FUSE is read at load)
    set_opt(*a.split("=", 1))
z = zoo("full", ver)
eng = RvcInfer(z["data"], device=0); eng.load_contentvec(ver); eng.load_f0(1); eng.load_model(z["model"]); eng.set_streams(S); eng.set_noise_seed(1, 0)
L, chunk, N = g.input_buffer_16k_size, g.sample_frame_16k, g.model_return_size
x = torch.from_numpy(np.stack([voice_signal(L, seed=1 + s) for s in range(S)])).cuda(); out = torch.empty((S, N), device="cuda")
lib = eng._L
lib.rvc_debug_stamps.restype = ctypes.c_int
lib.rvc_debug_stamps.argtypes = [ctypes.c_void_p, ctypes.c_char_p, ctypes.c_size_t]
acc = {}
order = []
for it in range(12):
    eng.infer_device(x.data_ptr(), L, chunk, 12, g.skip_head, g.model_return_length, out.data_ptr(), N, sync=True)
    buf = ctypes.create_string_buffer(1 << 16)
    lib.rvc_debug_stamps(eng._h, buf, len(buf))
    if it < 4:
        continue
    seen = {}
    for ln in buf.value.decode().splitlines():
        name, t = ln.rsplit(" ", 1)
        k = seen.get(name, 0); seen[name] = k + 1
        key = "%s#%d" % (name, k)
        if key not in acc:
            acc[key] = []; order.append(key)
        acc[key].append(float(t))
print("streams %d, v%d: section boundaries, us since the first stamp (median of 8 chunks), gpu ms %.3f" % (S, ver, eng.last_gpu_ms()))
print("  ".join("%s %.0f" % (k, np.median(acc[k])) for k in order))
