#!/usr/bin/env python
"""Tuning aid: per-launch timing of the implicit-GEMM class over one chunk (HIP events of the dispatches themselves).

usage: op_profile.py [streams] [preset]     -> one line per launch: us, GFLOP, TF/s, % of the fp32 MFMA peak, description
"""
import ctypes, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from common import BASELINE_160MS as g, voice_signal, zoo
from obs_rvc_amd.rvc import RvcInfer
S = int(sys.argv[1]) if len(sys.argv) > 1 else 1
from common import set_opt
for a in sys.argv[3:]:          # test hooks: NAME=VALUE
    set_opt(*a.split("=", 1))
z = zoo(sys.argv[2] if len(sys.argv) > 2 else "full")
eng = RvcInfer(z["data"], device=0); eng.load_contentvec(2); eng.load_f0(1); eng.load_model(z["model"]); eng.set_streams(S); eng.set_noise_seed(1, 0)
L, chunk, N = g.input_buffer_16k_size, g.sample_frame_16k, g.model_return_size
x = torch.from_numpy(np.stack([voice_signal(L, seed=1 + s) for s in range(S)])).cuda(); out = torch.empty((S, N), device="cuda")
for _ in range(3):
    eng.infer_device(x.data_ptr(), L, chunk, 12, g.skip_head, g.model_return_length, out.data_ptr(), N, sync=True)
eng.set_profile(True)
for _ in range(2):
    eng.infer_device(x.data_ptr(), L, chunk, 12, g.skip_head, g.model_return_length, out.data_ptr(), N, sync=True)
lib = eng._L
lib.rvc_debug_profile_dump.restype = ctypes.c_int
lib.rvc_debug_profile_dump.argtypes = [ctypes.c_void_p, ctypes.c_char_p, ctypes.c_size_t]
buf = ctypes.create_string_buffer(1 << 20)
n = lib.rvc_debug_profile_dump(eng._h, buf, len(buf))
tot_us = tot_gf = 0.0
for ln in buf.value.decode().splitlines():
    us, gf, desc = ln.split(" ", 2)
    us, gf = float(us), float(gf)
    tot_us += us; tot_gf += gf
    tf = gf / us * 1e3 if us > 0 else 0.0
    print("%8.2f us %8.4f GF %6.1f TF %5.1f%%  %s" % (us, gf, tf, tf / 157.3 * 100, desc))
import time
eng.set_profile(False)
ts = []
for _ in range(40):
    t0 = time.perf_counter()
    eng.infer_device(x.data_ptr(), L, chunk, 12, g.skip_head, g.model_return_length, out.data_ptr(), N, sync=True)
    ts.append(time.perf_counter() - t0)
print("wall ms/chunk median %.4f min %.4f" % (np.median(ts[5:]) * 1e3, min(ts) * 1e3))
print("total %d launches, %.1f us, %.2f GFLOP, %.1f TF/s" % (n, tot_us, tot_gf, tot_gf / tot_us * 1e3))
