#!/usr/bin/env python
"""Measurement aid (round 6): per-step wall time of the first 40 one-stream chunks behind an idle GPU (2 s of host-only work, as between the bench's model load and
its warm-up), and the same behind 80 ms of matrix-core load (rvc_calibrate).  usage: ramp_probe.py"""
import os, sys, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from common import BASELINE_160MS as g, voice_signal, zoo
from obs_rvc_amd import _native
from obs_rvc_amd.rvc import RvcInfer
z = zoo("full")
e = RvcInfer(z["data"], device=0); e.load_contentvec(2); e.load_f0(1); e.load_model(z["model"]); e.set_noise_seed(1, 0)
L, chunk, N = g.input_buffer_16k_size, g.sample_frame_16k, g.model_return_size
x = torch.from_numpy(voice_signal(L, seed=1)[None]).cuda(); o = torch.empty((1, N), device="cuda")
def run(n):
    ts = []
    for _ in range(n):
        t0 = time.perf_counter(); e.infer_device(x.data_ptr(), L, chunk, 12, g.skip_head, g.model_return_length, o.data_ptr(), N, sync=True); ts.append((time.perf_counter() - t0) * 1e3)
    return ts
run(3)                      # plan build
for label, pre in (("idle 2 s", None), ("idle 2 s, then rvc_calibrate", "calib"), ("idle 2 s", None), ("idle 2 s, then rvc_calibrate", "calib")):
    time.sleep(2.0)
    if pre:
        _native.calibrate(0)
    ts = run(40)
    print("%-30s steps 1-5 %s | 6-25 mean %.3f | 26-40 mean %.3f" % (label, " ".join("%.2f" % t for t in ts[:5]), np.mean(ts[5:25]), np.mean(ts[25:])), flush=True)
