#!/usr/bin/env python
"""Measurement aid (round 6): wall time per synchronised one-stream chunk against its GPU time, under the runtime's wait settings given in the environment
(ROC_ACTIVE_WAIT_TIMEOUT, ...).  usage: [ENV=...] sync_probe.py"""
import os, sys, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from common import BASELINE_160MS as g, voice_signal, zoo
from obs_rvc_amd.rvc import RvcInfer
z = zoo("full")
e = RvcInfer(z["data"], device=0); e.load_contentvec(2); e.load_f0(1); e.load_model(z["model"]); e.set_noise_seed(1, 0)
L, chunk, N = g.input_buffer_16k_size, g.sample_frame_16k, g.model_return_size
x = torch.from_numpy(voice_signal(L, seed=1)[None]).cuda(); o = torch.empty((1, N), device="cuda")
import gc; gc.collect(); gc.freeze(); gc.disable()
for _ in range(20):
    e.infer_device(x.data_ptr(), L, chunk, 12, g.skip_head, g.model_return_length, o.data_ptr(), N, sync=True)
w, gm = [], []
for _ in range(300):
    t0 = time.perf_counter(); e.infer_device(x.data_ptr(), L, chunk, 12, g.skip_head, g.model_return_length, o.data_ptr(), N, sync=True); w.append((time.perf_counter() - t0) * 1e3); gm.append(e.last_gpu_ms())
print("%-40s wall p50 %.4f mean %.4f  gpu p50 %.4f  wall - gpu %.1f us" % (os.environ.get("PROBE_LABEL", "default"), np.median(w), np.mean(w), np.median(gm), (np.median(w) - np.median(gm)) * 1e3), flush=True)
