#!/usr/bin/env python
"""Soak run: N chunks of real-time-shaped streaming through the native session and through infer_device; reports latency tail,
device memory drift and output sanity.  usage: soak.py [chunks=3000]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from common import voice_signal, zoo
from obs_rvc_amd.rvc import RvcInfer
from obs_rvc_amd.streaming import NativeStreamingSession
n = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
z = zoo("full")
eng = RvcInfer(z["data"]); eng.load_contentvec(2); eng.load_f0(1); eng.load_model(z["model"]); eng.set_noise_seed(5, 0)
ses = NativeStreamingSession(eng, 48000, 0.16, 0.07, 2.0, 48000, 12, 0.75)
F = ses.sample_frame_size
base = np.interp(np.arange(F * 64) / 48000.0, np.arange(2560 * 64) / 16000.0, voice_signal(2560 * 64, seed=3)).astype(np.float32)
lat, panics, bad = [], 0, 0
free0 = None
for i in range(n):
    if i == 50:
        free0 = torch.cuda.mem_get_info()[0]      # after the plan (activation arena) exists
    ch = base[(i % 64) * F:(i % 64 + 1) * F]
    t0 = time.perf_counter()
    try:
        y = ses.process_one_frame(ch)
        if not np.isfinite(y).all() or np.abs(y).max() > 1.5:
            bad += 1
    except Exception as ex:
        if "Panic" in str(ex):
            panics += 1
        else:
            raise
    lat.append(time.perf_counter() - t0)
lat = np.array(lat[20:]) * 1e3
free1 = torch.cuda.mem_get_info()[0]
print("chunks %d  p50 %.3f  p99 %.3f  p99.9 %.3f  max %.3f ms  panics(ref quirk) %d  bad outputs %d  device memory drift %.1f MB" %
      (n, np.percentile(lat, 50), np.percentile(lat, 99), np.percentile(lat, 99.9), lat.max(), panics, bad, (free0 - free1) / 1e6))
