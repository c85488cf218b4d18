#!/usr/bin/env python
"""Per-call time of the HIP polyphase resampler for the plugin's converter shapes (device-resident, HIP events via torch)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from common import zoo
from obs_rvc_amd.resample import FftFixedInOut
from obs_rvc_amd.rvc import RvcInfer
eng = RvcInfer(zoo("tiny")["data"])
for ri, ro, ch in [(48000, 16000, 8640), (48000, 48000, 10080), (40000, 48000, 14000), (48000, 44100, 10080), (44100, 16000, 7938)]:
    t0 = time.perf_counter(); r = FftFixedInOut(eng, ri, ro, ch); tc = time.perf_counter() - t0
    fi, fo = r.input_frames_next(), r.output_frames_max()
    x = torch.randn(fi, device="cuda"); y = torch.empty(fo, device="cuda")
    for _ in range(5): r.process_device(x.data_ptr(), y.data_ptr(), sync=True)
    t0 = time.perf_counter()
    for _ in range(200): r.process_device(x.data_ptr(), y.data_ptr(), sync=False)
    eng.synchronize(); dt = (time.perf_counter() - t0) / 200
    print("%d->%d chunk %d: fft_in %d fft_out %d  create %.1f ms  process %.1f us  (%.1f GMAC/s)" % (ri, ro, ch, fi, fo, tc * 1e3, dt * 1e6, fi * 2 * fo / dt / 1e9))
