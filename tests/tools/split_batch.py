#!/usr/bin/env python
"""Tuning aid: S streams as ONE engine of S streams versus k engines of S/k streams issued back to back (their kernels overlap on the
GPU: one engine's kernel tail is filled by the other's).   usage: split_batch.py [S] [k]"""
import os, sys, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from common import BASELINE_160MS as g, voice_signal, zoo
from obs_rvc_amd.rvc import RvcInfer
S = int(sys.argv[1]) if len(sys.argv) > 1 else 64
k = int(sys.argv[2]) if len(sys.argv) > 2 else 2
z = zoo("full")
L, chunk, N = g.input_buffer_16k_size, g.sample_frame_16k, g.model_return_size
def mk(s):
    e = RvcInfer(z["data"], device=0); e.load_contentvec(2); e.load_f0(1); e.load_model(z["model"]); e.set_streams(s); e.set_noise_seed(1, 0); return e
def run(engs, tag):
    s = S // len(engs)
    xs = [torch.from_numpy(np.stack([voice_signal(L, seed=1 + i * s + j) for j in range(s)])).cuda() for i in range(len(engs))]
    outs = [torch.empty((s, N), device="cuda") for _ in engs]
    def step():
        for e, x, o in zip(engs, xs, outs):
            e.infer_device(x.data_ptr(), L, chunk, 12, g.skip_head, g.model_return_length, o.data_ptr(), N, sync=False)
        for e in engs: e.synchronize()
    for _ in range(3): step()
    ts = []
    for _ in range(12):
        t0 = time.perf_counter(); step(); ts.append(time.perf_counter() - t0)
    ms = np.median(ts) * 1e3
    print("%-40s %.3f ms per step of %d streams  -> %.0f frames/s" % (tag, ms, S, S * 16 / ms * 1e3), flush=True)
run([mk(S)], "1 engine x %d streams" % S)
run([mk(S // k) for _ in range(k)], "%d engines x %d streams" % (k, S // k))
