#!/usr/bin/env python
"""Tuning aid: 100k index delivered by a plain upload vs by rvc_index_broadcast (one-rank communicator): per-chunk latency after."""
import os, sys, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from common import BASELINE_160MS as g, voice_signal, zoo
from obs_rvc_amd import weights as W
from obs_rvc_amd.rvc import RvcInfer
z = zoo("full")
L, chunk, N = g.input_buffer_16k_size, g.sample_frame_16k, g.model_return_size
x = torch.from_numpy(voice_signal(L, seed=1)[None]).cuda(); out = torch.empty((1, N), device="cuda")
def mk():
    e = RvcInfer(z["data"], device=0); e.load_contentvec(2); e.load_f0(1); e.load_model(z["model"]); e.set_streams(1); e.set_noise_seed(1, 0); return e
def t(e, tag):
    for _ in range(10): e.infer_device(x.data_ptr(), L, chunk, 12, g.skip_head, g.model_return_length, out.data_ptr(), N, sync=True)
    ts = []
    for _ in range(60):
        t0 = time.perf_counter(); e.infer_device(x.data_ptr(), L, chunk, 12, g.skip_head, g.model_return_length, out.data_ptr(), N, sync=True); ts.append(time.perf_counter() - t0)
    print("%-60s %.4f ms  p99 %.4f" % (tag, np.median(ts) * 1e3, np.percentile(ts, 99) * 1e3), flush=True)
vecs = W.make_index()
mode = sys.argv[1]
e = mk()
if mode == "direct":
    e.load_index(vecs)
elif mode == "rccl":
    e.index_broadcast(e.rccl_unique_id(), 0, 1, vecs)
elif mode == "rccl_then_direct":
    e.index_broadcast(e.rccl_unique_id(), 0, 1, vecs[:1000]); e.load_index(vecs)
elif mode == "direct_then_rccl_resend":
    e.load_index(vecs); e.index_broadcast(e.rccl_unique_id(), 0, 1, None)
e.set_index_rate(0.75)
t(e, mode)
