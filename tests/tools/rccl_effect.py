#!/usr/bin/env python
"""Tuning aid: does creating (and destroying) an RCCL communicator in the process change the per-chunk latency afterwards?"""
import os, sys, time
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
    print("%-50s %.4f ms" % (tag, np.median(ts) * 1e3), flush=True)
e = mk(); t(e, "fresh engine, no index")
vecs = W.make_index()
e.load_index(vecs); e.set_index_rate(0.75); t(e, "index uploaded directly")
e2 = mk(); t(e2, "second engine, no index (before any RCCL)")
uid = e2.rccl_unique_id(); e2.index_broadcast(uid, 0, 1, vecs[:1000]); e2.set_index_rate(0.0); t(e2, "second engine after a one-rank RCCL comm, rate 0")
t(e, "first engine (direct index) after the RCCL comm")
e3 = mk(); t(e3, "third engine created after the RCCL comm, no index")
