#!/usr/bin/env python
"""Tuning aid: latency of a second engine (with index) after a first engine ran in the same process, by what happened before."""
import os, sys, time, gc
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
def t(e, tag, n=60):
    for _ in range(10): e.infer_device(x.data_ptr(), L, chunk, 12, g.skip_head, g.model_return_length, out.data_ptr(), N, sync=True)
    ts = []
    for _ in range(n):
        t0 = time.perf_counter(); e.infer_device(x.data_ptr(), L, chunk, 12, g.skip_head, g.model_return_length, out.data_ptr(), N, sync=True); ts.append(time.perf_counter() - t0)
    print("%-70s %.4f ms  p99 %.4f" % (tag, np.median(ts) * 1e3, np.percentile(ts, 99) * 1e3), flush=True)
vecs = W.make_index()
mode = sys.argv[1]
a = mk(); t(a, "A plain")
if "host" in mode:
    xh = voice_signal(L, seed=1)
    for _ in range(50): a.infer(xh, chunk, 12, g.skip_head, g.model_return_length)
if "pipe" in mode:
    outs = torch.empty((8, 1, N), device="cuda"); a.set_pipeline(True)
    for i in range(50): a.infer_device(x.data_ptr(), L, chunk, 12, g.skip_head, g.model_return_length, outs[i % 8].data_ptr(), N, sync=False)
    a.synchronize(); a.set_pipeline(False)
if "chain" in mode:
    from obs_rvc_amd.streaming import NativeStreamingSession
    ses = NativeStreamingSession(a, 48000, 0.16, 0.07, 2.0, 48000, 12, 0.75)
    F = ses.sample_frame_size
    for i in range(20):
        try: ses.process_one_frame(np.zeros(F, np.float32) + 0.01 * np.sin(np.arange(F) * 0.05).astype(np.float32))
        except Exception as ex: pass
    del ses
if "prof" in mode:
    a.set_profile(True)
    for _ in range(5): a.infer_device(x.data_ptr(), L, chunk, 12, g.skip_head, g.model_return_length, out.data_ptr(), N, sync=True)
    a.set_profile(False)
if "del" in mode:
    a.close(); del a; gc.collect(); torch.cuda.empty_cache()
b = mk()
if "rccl" in mode:
    b.index_broadcast(b.rccl_unique_id(), 0, 1, vecs)
else:
    b.load_index(vecs)
b.set_index_rate(0.75); t(b, "B with index after [%s]" % mode)
if "rings" in mode:
    from common import chunk_stream
    audio = voice_signal(chunk * 22, seed=0)
    rs = np.stack(list(chunk_stream(audio, L, chunk))[-8:])
    d = torch.from_numpy(rs[:, None, :]).cuda()
    ts = []
    for i in range(70):
        t0 = time.perf_counter(); b.infer_device(d[i % 8].data_ptr(), L, chunk, 12, g.skip_head, g.model_return_length, out.data_ptr(), N, sync=True); ts.append(time.perf_counter() - t0)
    print("   rotating rings: %.4f ms p99 %.4f" % (np.median(ts[10:]) * 1e3, np.percentile(ts[10:], 99) * 1e3))
