#!/usr/bin/env python
"""Does the chunk time drift over the first few hundred chunks of an engine's life (clock ramp, caches, allocator)?  Wall time of synchronised chunks right after
engine creation, in blocks of 10; GPU event time next to it.  usage: warm_drift.py [streams] [blocks]"""
import os, sys, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from common import BASELINE_160MS as g, voice_signal, zoo
from obs_rvc_amd.rvc import RvcInfer
S = int(sys.argv[1]) if len(sys.argv) > 1 else 1
nb = int(sys.argv[2]) if len(sys.argv) > 2 else 40
z = zoo("full", 2)
L, chunk, N = g.input_buffer_16k_size, g.sample_frame_16k, g.model_return_size
x = torch.from_numpy(np.stack([voice_signal(L, seed=1 + s) for s in range(S)])).cuda(); o = torch.empty((S, N), device="cuda")
torch.cuda.synchronize()
for rep in range(2):
    e = RvcInfer(z["data"], device=0); e.load_contentvec(2); e.load_f0(1); e.load_model(z["model"]); e.set_streams(S); e.set_noise_seed(1, 0)
    rows = []
    for b in range(nb):
        w, gm = [], []
        for _ in range(10):
            t = time.perf_counter()
            e.infer_device(x.data_ptr(), L, chunk, 12, g.skip_head, g.model_return_length, o.data_ptr(), N, sync=True)
            w.append(time.perf_counter() - t); gm.append(e.last_gpu_ms())
        rows.append((np.mean(w) * 1e3, np.median(w) * 1e3, np.max(w) * 1e3, np.median(gm)))
    print("engine %d: blocks of 10 chunks: wall mean / median / max, gpu median (ms)" % rep)
    for b, r in enumerate(rows):
        print("  chunks %4d-%4d  %.4f  %.4f  %.4f   gpu %.4f" % (b * 10, b * 10 + 9, *r), flush=True)
    e.close()
