#!/usr/bin/env python
"""Same-process A/B of one test hook on the PRODUCT library: chunk time (GPU events, median of 60) per stream count with the hook at value A and at value B
("-" = unset), two alternating runs each.  usage: ab_opt.py NAME A B [streams=8,16,32,64] [version=2]"""
import os, sys
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from common import BASELINE_160MS as g, set_opt, voice_signal, zoo
from obs_rvc_amd.rvc import RvcInfer

name, va, vb = sys.argv[1], sys.argv[2], sys.argv[3]
streams = [int(v) for v in (sys.argv[4] if len(sys.argv) > 4 else "8,16,32,64").split(",")]
ver = int(sys.argv[5]) if len(sys.argv) > 5 else 2
z = zoo("full", ver)
L, chunk, N = g.input_buffer_16k_size, g.sample_frame_16k, g.model_return_size
for S in streams:
    x = torch.from_numpy(np.stack([voice_signal(L, seed=1 + s) for s in range(S)])).cuda(); o = torch.empty((S, N), device="cuda")
    e = RvcInfer(z["data"], device=0); e.load_contentvec(ver); e.load_f0(1); e.load_model(z["model"]); e.set_streams(S); e.set_noise_seed(1, 0)
    res = {va: [], vb: []}
    for rep in range(2):
        for v in (va, vb):
            set_opt(name, None if v == "-" else v)         # (plans built under another hook generation are dropped)
            for _ in range(6):
                e.infer_device(x.data_ptr(), L, chunk, 12, g.skip_head, g.model_return_length, o.data_ptr(), N, sync=True)
            gm = []
            for _ in range(60 if S <= 16 else 30):
                e.infer_device(x.data_ptr(), L, chunk, 12, g.skip_head, g.model_return_length, o.data_ptr(), N, sync=True); gm.append(e.last_gpu_ms())
            res[v].append(float(np.median(gm)))
    print("streams %3d  %s=%s: %s ms   %s=%s: %s ms" % (S, name, va, " ".join("%.3f" % t for t in res[va]), name, vb, " ".join("%.3f" % t for t in res[vb])), flush=True)
    e.close()
set_opt(name, None)
