#!/usr/bin/env python
"""Many-stream soak of the final build: K steps of S streams through infer_device with rotating inputs; reports latency tail, device memory drift, output
sanity, and RUN-TO-RUN DETERMINISM (state reset, same inputs again: the outputs must be bit-identical -- every reduction in the engine has a fixed order, so a
difference means a race).  usage: soak_streams.py [S=24] [K=1200]"""
import os, sys, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from common import BASELINE_160MS as g, voice_signal, zoo
from obs_rvc_amd.rvc import RvcInfer
S = int(sys.argv[1]) if len(sys.argv) > 1 else 24
K = int(sys.argv[2]) if len(sys.argv) > 2 else 1200
z = zoo("full")
L, chunk, N = g.input_buffer_16k_size, g.sample_frame_16k, g.model_return_size
eng = RvcInfer(z["data"], device=0); eng.load_contentvec(2); eng.load_f0(1); eng.load_model(z["model"]); eng.set_streams(S); eng.set_noise_seed(3, 0)
rings = [torch.from_numpy(np.stack([voice_signal(L, seed=1 + 7 * r + s) for s in range(S)])).cuda() for r in range(4)]
out = torch.empty((S, N), device="cuda")
def run(k, keep):
    lat, outs, bad = [], [], 0
    for i in range(k):
        t0 = time.perf_counter()
        eng.infer_device(rings[i % 4].data_ptr(), L, chunk, 12, g.skip_head, g.model_return_length, out.data_ptr(), N, sync=True)
        lat.append(time.perf_counter() - t0)
        if i == 100: run.free_mid = torch.cuda.mem_get_info()[0]
        if i < keep:
            outs.append(out.clone())
        elif i % 50 == 0:
            o = out
            if not bool(torch.isfinite(o).all()) or float(o.abs().max()) > 1.5: bad += 1
    return np.array(lat) * 1e3, outs, bad
lat0, a, _ = run(12, 12)
eng.reset_state(); eng.set_noise_seed(3, 0)
lat1, b, _ = run(12, 12)
same = all(bool(torch.equal(x, y)) for x, y in zip(a, b))
lat, _, bad = run(K, 0)
free1 = torch.cuda.mem_get_info()[0]
free0 = run.free_mid          # (step 100 of the long run: every plan, arena and torch buffer exists)
lat = lat[10:]
print("streams %d steps %d: p50 %.3f p99 %.3f p99.9 %.3f max %.3f ms; bad outputs %d; device memory drift (step 100 -> end) %.1f MB; 12 steps repeated after reset_state bit-identical: %s"
      % (S, K, np.percentile(lat, 50), np.percentile(lat, 99), np.percentile(lat, 99.9), lat.max(), bad, (free0 - free1) / 1e6, same))
