#!/usr/bin/env python
"""Soak of the one-launch retrieval's hand-over protocol (granules + ticket + selectors that wait for the last arrival): N chunks with the
100 k x 768 index at 1 stream and N/4 at 3 streams, the same inputs cycling; every chunk's hits must equal the first pass's (a lost or stale
granule, a selector that started early or a counter that was not re-armed would show as different hits, a time-out as status 7).
usage: knn_soak.py [chunks=20000]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from common import BASELINE_160MS as g, voice_signal, zoo
from obs_rvc_amd import weights as W
from obs_rvc_amd.rvc import RvcInfer
n = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
z = zoo("full")
index = W.make_index(100000, 768, seed=7)
L, chunk, N = g.input_buffer_16k_size, g.sample_frame_16k, g.model_return_size
for S, count in ((1, n), (3, n // 4)):
    eng = RvcInfer(z["data"]); eng.load_contentvec(2); eng.load_f0(1); eng.load_model(z["model"]); eng.set_streams(S); eng.set_noise_seed(5, 0)
    eng.load_index(index); eng.set_index_rate(0.75)
    xs = [torch.from_numpy(np.stack([voice_signal(L, seed=100 * k + s) for s in range(S)])).cuda() for k in range(8)]
    out = torch.empty((S, N), device="cuda")
    ref = {}
    bad = 0
    t0 = time.perf_counter()
    for i in range(count):
        k = i % 8
        eng.infer_device(xs[k].data_ptr(), L, chunk, 12, g.skip_head, g.model_return_length, out.data_ptr(), N, sync=True)     # raises on status 7
        if i < 8 or i % 16 == 0:
            idx, dist = eng.knn(rows_cap=S * 64)
            if k not in ref:
                ref[k] = (idx.copy(), dist.copy())
            elif not (np.array_equal(ref[k][0], idx) and np.array_equal(ref[k][1], dist)):
                bad += 1
    dt = time.perf_counter() - t0
    print("%d stream(s): %d chunks in %.1f s (%.3f ms per chunk), hits checked on every 16th: %d mismatches" % (S, count, dt, dt / count * 1e3, bad), flush=True)
    eng.close()
    assert bad == 0
print("ok")
