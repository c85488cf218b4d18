#!/usr/bin/env python
"""Host-side launch cost vs GPU completion time of one chunk (graph replay and eager launches)."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from common import BASELINE_160MS as g, voice_signal, zoo
from obs_rvc_amd.rvc import RvcInfer
z = zoo("full")
eng = RvcInfer(z["data"], device=0); eng.load_contentvec(2); eng.load_f0(1); eng.load_model(z["model"]); eng.set_streams(1); eng.set_noise_seed(1, 0)
L, chunk, N = g.input_buffer_16k_size, g.sample_frame_16k, g.model_return_size
x = torch.from_numpy(voice_signal(L, seed=1)[None]).cuda(); out = torch.empty((1, N), device="cuda")
for graph in (True, False):
    eng.set_use_graph(graph)
    for _ in range(5):
        eng.infer_device(x.data_ptr(), L, chunk, 12, g.skip_head, g.model_return_length, out.data_ptr(), N, sync=True)
    ret, tot = [], []
    for _ in range(50):
        t0 = time.perf_counter()
        eng.infer_device(x.data_ptr(), L, chunk, 12, g.skip_head, g.model_return_length, out.data_ptr(), N, sync=False)
        t1 = time.perf_counter()
        eng.synchronize()
        t2 = time.perf_counter()
        ret.append(t1 - t0); tot.append(t2 - t0)
    print("graph=%d  launch-return %.3f ms   total %.3f ms   gpu_ms(last) %.3f" % (graph, np.median(ret) * 1e3, np.median(tot) * 1e3, eng.last_gpu_ms()))
xh = voice_signal(L, seed=1)
for graph in (True, False):
    eng.set_use_graph(graph)
    for name, fn in (("hubert", lambda: eng.hubert(xh)), ("pitch", lambda: eng.pitch(xh, 12, chunk)),
                     ("infer", lambda: eng.infer(xh, chunk, 12, g.skip_head, g.model_return_length))):
        ms = []
        for _ in range(30):
            t0 = time.perf_counter(); fn(); ms.append((time.perf_counter() - t0) * 1e3)
        print("graph=%d %-7s wall ms median %.3f min %.3f" % (graph, name, np.median(ms[5:]), min(ms[5:])))

if os.environ.get("RVC_STAMPS"):
    import ctypes
    lib = eng._L
    lib.rvc_debug_stamps.restype = ctypes.c_int
    lib.rvc_debug_stamps.argtypes = [ctypes.c_void_p, ctypes.c_char_p, ctypes.c_size_t]
    eng.set_use_graph(bool(int(os.environ.get('PROBE_GRAPH', '0'))))
    for _ in range(4):
        eng.infer_device(x.data_ptr(), L, chunk, 12, g.skip_head, g.model_return_length, out.data_ptr(), N, sync=True)
    buf = ctypes.create_string_buffer(1 << 16)
    n = lib.rvc_debug_stamps(eng._h, buf, len(buf))
    print("stamps", n); print(buf.value.decode())
