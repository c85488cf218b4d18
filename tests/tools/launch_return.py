import os, sys, time
import numpy as np
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
import torch
from common import BASELINE_160MS as g, voice_signal, zoo
from obs_rvc_amd.rvc import RvcInfer
S = int(sys.argv[1]); ver = int(sys.argv[2])
z = zoo("full", ver)
eng = RvcInfer(z["data"], device=0); eng.load_contentvec(ver); eng.load_f0(1); eng.load_model(z["model"]); eng.set_streams(S); eng.set_noise_seed(1, 0)
L, chunk, N = g.input_buffer_16k_size, g.sample_frame_16k, g.model_return_size
x = torch.from_numpy(np.stack([voice_signal(L, seed=1 + s) for s in range(S)])).cuda(); out = torch.empty((S, N), device="cuda")
for _ in range(10):
    eng.infer_device(x.data_ptr(), L, chunk, 12, g.skip_head, g.model_return_length, out.data_ptr(), N, sync=True)
ret, tot, gpu = [], [], []
for _ in range(50):
    t0 = time.perf_counter()
    eng.infer_device(x.data_ptr(), L, chunk, 12, g.skip_head, g.model_return_length, out.data_ptr(), N, sync=False)
    t1 = time.perf_counter()
    eng.synchronize()
    t2 = time.perf_counter()
    ret.append(t1 - t0); tot.append(t2 - t0); gpu.append(eng.last_gpu_ms())
import ctypes
n_ops, pers = ctypes.c_int(0), ctypes.c_int(0)
eng._L.rvc_debug_last_plan(eng._h, ctypes.byref(n_ops), ctypes.byref(pers))
print("S=%d v%d: ops %d  launch-return %.3f ms  total %.3f ms  gpu %.3f ms" % (S, ver, n_ops.value, np.median(ret) * 1e3, np.median(tot) * 1e3, np.median(gpu)))
