"""Tuning aid: time of the retrieval kernels (per-launch HIP events of the scan; wall clock of index-on vs index-off chunks)."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
import torch
from common import BASELINE_160MS as g, voice_signal, zoo
from obs_rvc_amd import weights as W
from obs_rvc_amd.rvc import RvcInfer
z = zoo("full")
eng = RvcInfer(z["data"], device=0); eng.load_contentvec(2); eng.load_f0(1); eng.load_model(z["model"]); eng.set_noise_seed(1, 0)
eng.load_index(W.make_index()); eng.set_index_rate(0.75)
L, chunk, N = g.input_buffer_16k_size, g.sample_frame_16k, g.model_return_size
x = torch.from_numpy(voice_signal(L, seed=1)[None]).cuda(); out = torch.empty((1, N), device="cuda")
def step(): eng.infer_device(x.data_ptr(), L, chunk, 12, g.skip_head, g.model_return_length, out.data_ptr(), N, sync=True)
for _ in range(20): step()
t0 = time.perf_counter()
for _ in range(100): step()
wall = (time.perf_counter() - t0) / 100 * 1e3
eng.set_profile(True)
ms = []
for _ in range(10):
    step(); kn, kms, kby = eng.profile_last_knn(); ms.append(kms / max(kn, 1))
eng.set_profile(False)
print("RVC_KNN_WGS=%s  chunk %.4f ms  scan %.2f us = %.2f TB/s" % (os.environ.get("RVC_KNN_WGS", "768"), wall, np.median(ms) * 1e3, 307.2e6 / (np.median(ms) * 1e-3) / 1e12))
