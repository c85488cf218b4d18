#!/usr/bin/env python
"""Second part of the late-engine question (tests/tools/late_engine.py): is the first engine of a process slower while ANOTHER engine merely exists,
while it has recently run, or after it is gone?   usage: late_engine2.py"""
import gc, os, sys, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from common import BASELINE_160MS as g, set_opt, voice_signal, zoo
from obs_rvc_amd.rvc import RvcInfer
z = zoo("full")
L, chunk, N = g.input_buffer_16k_size, g.sample_frame_16k, g.model_return_size
def mk(S=1):
    e = RvcInfer(z["data"], device=0); e.load_contentvec(2); e.load_f0(1); e.load_model(z["model"]); e.set_streams(S); e.set_noise_seed(1, 0); return e
x1 = torch.from_numpy(voice_signal(L, seed=1)[None]).cuda(); o1 = torch.empty((1, N), device="cuda")
def measure(e, tag, n=150):
    for _ in range(15): e.infer_device(x1.data_ptr(), L, chunk, 12, g.skip_head, g.model_return_length, o1.data_ptr(), N, sync=True)
    gm = []
    for _ in range(n):
        e.infer_device(x1.data_ptr(), L, chunk, 12, g.skip_head, g.model_return_length, o1.data_ptr(), N, sync=True); gm.append(e.last_gpu_ms())
    print("%-64s gpu ms p50 %.4f  p10 %.4f  p90 %.4f" % (tag, np.median(gm), np.percentile(gm, 10), np.percentile(gm, 90)), flush=True)
keeper = mk()
measure(keeper, "keeper alone")
measure(keeper, "keeper alone, again")
e = mk()
measure(keeper, "keeper, a second engine exists (never ran)")
measure(e, "second engine")
measure(keeper, "keeper, right after the second engine ran")
time.sleep(3)
measure(keeper, "keeper, second engine idle for 3 s")
e.close(); del e; gc.collect()
measure(keeper, "keeper, second engine destroyed")
h = mk(16); x16 = torch.from_numpy(np.stack([voice_signal(L, seed=1 + s) for s in range(16)])).cuda(); o16 = torch.empty((16, N), device="cuda")
measure(keeper, "keeper, a 16-stream engine exists (never ran)")
for _ in range(60): h.infer_device(x16.data_ptr(), L, chunk, 12, g.skip_head, g.model_return_length, o16.data_ptr(), N, sync=True)
measure(keeper, "keeper, right after the 16-stream engine ran 60 steps")
h.close(); del h, x16, o16; gc.collect(); torch.cuda.empty_cache()
measure(keeper, "keeper, 16-stream engine destroyed")
time.sleep(3)
measure(keeper, "keeper, 3 s later")
keeper.close()
