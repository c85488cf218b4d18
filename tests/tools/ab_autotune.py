#!/usr/bin/env python
"""Same-process A/B of plan-time selection by measurement (rvc_set_plan_autotune): chunk time (GPU events, median) per stream count with the rules only and with
the autotuned plan, two alternating runs each on ONE engine per stream count; what the tuner changed against the rules; what a plan build costs.
usage: ab_autotune.py [streams=8,16,32,64] [version=2] [--dump]"""
import ctypes, os, sys
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from common import BASELINE_160MS as g, voice_signal, zoo
from obs_rvc_amd.rvc import RvcInfer

args = [a for a in sys.argv[1:] if not a.startswith("--")]
streams = [int(v) for v in (args[0] if args else "8,16,32,64").split(",")]
ver = int(args[1]) if len(args) > 1 else 2
z = zoo("full", ver)
L, chunk, N = g.input_buffer_16k_size, g.sample_frame_16k, g.model_return_size
for S in streams:
    x = torch.from_numpy(np.stack([voice_signal(L, seed=1 + s) for s in range(S)])).cuda(); o = torch.empty((S, N), device="cuda")
    e = RvcInfer(z["data"], device=0); e.load_contentvec(ver); e.load_f0(1); e.load_model(z["model"]); e.set_streams(S); e.set_noise_seed(1, 0)
    res = {0: [], 1: []}; info = {}
    for rep in range(2):
        for on in (0, 1):
            e.set_plan_autotune(bool(on))          # (drops the cached plans: the next call builds one)
            for _ in range(6):
                e.infer_device(x.data_ptr(), L, chunk, 12, g.skip_head, g.model_return_length, o.data_ptr(), N, sync=True)
            if on and rep == 0:
                info = e.plan_autotune_info()
            elif on:
                info["second_build"] = e.plan_autotune_info()
            elif rep == 0:
                info_off = e.plan_autotune_info()
            gm = []
            for _ in range(60 if S <= 16 else 30):
                e.infer_device(x.data_ptr(), L, chunk, 12, g.skip_head, g.model_return_length, o.data_ptr(), N, sync=True); gm.append(e.last_gpu_ms())
            res[on].append(float(np.median(gm)))
    print("streams %3d  rules: %s ms (build %.0f ms)   autotuned: %s ms   first tuned build: %s" % (S, " ".join("%.3f" % t for t in res[0]), info_off["build_ms"], " ".join("%.3f" % t for t in res[1]), info), flush=True)
    e.close()
if "--dump" in sys.argv:
    lib = ctypes.CDLL(None)
    from obs_rvc_amd import _native
    Lb = _native.lib()
    buf = ctypes.create_string_buffer(1 << 22)
    Lb.rvc_debug_autotune_dump.argtypes = [ctypes.c_char_p, ctypes.c_size_t]
    n = Lb.rvc_debug_autotune_dump(buf, len(buf))
    print("%d decisions" % n)
    for ln in buf.value.decode().splitlines():
        if "-> [0," not in ln:
            print("CHANGED", ln)
    for ln in buf.value.decode().splitlines():
        if "-> [0," in ln:
            print("kept   ", ln)
