#!/usr/bin/env python
"""Chunk time against the number of XCDs given to the f0 branch (test hook RVC_F0_XCDS), per model version and stream count, on the PRODUCT library
(engine.hip configure_aux_streams: partition sizes are tuned on the product build only).  usage: f0_xcds.py [versions] [streams] [xcds]"""
import os, sys, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from common import BASELINE_160MS as g, set_opt, voice_signal, zoo
from obs_rvc_amd.rvc import RvcInfer

vers = [int(v) for v in (sys.argv[1] if len(sys.argv) > 1 else "1,2").split(",")]
streams = [int(v) for v in (sys.argv[2] if len(sys.argv) > 2 else "1,2,4").split(",")]
xcds = [int(v) for v in (sys.argv[3] if len(sys.argv) > 3 else "1,2").split(",")]
L, chunk, N = g.input_buffer_16k_size, g.sample_frame_16k, g.model_return_size
for ver in vers:
    z = zoo("full", ver)
    for S in streams:
        x = torch.from_numpy(np.stack([voice_signal(L, seed=1 + s) for s in range(S)])).cuda(); o = torch.empty((S, N), device="cuda")
        for rep in range(2):
            for xc in xcds:
                set_opt("RVC_F0_XCDS", str(xc))
                e = RvcInfer(z["data"], device=0); e.load_contentvec(ver); e.load_f0(1); e.load_model(z["model"]); e.set_streams(S); e.set_noise_seed(1, 0)
                for _ in range(15):
                    e.infer_device(x.data_ptr(), L, chunk, 12, g.skip_head, g.model_return_length, o.data_ptr(), N, sync=True)
                gm = []
                for _ in range(100):
                    e.infer_device(x.data_ptr(), L, chunk, 12, g.skip_head, g.model_return_length, o.data_ptr(), N, sync=True); gm.append(e.last_gpu_ms())
                print("v%d streams %d  f0 on %d XCD(s): gpu p50 %.4f ms" % (ver, S, xc, np.median(gm)), flush=True)
                e.close()
set_opt("RVC_F0_XCDS", None)
