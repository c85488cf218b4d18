#!/usr/bin/env python
"""Plan-cache churn: many geometries in rotation (the cache keeps 6 plans); device memory must stay bounded."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from common import voice_signal, zoo
from obs_rvc_amd.rvc import RvcInfer
z = zoo("full")
e = RvcInfer(z["data"]); e.load_contentvec(2); e.load_f0(1); e.load_model(z["model"])
x = voice_signal(35840, seed=1)
free = []
for it in range(60):
    R = 1 + (it * 7) % 21
    y = e.infer(x, 2560, 12, 200, R)
    assert np.isfinite(y).all()
    if it % 10 == 9:
        torch.cuda.synchronize(); free.append(torch.cuda.mem_get_info()[0] / 1e6)
print("free MB every 10 geometries:", [round(f) for f in free], "spread", round(max(free) - min(free)))
