#!/usr/bin/env python
"""Edge-case probe: each case runs in its own process (a GPU fault kills the process, not the probe)."""
import subprocess, sys, os
CASES = {
 "short_L": "y=e.infer(x[:400],2560,12,0,1)",
 "L_5120": "y=e.infer(x[:5120],2560,12,0,1)",
 "L_not_multiple": "y=e.infer(x[:35001],2560,12,200,17)",
 "big_R": "y=e.infer(x,2560,12,0,223)",
 "R_too_big": "y=e.infer(x,2560,12,200,100)",
 "R_zero": "y=e.infer(x,2560,12,200,0)",
 "frame16k_zero": "y=e.infer(x,0,12,200,21)",
 "frame16k_huge": "y=e.infer(x,10**6,12,200,21)",
 "shift_huge": "y=e.infer(x,2560,1200,200,21)",
 "shift_neg_huge": "y=e.infer(x,2560,-1200,200,21)",
 "tiny_index": "e.load_index(W.make_index(3,48,seed=1)); e.set_index_rate(0.5); y=e.infer(x,2560,12,200,21); print(e.knn()[0][:2])",
 "index_one": "e.load_index(W.make_index(1,48,seed=1)); e.set_index_rate(1.0); y=e.infer(x,2560,12,200,21)",
 "index_dup": "v=np.repeat(W.make_index(1,48,seed=1),5000,0); e.load_index(v); e.set_index_rate(0.7); y=e.infer(x,2560,12,200,21)",
 "streams_0": "e.set_streams(0)",
 "streams_300": "e.set_streams(300); y=e.infer_batch(np.stack([x]*300),2560,12,200,21)",
 "pitch_short": "y=e.pitch(x[:3000],12,2560)",
 "pitch_frame_big": "y=e.pitch(x,12,30000)",
 "hubert_400": "y=e.hubert(x[:400])",
 "hubert_401": "y=e.hubert(x[:401])",
 "long_L": "y=e.infer(np.concatenate([x]*6),2560,12,200,21)",
 "zeros_rate1": "e.load_index(np.zeros((100,48),np.float32)); e.set_index_rate(1.0); y=e.infer(np.zeros_like(x),2560,12,200,21)",
}
PRE = """
import sys, numpy as np
sys.path.insert(0,%r); sys.path.insert(0,%r)
from common import zoo, voice_signal, BASELINE_160MS as g
from obs_rvc_amd.rvc import RvcInfer
from obs_rvc_amd import weights as W
z=zoo("tiny"); e=RvcInfer(z["data"]); e.load_contentvec(2); e.load_f0(); e.load_model(z["model"])
x=voice_signal(g.input_buffer_16k_size, seed=1)
y=None
try:
    %s
    print("OK", None if y is None else (np.asarray(y).shape, bool(np.isfinite(np.asarray(y)).all())))
except Exception as ex:
    print("RAISED", type(ex).__name__, str(ex)[:70])
y2=e.infer(x,2560,12,200,21) if getattr(e,'n_streams',1)==1 else None
print("AFTER", None if y2 is None else bool(np.isfinite(y2).all()))
"""
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for name, code in CASES.items():
    src = PRE % (root, os.path.join(root, "tests"), code)
    r = subprocess.run([sys.executable, "-c", src], capture_output=True, text=True, timeout=120)
    out = [l for l in r.stdout.strip().splitlines() if l.startswith(("OK", "RAISED", "AFTER"))]
    print("%-16s rc=%d %s" % (name, r.returncode, " | ".join(out) if out else r.stderr.strip().splitlines()[-1:][0] if r.stderr.strip() else "?"))
