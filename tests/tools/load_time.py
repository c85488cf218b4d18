"""Tuning aid: model load time (the composed WaveNets are built with the model)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from common import zoo
from obs_rvc_amd.rvc import RvcInfer
z = zoo("full")
eng = RvcInfer(z["data"], device=0); eng.load_contentvec(2); eng.load_f0(1)
for _ in range(3):
    t0 = time.perf_counter(); eng.load_model(z["model"]); print("load_model %.3f s (RVC_NO_WN_COMPOSE=%s)" % (time.perf_counter() - t0, os.environ.get("RVC_NO_WN_COMPOSE", "")))
