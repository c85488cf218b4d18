import os, sys, time
import numpy as np
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
from common import derive, voice_signal, zoo
from obs_rvc_amd.resample import FftFixedInOut
from obs_rvc_amd.rvc import RvcInfer
from obs_rvc_amd.streaming import NativeStreamingSession, StreamingSession
g = derive(48000, 0.16, 0.07, 2.0, 4800)
z = zoo("tiny")
def engine():
    e = RvcInfer(z["data"]); e.load_contentvec(2); e.load_f0(); e.load_model(z["model"]); e.set_noise_seed(3, 0)
    return e
e1, e2 = engine(), engine()
nat = NativeStreamingSession(e1, 48000, 0.16, 0.07, 2.0, 4800, 12, 0.6)
pys = StreamingSession(e2, g, 12, 0.6, 4800, lambda ri, ro, n: FftFixedInOut(e2, ri, ro, n))
a = np.interp(np.arange(7680 * 16) / 48000.0, np.arange(2560 * 16) / 16000.0, voice_signal(2560 * 16, seed=10)).astype(np.float32)
t_nat, t_py = [], []
for c in range(16):
    ch = a[c * 7680:(c + 1) * 7680]
    t0 = time.perf_counter(); fn = nat.process_one_frame(ch); t1 = time.perf_counter(); fp = pys.process_one_frame(ch); t2 = time.perf_counter()
    t_nat.append(t1 - t0); t_py.append(t2 - t1)
print(os.environ.get("TAG", ""), "native %.3f ms  python %.3f ms" % (np.median(t_nat[4:]) * 1e3, np.median(t_py[4:]) * 1e3))
