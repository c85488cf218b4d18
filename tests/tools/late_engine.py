#!/usr/bin/env python
"""Why is a one-stream engine that is created late in a long process slower (v1_256 2.23 ms as the bench's last leg, 2.03 alone)?

One process, three suspects told apart:
  * allocation: engine k of a process (k = 1 .. 8, each created after the previous one was destroyed) -- weights land in recycled blocks;
  * heat / clocks: a one-stream engine that was created FIRST and kept alive, measured again after heavy many-stream load;
  * the load itself: a fresh engine right after the heavy load, and again after an idle pause.
Per measurement: wall median / p99 over 120 synchronised chunks, GPU time of the last chunk, section stamps (f0 branch / ContentVec branch ends),
and rocm-smi clocks where available.

usage: late_engine.py [version]"""
import ctypes, gc, os, subprocess, sys, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from common import BASELINE_160MS as g, set_opt, voice_signal, zoo
from obs_rvc_amd.rvc import RvcInfer

ver = int(sys.argv[1]) if len(sys.argv) > 1 else 2
z = zoo("full", ver)
L, chunk, N = g.input_buffer_16k_size, g.sample_frame_16k, g.model_return_size


def mk(S=1):
    e = RvcInfer(z["data"], device=0); e.load_contentvec(ver); e.load_f0(1); e.load_model(z["model"]); e.set_streams(S); e.set_noise_seed(1, 0)
    return e


def bufs(S):
    x = torch.from_numpy(np.stack([voice_signal(L, seed=1 + s) for s in range(S)])).cuda()
    return x, torch.empty((S, N), device="cuda")


x1, o1 = bufs(1)


def clocks():
    try:
        out = subprocess.run(["rocm-smi", "--showclocks", "--showpower", "--showtemp"], capture_output=True, text=True, timeout=20).stdout
        keep = [ln.strip() for ln in out.splitlines() if any(k in ln for k in ("sclk", "mclk", "Power", "Temperature (Sensor junction)", "fclk"))]
        return " | ".join(k.split(":", 1)[-1].strip() for k in keep[:6])
    except Exception as ex:
        return "rocm-smi unavailable (%s)" % ex


def stamps(e):
    lib = e._L
    lib.rvc_debug_stamps.restype = ctypes.c_int
    lib.rvc_debug_stamps.argtypes = [ctypes.c_void_p, ctypes.c_char_p, ctypes.c_size_t]
    buf = ctypes.create_string_buffer(1 << 16)
    lib.rvc_debug_stamps(e._h, buf, len(buf))
    d = {}
    for ln in buf.value.decode().splitlines():
        name, t = ln.rsplit(" ", 1)
        d.setdefault(name, float(t))
    return d


def measure(e, tag, n=120):
    for _ in range(15):
        e.infer_device(x1.data_ptr(), L, chunk, 12, g.skip_head, g.model_return_length, o1.data_ptr(), N, sync=True)
    ts, gm = [], []
    for _ in range(n):
        t0 = time.perf_counter()
        e.infer_device(x1.data_ptr(), L, chunk, 12, g.skip_head, g.model_return_length, o1.data_ptr(), N, sync=True)
        ts.append(time.perf_counter() - t0); gm.append(e.last_gpu_ms())
    st = stamps(e)
    keys = [k for k in st if any(w in k for w in ("f0", "cv", "rmvpe", "contentvec", "join", "synth", "end"))]
    print("%-58s wall p50 %.4f p99 %.4f | gpu p50 %.4f | %s | %s" % (tag, np.median(ts) * 1e3, np.percentile(ts, 99) * 1e3, np.median(gm),
          " ".join("%s=%.0f" % (k, st[k]) for k in list(st)[:14]), clocks()), flush=True)
    return float(np.median(gm))


set_opt("RVC_STAMPS", "1")
keeper = mk()
base = measure(keeper, "keeper (first engine of the process)")
for k in range(2, 7):
    e = mk(); measure(e, "engine %d of the process (previous ones destroyed)" % k); e.close(); del e; gc.collect(); torch.cuda.empty_cache()
measure(keeper, "keeper again (after 5 create / destroy cycles)")
# heavy load: 64 streams for ~6 s
h = mk(64); x64, o64 = bufs(64)
t0 = time.perf_counter(); nst = 0
while time.perf_counter() - t0 < 6.0:
    h.infer_device(x64.data_ptr(), L, chunk, 12, g.skip_head, g.model_return_length, o64.data_ptr(), N, sync=True); nst += 1
print("heavy load: %d steps of 64 streams" % nst, flush=True)
measure(keeper, "keeper right after the heavy load (64-stream engine alive)")
h.close(); del h, x64, o64; gc.collect(); torch.cuda.empty_cache()
measure(keeper, "keeper, heavy engine destroyed")
e = mk(); measure(e, "fresh engine after the heavy load")
time.sleep(5.0)
measure(e, "the same fresh engine after 5 s idle")
measure(keeper, "keeper after 5 s idle")
e.close(); keeper.close()
