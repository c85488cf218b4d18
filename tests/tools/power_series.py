#!/usr/bin/env python
"""How the box reports power / clock over time under a known load: a 64-stream leg for ~6 s, sysfs hwmon (power1_input, freq1_input, temp2_input) and amdsmi
sampled every 50 ms from a thread, the in-kernel clock monitor around it.  Prints the series (bench.py's BoxProbe is built on what this shows).
usage (gpurun): python tests/tools/power_series.py [streams] [seconds]"""
import glob, json, os, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
import numpy as np
import torch
from common import BASELINE_160MS as g, voice_signal, zoo
from obs_rvc_amd import _native
from obs_rvc_amd.rvc import RvcInfer

S = int(sys.argv[1]) if len(sys.argv) > 1 else 64
secs = float(sys.argv[2]) if len(sys.argv) > 2 else 6.0
hw = sorted(glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*"))
print("hwmon:", hw)
smi = None
try:
    import amdsmi
    amdsmi.amdsmi_init()
    smi = amdsmi.amdsmi_get_processor_handles()[0]
except Exception as ex:
    print("amdsmi unavailable:", ex)


def rd(p):
    try:
        return int(open(p).read().strip())
    except Exception:
        return None


rows, stop = [], threading.Event()


def sampler():
    t0 = time.perf_counter()
    while not stop.is_set():
        r = {"t": round(time.perf_counter() - t0, 3)}
        if hw:
            r["W"] = (rd(hw[0] + "/power1_input") or 0) / 1e6; r["sclk"] = (rd(hw[0] + "/freq1_input") or 0) / 1e6; r["Tj"] = (rd(hw[0] + "/temp2_input") or 0) / 1e3
        if smi is not None:
            try:
                pi = amdsmi.amdsmi_get_power_info(smi)
                r["smiW"] = pi.get("current_socket_power") or pi.get("average_socket_power")
                ci = amdsmi.amdsmi_get_clock_info(smi, amdsmi.AmdSmiClkType.GFX)
                r["smi_clk"] = ci.get("clk") or ci.get("cur_clk")
            except Exception as ex:
                r["smi_err"] = str(ex)[:60]
        rows.append(r)
        stop.wait(0.05)


z = zoo("full")
eng = RvcInfer(z["data"]); eng.load_contentvec(2); eng.load_f0(); eng.load_model(z["model"]); eng.set_streams(S); eng.set_noise_seed(1, 0)
x = np.stack([voice_signal(g.input_buffer_16k_size, seed=s) for s in range(S)])
d_in = torch.from_numpy(x).cuda(); d_out = torch.empty((S, g.model_return_size), dtype=torch.float32, device="cuda")
step = lambda: eng.infer_device(d_in.data_ptr(), g.input_buffer_16k_size, g.sample_frame_16k, 12, g.skip_head, g.model_return_length, d_out.data_ptr(), g.model_return_size, sync=True)
for _ in range(3):
    step()
th = threading.Thread(target=sampler, daemon=True); th.start()
time.sleep(1.0)                       # idle second
_native.clock_monitor_start(0)
t0 = time.perf_counter(); n = 0
while time.perf_counter() - t0 < secs:
    step(); n += 1
el = time.perf_counter() - t0
mon = _native.clock_monitor_stop(0)
time.sleep(1.5)                       # idle again
stop.set(); th.join()
print("streams %d: %d steps, %.3f ms per step, monitor %s" % (S, n, el / n * 1e3, json.dumps(mon)))
for r in rows[::2]:
    print(json.dumps(r))
