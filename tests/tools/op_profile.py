#!/usr/bin/env python
"""Tuning aid: per-launch timing of the implicit-GEMM class over one chunk (HIP events of the dispatches themselves).

usage: op_profile.py [streams] [preset]     -> one line per launch: us, GFLOP, TF/s, % of the fp32 MFMA peak, description
"""
import ctypes, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from common import BASELINE_160MS as g, voice_signal, zoo
from obs_rvc_amd.rvc import RvcInfer
S = int(sys.argv[1]) if len(sys.argv) > 1 else 1
from common import set_opt
json_out = None
rest = sys.argv[3:]
if "--json" in rest:
    i = rest.index("--json"); json_out = rest[i + 1]; rest = rest[:i] + rest[i + 2:]
bf3 = "--bf3" in rest            # exploratory split-bf16 mode (rvc_set_gemm_precision(e, 1))
rest = [a for a in rest if a != "--bf3"]
for a in rest:                  # test hooks: NAME=VALUE
    set_opt(*a.split("=", 1))
z = zoo(sys.argv[2] if len(sys.argv) > 2 else "full")
eng = RvcInfer(z["data"], device=0); eng.load_contentvec(2); eng.load_f0(1); eng.load_model(z["model"]); eng.set_streams(S); eng.set_noise_seed(1, 0)
if bf3:
    eng.set_gemm_precision(1)
L, chunk, N = g.input_buffer_16k_size, g.sample_frame_16k, g.model_return_size
x = torch.from_numpy(np.stack([voice_signal(L, seed=1 + s) for s in range(S)])).cuda(); out = torch.empty((S, N), device="cuda")
for _ in range(3):
    eng.infer_device(x.data_ptr(), L, chunk, 12, g.skip_head, g.model_return_length, out.data_ptr(), N, sync=True)
eng.set_profile(True)
for _ in range(2):
    eng.infer_device(x.data_ptr(), L, chunk, 12, g.skip_head, g.model_return_length, out.data_ptr(), N, sync=True)
lib = eng._L
lib.rvc_debug_profile_dump.restype = ctypes.c_int
lib.rvc_debug_profile_dump.argtypes = [ctypes.c_void_p, ctypes.c_char_p, ctypes.c_size_t]
buf = ctypes.create_string_buffer(1 << 20)
n = lib.rvc_debug_profile_dump(eng._h, buf, len(buf))
tot_us = tot_gf = 0.0
layers = []
import re
for ln in buf.value.decode().splitlines():
    us, gf, desc = ln.split(" ", 2)
    us, gf = float(us), float(gf)
    tot_us += us; tot_gf += gf
    tf = gf / us * 1e3 if us > 0 else 0.0
    print("%8.2f us %8.4f GF %6.1f TF %5.1f%%  %s" % (us, gf, tf, tf / 157.3 * 100, desc))
    f = dict(re.findall(r"(\w+)=([\w.x]+)", desc))
    lM, lN, lK, nph = int(f.get("M", 0)), int(f.get("N", 0)), int(f.get("K", 0)), int(f.get("nph", 1))
    ksum = float(f.get("ksum", lK * nph))
    layers.append({"launch": len(layers), "kernel": desc.split(" ", 1)[0], "M": lM, "N": lN, "K": lK, "phases": nph, "tile": f.get("tile"), "grid": f.get("grid"),
                   "k_split_waves": int(f.get("ks", 1)), "us": round(us, 2), "gflop": round(gf, 4), "tflops": round(tf, 2), "frac_of_fp32_mfma_peak": round(tf / 157.3, 4),
                   # bytes every launch must move at least once: its weight panel(s) and its output (the input is re-used across taps / phases and not counted)
                   "weight_bytes": int(4 * lM * ksum), "output_bytes": int(4 * lM * lN * nph), "desc": desc})
import time
eng.set_profile(False)
ts = []
for _ in range(40):
    t0 = time.perf_counter()
    eng.infer_device(x.data_ptr(), L, chunk, 12, g.skip_head, g.model_return_length, out.data_ptr(), N, sync=True)
    ts.append(time.perf_counter() - t0)
print("wall ms/chunk median %.4f min %.4f" % (np.median(ts[5:]) * 1e3, min(ts) * 1e3))
print("total %d launches, %.1f us, %.2f GFLOP, %.1f TF/s" % (n, tot_us, tot_gf, tot_gf / tot_us * 1e3))
if json_out:
    import json
    from obs_rvc_amd import _native
    json.dump({"source": "tests/tools/op_profile.py %d %s: HIP events of the dispatches themselves (hipExtLaunchKernelGGL start / stop), one chunk, the two front "
                         "branches issued one after the other above 4 streams" % (S, sys.argv[2] if len(sys.argv) > 2 else "full"),
               "build": _native.binary_hash(), "streams": S, "peak_tflops_fp32_mfma": 157.3, "launches": n, "sum_us": round(tot_us, 1), "gflop": round(tot_gf, 2),
               "tflops": round(tot_gf / tot_us * 1e3, 2), "wall_ms_per_chunk_median": round(float(np.median(ts[5:]) * 1e3), 4),
               "kernel_names": {"reg": "igemm2_kernel (register-direct, 16x16x4 MFMA)", "g32": "igemm32_kernel (LDS-staged activations, 32x32x2 MFMA)",
                                "g32l": "igemm32l_kernel (igemm32 tiles for table-free 1x1 layers, buffer loads with scalar row offsets)", "g32t": "igemm32l_kernel, table variant (offset-table entries as scalar loads)", "c32s": "conv32s_kernel / conv32s_buf_kernel (input rows staged once per workgroup and 32-channel block, 32x32x2 MFMA)", "g32w": "igemm32w_kernel (wide register tiles, 32x32x2 MFMA)", "lds": "igemm_lds_kernel (16x16x4 MFMA)", "ct": "conv_tile_kernel", "g2w": "igemm2w_kernel (register-direct 32x32x2 MFMA, K split in the workgroup)",
                                "bf3": "igemm_bf3_kernel (exploratory: three bf16 MFMAs per fp32 product)", "rmb": "rm_block_kernel (one ConvBlockRes of RMVPE's 16- / 32-channel levels per launch, 16x16x4 MFMA; K = both convolutions + shortcut)"},
               "layers": layers}, open(json_out, "w"), indent=1)
