#!/usr/bin/env python
"""Tuning aid: one-stream layers with N in the thousands on the workgroup-tiled 32x32x2 kernel versus the register-direct kernel."""
import os, subprocess, sys
SH = {"cv_conv1": (512, 512, 3, 1, 3583), "cv_conv2": (512, 512, 3, 1, 1791), "cv_conv3": (512, 512, 3, 1, 895), "dec0": (128, 128, 11, 1, 2520), "dec1": (64, 64, 11, 1, 5040), "dec2": (32, 32, 11, 1, 10080)}
if len(sys.argv) > 1 and sys.argv[1] == "child":
    import ctypes as C
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
    os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
    from obs_rvc_amd import _native
    L = _native.lib()
    L.rvc_debug_conv_bench.restype = C.c_double
    L.rvc_debug_conv_bench.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]
    h = C.c_void_p()
    assert L.rvc_create(b"/tmp", 0, C.byref(h)) == 0
    out = []
    for name, (M, Cin, KW, dil, N) in SH.items():
        us = L.rvc_debug_conv_bench(h, M, Cin, KW, dil, N, 100, 0)
        out.append("%s %.1fus %.0fTF" % (name, us, 2.0 * M * Cin * KW * N / us / 1e6))
    print("%-28s" % sys.argv[2], " | ".join(out), flush=True)
else:
    for tag, env in (("register-direct", {"RVC_GEMM32": "0"}), ("g32 auto tile", {"RVC_GEMM32_MIN": "1"}), ("g32 BM=64", {"RVC_GEMM32_MIN": "1", "RVC_G32_BM": "64"}),
                     ("g32 BM=32", {"RVC_GEMM32_MIN": "1", "RVC_G32_BM": "32"})):
        e = dict(os.environ); e.update(env)
        subprocess.run([sys.executable, __file__, "child", tag], env=e)
