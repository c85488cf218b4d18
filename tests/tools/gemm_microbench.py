"""Tuning aid: isolated timings of the implicit-GEMM kernel on representative layer shapes (needs the tuning build:
RVC_TUNING=1 RVC_LIB_OVERRIDE=$(python tests/tools/build_tuning.py) python tests/tools/gemm_microbench.py [quick])."""
import ctypes as C
import os
import subprocess
import sys

SHAPES = {  # name: (M, Cin, KW, dil, N)
    "cv_qkv": (2304, 768, 1, 1, 111), "cv_o": (768, 768, 1, 1, 111), "cv_ff1": (3072, 768, 1, 1, 111), "cv_ff2": (768, 3072, 1, 1, 111),
    "hg3_k11": (32, 32, 11, 1, 10080), "hg3_k3": (32, 32, 3, 5, 10080), "hg2_k11": (64, 64, 11, 1, 5040), "hg1_k7": (128, 128, 7, 3, 2520),
    "hg0_k11": (256, 256, 11, 1, 252), "rm_l5x64": (512, 512, 3, 1, 256), "rm_l4x64": (256, 256, 9, 1, 1024), "rm_l3x64": (128, 128, 9, 1, 4096), "cv_conv2": (512, 512, 3, 1, 1791), "enc_ff1": (768, 192, 3, 1, 21),
    "d128_k3": (128, 128, 3, 1, 2520), "d128_k7": (128, 128, 7, 1, 2520), "d128_k11": (128, 128, 11, 1, 2520), "d128_k11d5": (128, 128, 11, 5, 2520),
    "d64_k3": (64, 64, 3, 1, 5040), "d64_k7": (64, 64, 7, 1, 5040), "d64_k11": (64, 64, 11, 1, 5040), "d64_k11d5": (64, 64, 11, 5, 5040),
    "d32_k3": (32, 32, 3, 1, 10080), "d32_k7": (32, 32, 7, 1, 10080), "d32_k11": (32, 32, 11, 1, 10080), "d32_k11d5": (32, 32, 11, 5, 10080),
}
if __name__ == "__main__" and len(sys.argv) > 1 and sys.argv[1] == "child":
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
    os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
    from obs_rvc_amd import _native
    L = _native.lib()
    L.rvc_debug_conv_bench.restype = C.c_double
    L.rvc_debug_conv_bench.argtypes = [C.c_void_p] + [C.c_int] * 9
    h = C.c_void_p()
    assert L.rvc_create(b"/tmp", 0, C.byref(h)) == 0
    out = []
    for name in sys.argv[2].split(","):
        M, Cin, KW, dil, N = SHAPES[name]
        us = L.rvc_debug_conv_bench(h, M, Cin, KW, dil, N, 200 if int(os.environ.get("RVC_BENCH_B", "1")) == 1 else 20, 0, int(os.environ.get("RVC_BENCH_B", "1")), 0)
        fl = 2.0 * M * Cin * KW * N * int(os.environ.get("RVC_BENCH_B", "1"))
        out.append("%s %.1fus %.1fTF" % (name, us, fl / us / 1e6))
    print(os.environ.get("RVC_FORCE_CFG", "auto"), os.environ.get("RVC_FORCE_MFAST", "-"), " | ".join(out))
elif __name__ == "__main__":
    names = ",".join(SHAPES)
    cfgs = [None] + ["%d,%d" % (c, k) for c in (0, 3, 4) for k in (1, 4, 8)] + ["0,16", "1,4", "2,4"]
    if len(sys.argv) > 1 and sys.argv[1] == "quick":
        cfgs = [None]
    for cfg in cfgs:
        for mf in ("0",) if len(sys.argv) > 1 else ("0", "1"):
            env = dict(os.environ)
            if cfg:
                env["RVC_FORCE_CFG"] = cfg
            env["RVC_FORCE_MFAST"] = mf
            subprocess.run([sys.executable, __file__, "child", names], env=env)
