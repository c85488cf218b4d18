"""Tuning aid: run one layer shape N times (for rocprofv3 --pmc passes)."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from obs_rvc_amd import _native
from gemm_microbench import SHAPES
L = _native.lib()
L.rvc_debug_conv_bench.restype = C.c_double
L.rvc_debug_conv_bench.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]
h = C.c_void_p(); assert L.rvc_create(b"/tmp", 0, C.byref(h)) == 0
for name in sys.argv[1].split(","):
    M, Cin, KW, dil, N = SHAPES[name]
    print(name, L.rvc_debug_conv_bench(h, M, Cin, KW, dil, N, 20, 0))
