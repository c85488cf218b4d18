"""Loader for the in-tree C-ABI library (csrc/librvc_mi355x.so).  No fallback: if the HIP
extension is missing or no GPU is present the product path raises."""
from __future__ import annotations

import ctypes as C
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, "csrc")
SO_PATH = os.path.join(CSRC, "librvc_mi355x.so")
RPC_PATH = os.path.join(CSRC, "rvc-rpc")
_LIB = None

# every symbol include/rvc_mi355x.h declares
SYMBOLS = [
    "rvc_create", "rvc_destroy", "rvc_load_contentvec", "rvc_load_model", "rvc_load_f0", "rvc_unload_model",
    "rvc_hubert", "rvc_extract_feature", "rvc_pitch", "rvc_infer", "rvc_last_error_message",
    "rvc_load_index", "rvc_load_index_device", "rvc_set_index_rate", "rvc_get_knn", "rvc_set_noise_seed", "rvc_reset_state",
    "rvc_set_streams", "rvc_infer_batch", "rvc_infer_device", "rvc_infer_batch_v", "rvc_infer_device_v", "rvc_infer_batch_g", "rvc_synchronize", "rvc_set_use_graph", "rvc_set_pipeline",
    "rvc_last_gpu_ms", "rvc_profile_last", "rvc_set_profile", "rvc_enable_taps", "rvc_get_tap", "rvc_get_pitch_cache",
    "rvc_index_device_ptr", "rvc_device", "rvc_version", "rvc_envelop_mixing", "rvc_sola_step", "rvc_profile_last_knn",
    "rvc_resampler_create", "rvc_resampler_destroy", "rvc_resampler_input_frames_next", "rvc_resampler_output_frames_max",
    "rvc_resampler_reset", "rvc_resampler_process", "rvc_resampler_process_device",
    "rvc_rccl_unique_id", "rvc_index_broadcast", "rvc_rccl_available", "rvc_index_broadcast_info",
    "rvc_set_plan_cache", "rvc_plan_cache_info", "rvc_set_plan_autotune", "rvc_plan_autotune_info", "rvc_retrieval_recoveries", "rvc_set_gemm_precision",
    "rvc_calibrate", "rvc_clock_monitor_start", "rvc_clock_monitor_stop",
    "rvc_session_create", "rvc_session_destroy", "rvc_session_process", "rvc_session_frame_size", "rvc_session_set_params", "rvc_session_set_params_stream", "rvc_session_geometry",
]


SOURCES = ("engine.hip", "engine_int.h", "plan.hip", "model_cv.hip", "model_rmvpe.hip", "model_synth.hip", "retrieval.hip", "kernels.hip.h", "igemm.hip.h", "igemm_launch.h", "igemm2_inst.hip", "igemm_tiled_inst.hip", "igemm2w_inst.hip", "igemm_bf3_inst.hip", "conv_tile.hip.h", "conv_tile_inst.hip", "conv32s.hip.h", "conv32s_inst.hip", "rmblock.hip.h", "igemm32l.hip.h", "igemm32l_inst.hip", "version.cpp", "calib.hip", "exports.map",
           "state.hip.h",
           "resample.hip.h", "session.hip.h", "rccl_bcast.hip.h", "blob.h", "rvc_rpc.cpp")

# translation units of the library: (source, extra flags, files whose contents decide whether the object is stale).  The implicit-GEMM
# template instantiations are the bulk of the compile time; as separate units they build in parallel (5 min -> about 1.5 min on 8 cores)
# and are not rebuilt when only the engine changes.
_IGEMM_DEPS = ("igemm.hip.h", "igemm_launch.h")
_INT_DEPS = ("engine_int.h", "kernels.hip.h", "igemm.hip.h", "igemm_launch.h", "blob.h", "state.hip.h")
_ENGINE_DEPS = ("engine.hip", "resample.hip.h", "session.hip.h", "rccl_bcast.hip.h") + _INT_DEPS
UNITS = [("engine.hip", [], _ENGINE_DEPS), ("calib.hip", [], ("calib.hip",))] + [(u, [], (u,) + _INT_DEPS + (("rmblock.hip.h",) if u == "model_rmvpe.hip" else ())) for u in ("plan.hip", "model_cv.hip", "model_rmvpe.hip", "model_synth.hip", "retrieval.hip")] + \
        [("igemm2_inst.hip", ["-DRVC_IGEMM2_CFG=%d" % c], ("igemm2_inst.hip",) + _IGEMM_DEPS) for c in range(5)] + \
        [("igemm_tiled_inst.hip", ["-DRVC_TILED_PART=%d" % c], ("igemm_tiled_inst.hip",) + _IGEMM_DEPS) for c in range(4)] + \
        [("igemm2w_inst.hip", ["-DRVC_G2W_PART=%d" % c], ("igemm2w_inst.hip",) + _IGEMM_DEPS) for c in range(3)] + \
        [("igemm_bf3_inst.hip", [], ("igemm_bf3_inst.hip",) + _IGEMM_DEPS)] + \
        [("conv_tile_inst.hip", [], ("conv_tile_inst.hip", "conv_tile.hip.h") + _IGEMM_DEPS)] + \
        [("conv32s_inst.hip", ["-DRVC_C32S_PART=%d" % c], ("conv32s_inst.hip", "conv32s.hip.h") + _IGEMM_DEPS) for c in range(3)] + \
        [("igemm32l_inst.hip", ["-DRVC_G32L_PART=%d" % c], ("igemm32l_inst.hip", "igemm32l.hip.h") + _IGEMM_DEPS) for c in range(2)]
HIPCC_FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC"]
# Object cache: content-addressed (sources + flags).  It lives under the repository's build/ directory (git- and gpurun-ignored), is
# created 0700, and a directory that is not ours (other owner, or writable by group / others) is refused: objects are linked straight
# into the product library, so nobody else may be able to plant one.
OBJ_CACHE = os.environ.get("RVC_OBJ_CACHE", os.path.join(os.path.dirname(_HERE), "build", "obj_cache"))


def _private_cache_dir() -> str:
    os.makedirs(OBJ_CACHE, mode=0o700, exist_ok=True)
    st = os.stat(OBJ_CACHE)
    if st.st_uid != os.getuid() or (st.st_mode & 0o022):
        raise RuntimeError("object cache %s is not private to uid %d (owner %d, mode %o): refusing to link objects from it"
                           % (OBJ_CACHE, os.getuid(), st.st_uid, st.st_mode & 0o777))
    return OBJ_CACHE


def _tmp_name(path: str) -> str:
    """unique per builder (several ranks may call build() at once); published with os.replace, which is atomic"""
    import uuid
    return "%s.%d.%s.tmp.o" % (path, os.getpid(), uuid.uuid4().hex[:8])


def source_hash() -> str:
    """sha256 over the library's sources (+ the public header), first 16 hex digits.  It is compiled into the binary
    (`rvc_version()` ends in "src:<hash>"), so a tested .so can be tied to the sources it was built from."""
    import hashlib
    h = hashlib.sha256()
    for f in [os.path.join(CSRC, n) for n in SOURCES] + [os.path.join(os.path.dirname(_HERE), "include", "rvc_mi355x.h")]:
        h.update(os.path.basename(f).encode() + b"\0")
        with open(f, "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


def binary_hash(path: str = None) -> str:
    """The source hash embedded in a built library ("" if the file is missing or predates the scheme); read from the file's bytes,
    so no GPU and no dlopen is needed."""
    path = path or SO_PATH
    if not os.path.exists(path):
        return ""
    with open(path, "rb") as fh:
        blob = fh.read()
    i = blob.find(b"rvc-mi355x-src:")
    return blob[i + 15:i + 31].decode("ascii", "replace") if i >= 0 else ""


def _unit_key(src, flags, deps, extra_flags):
    import hashlib
    h = hashlib.sha256(" ".join(HIPCC_FLAGS + list(flags) + list(extra_flags)).encode())
    for n in tuple(deps) + ("../../include/rvc_mi355x.h",):
        with open(os.path.join(CSRC, n), "rb") as fh:
            h.update(n.encode() + b"\0" + fh.read())
    return "%s-%s" % (os.path.splitext(src)[0], h.hexdigest()[:20])


def compile_units(extra_flags=(), verbose=False, extra_units=()):
    """Compile every translation unit whose object is not in the cache (content-addressed: sources + flags), in parallel.
    -> list of object paths"""
    from concurrent.futures import ThreadPoolExecutor
    _private_cache_dir()
    jobs, objs = [], []
    for src, flags, deps in list(UNITS) + list(extra_units):
        obj = os.path.join(OBJ_CACHE, _unit_key(src, flags, tuple(deps), extra_flags) + ".o")
        objs.append(obj)
        if not os.path.exists(obj):
            tmp = _tmp_name(obj)
            jobs.append((["hipcc"] + HIPCC_FLAGS + list(flags) + list(extra_flags) + ["-c", os.path.join(CSRC, src), "-o", tmp], obj, tmp))

    def run(job):
        cmd, obj, tmp = job
        if verbose:
            print(" ".join(cmd), flush=True)
        try:
            subprocess.check_call(cmd)
            os.replace(tmp, obj)
        finally:
            if os.path.exists(tmp):
                os.unlink(tmp)
    if jobs:
        with ThreadPoolExecutor(max_workers=max(1, min(len(jobs), os.cpu_count() or 1))) as ex:
            list(ex.map(run, jobs))
    return objs


def link_library(objs, out, want_hash, verbose=False):
    ver = _tmp_name(os.path.join(_private_cache_dir(), "version-%s" % want_hash))
    tmp_out = "%s.%d.tmp" % (out, os.getpid())
    try:
        cmd = ["hipcc", "-O2", "-fPIC", "-c", '-DRVC_SRC_HASH="%s"' % want_hash, os.path.join(CSRC, "version.cpp"), "-o", ver]
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)
        # exports.map: only the C ABI (rvc_*) is visible; the planner's C++ internals and the kernels' host stubs stay local
        cmd = ["hipcc", "--offload-arch=gfx950", "-shared", "-fPIC"] + objs + [ver, "-o", tmp_out, "-ldl", "-Wl,--version-script=" + os.path.join(CSRC, "exports.map")]
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)
        os.replace(tmp_out, out)
    finally:
        for f in (ver, tmp_out):
            if os.path.exists(f):
                os.unlink(f)


def build(force: bool = False, verbose: bool = False) -> str:
    """hipcc --offload-arch=gfx950: the engine library (one object per translation unit, built in parallel, linked into
    csrc/librvc_mi355x.so) and the rvc-rpc protocol-compatible executable.  A library is reused only if the source hash compiled
    into it equals the hash of the sources on disk (never by modification time)."""
    want = source_hash()
    have = binary_hash()
    if force or have != want:
        if verbose:
            print("source hash %s, binary %s -> rebuilding" % (want, have or "(none)"))
        if force:
            import shutil
            shutil.rmtree(OBJ_CACHE, ignore_errors=True)
        link_library(compile_units(verbose=verbose), SO_PATH, want, verbose)
    elif verbose:
        print("librvc_mi355x.so carries source hash %s = sources on disk: up to date" % have)
    rpc_src = os.path.join(CSRC, "rvc_rpc.cpp")
    if force or have != want or not os.path.exists(RPC_PATH) or os.path.getmtime(RPC_PATH) < os.path.getmtime(rpc_src):
        cmd = ["hipcc", "-O2", "-std=c++17", rpc_src, "-o", RPC_PATH, "-L" + CSRC, "-lrvc_mi355x", "-Wl,-rpath,$ORIGIN"]
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)
    return SO_PATH


def lib():
    global _LIB
    if _LIB is not None:
        return _LIB
    if not os.path.exists(SO_PATH):
        raise RuntimeError("HIP extension %s is missing: run `python -c 'import __graft_entry__ as g; g.build()'`" % SO_PATH)
    # RVC_LIB_OVERRIDE (A/B timing of an older build) is honoured only together with RVC_TUNING=1: outside the tuning tools nothing but the
    # in-tree library is ever loaded
    override = os.environ.get("RVC_LIB_OVERRIDE") if os.environ.get("RVC_TUNING") == "1" else None
    L = C.CDLL(override or SO_PATH)
    vp, fp, sz, i32, u32 = C.c_void_p, C.POINTER(C.c_float), C.c_size_t, C.c_int32, C.c_uint32
    L.rvc_create.argtypes = [C.c_char_p, C.c_int, C.POINTER(vp)]
    L.rvc_destroy.argtypes = [vp]
    L.rvc_destroy.restype = None
    L.rvc_load_contentvec.argtypes = [vp, C.c_int]
    L.rvc_load_model.argtypes = [vp, C.c_char_p]
    L.rvc_load_f0.argtypes = [vp, C.c_int]
    L.rvc_unload_model.argtypes = [vp]
    L.rvc_unload_model.restype = None
    L.rvc_hubert.argtypes = [vp, fp, sz, fp, sz, C.POINTER(sz)]
    L.rvc_extract_feature.argtypes = [vp, fp, sz, fp, sz, C.POINTER(sz)]
    L.rvc_pitch.argtypes = [vp, fp, sz, i32, sz, fp, sz, C.POINTER(sz)]
    L.rvc_infer.argtypes = [vp, fp, sz, sz, C.c_int, i32, u32, u32, fp, sz, C.POINTER(sz)]
    L.rvc_last_error_message.argtypes = [vp]
    L.rvc_last_error_message.restype = C.c_char_p
    L.rvc_load_index.argtypes = [vp, fp, sz, sz]
    L.rvc_load_index_device.argtypes = [vp, vp, sz, sz]
    L.rvc_set_index_rate.argtypes = [vp, C.c_float]
    L.rvc_set_index_rate.restype = None
    L.rvc_get_knn.argtypes = [vp, C.POINTER(i32), fp, sz, C.POINTER(sz)]
    L.rvc_set_noise_seed.argtypes = [vp, u32, u32]
    L.rvc_set_noise_seed.restype = None
    L.rvc_reset_state.argtypes = [vp]
    L.rvc_reset_state.restype = None
    L.rvc_set_streams.argtypes = [vp, C.c_int]
    L.rvc_infer_batch.argtypes = [vp, fp, sz, sz, i32, u32, u32, fp, sz, C.POINTER(sz)]
    L.rvc_infer_device.argtypes = [vp, vp, sz, sz, i32, u32, u32, vp, sz, C.POINTER(sz), C.c_int]
    L.rvc_infer_batch_v.argtypes = [vp, fp, sz, sz, C.POINTER(i32), u32, u32, fp, sz, C.POINTER(sz)]
    L.rvc_infer_device_v.argtypes = [vp, vp, sz, sz, C.POINTER(i32), u32, u32, vp, sz, C.POINTER(sz), C.c_int]
    L.rvc_infer_batch_g.argtypes = [vp, C.POINTER(fp), C.POINTER(sz), C.POINTER(sz), C.POINTER(i32), C.POINTER(u32), C.POINTER(u32), C.POINTER(fp), C.POINTER(sz), C.POINTER(sz)]
    L.rvc_synchronize.argtypes = [vp]
    L.rvc_set_use_graph.argtypes = [vp, C.c_int]
    L.rvc_set_use_graph.restype = None
    if hasattr(L, "rvc_set_plan_cache") or not override:      # (an older build loaded for an A/B timing lacks the newer entry points)
        L.rvc_set_plan_cache.argtypes = [vp, C.c_int]
        L.rvc_plan_cache_info.argtypes = [vp, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_longlong)]
        L.rvc_retrieval_recoveries.argtypes = [vp]
        L.rvc_retrieval_recoveries.restype = C.c_longlong
    if hasattr(L, "rvc_set_gemm_precision") or not override:
        L.rvc_set_gemm_precision.argtypes = [vp, C.c_int]
    L.rvc_set_pipeline.argtypes = [vp, C.c_int]
    L.rvc_set_pipeline.restype = None
    L.rvc_last_gpu_ms.argtypes = [vp]
    L.rvc_last_gpu_ms.restype = C.c_float
    L.rvc_profile_last.argtypes = [vp, C.POINTER(C.c_int), C.POINTER(C.c_double), C.POINTER(C.c_double)]
    L.rvc_profile_last_knn.argtypes = [vp, C.POINTER(C.c_int), C.POINTER(C.c_double), C.POINTER(C.c_double)]
    L.rvc_set_profile.argtypes = [vp, C.c_int]
    L.rvc_set_profile.restype = None
    L.rvc_enable_taps.argtypes = [vp, C.c_int]
    L.rvc_enable_taps.restype = None
    L.rvc_get_tap.argtypes = [vp, C.c_char_p, fp, sz, C.POINTER(sz)]
    L.rvc_get_pitch_cache.argtypes = [vp, C.c_int, fp]
    L.rvc_get_pitch_cache.restype = None
    L.rvc_index_device_ptr.argtypes = [vp, C.POINTER(sz)]
    L.rvc_index_device_ptr.restype = vp
    L.rvc_device.argtypes = [vp]
    L.rvc_version.restype = C.c_char_p
    L.rvc_envelop_mixing.argtypes = [vp, fp, fp, sz, sz, C.c_double]
    L.rvc_sola_step.argtypes = [vp, fp, sz, fp, sz, sz, sz, fp, C.POINTER(sz)]
    L.rvc_resampler_create.argtypes = [vp, sz, sz, sz, C.POINTER(vp)]
    L.rvc_resampler_destroy.argtypes = [vp]
    L.rvc_resampler_destroy.restype = None
    L.rvc_resampler_input_frames_next.argtypes = [vp]
    L.rvc_resampler_input_frames_next.restype = sz
    L.rvc_resampler_output_frames_max.argtypes = [vp]
    L.rvc_resampler_output_frames_max.restype = sz
    L.rvc_resampler_reset.argtypes = [vp]
    L.rvc_resampler_reset.restype = None
    L.rvc_resampler_process.argtypes = [vp, fp, sz, fp, sz, C.POINTER(sz)]
    L.rvc_resampler_process_device.argtypes = [vp, vp, vp, C.c_int]
    if hasattr(L, "rvc_rccl_unique_id") or not override:
        L.rvc_rccl_unique_id.argtypes = [vp]
        L.rvc_index_broadcast.argtypes = [vp, vp, C.c_int, C.c_int, fp, sz, sz]
        L.rvc_index_broadcast_info.argtypes = [vp, C.POINTER(C.c_double), C.POINTER(C.c_int)]
    L.rvc_session_create.argtypes = [vp, sz, C.c_double, C.c_double, C.c_double, sz, i32, C.c_double, C.c_int, C.POINTER(vp)]
    L.rvc_session_destroy.argtypes = [vp]
    L.rvc_session_destroy.restype = None
    L.rvc_session_process.argtypes = [vp, fp, sz, fp, sz, C.POINTER(sz)]
    L.rvc_session_frame_size.argtypes = [vp]
    L.rvc_session_frame_size.restype = sz
    L.rvc_session_set_params.argtypes = [vp, i32, C.c_double]
    L.rvc_session_set_params.restype = None
    L.rvc_session_set_params_stream.argtypes = [vp, C.c_int, i32, C.c_double]
    L.rvc_session_geometry.argtypes = [vp, C.POINTER(i32)]
    L.rvc_session_geometry.restype = None
    if hasattr(L, "rvc_set_plan_autotune") or not override:
        L.rvc_set_plan_autotune.argtypes = [vp, C.c_int]
        L.rvc_plan_autotune_info.argtypes = [vp, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_double), C.POINTER(C.c_double)]
    if hasattr(L, "rvc_calibrate") or not override:
        L.rvc_calibrate.argtypes = [C.c_int, C.POINTER(Calibration)]
        L.rvc_clock_monitor_start.argtypes = [C.c_int]
        L.rvc_clock_monitor_stop.argtypes = [C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_double)]
    _LIB = L
    return L


class Calibration(C.Structure):
    """rvc_calibration (include/rvc_mi355x.h)"""
    _fields_ = [("mfma_f32_tflops", C.c_double), ("mfma_sclk_mhz", C.c_double), ("mfma_ms", C.c_double),
                ("hbm_read_tbs", C.c_double), ("hbm_sclk_mhz", C.c_double), ("ms_total", C.c_double), ("compute_units", C.c_int)]


def calibrate(device: int = 0) -> dict:
    """rvc_calibrate: what this GPU sustains right now (bare fp32-MFMA stream, HBM read stream) -> dict"""
    c = Calibration()
    rc = lib().rvc_calibrate(device, C.byref(c))
    if rc != 0:
        raise RuntimeError("rvc_calibrate failed (%d)" % rc)
    return {k: getattr(c, k) for k, _ in Calibration._fields_}


def clock_monitor_start(device: int = 0) -> None:
    if lib().rvc_clock_monitor_start(device) != 0:
        raise RuntimeError("rvc_clock_monitor_start failed")


def clock_monitor_stop(device: int = 0) -> dict:
    mean, mn, secs = C.c_double(), C.c_double(), C.c_double()
    if lib().rvc_clock_monitor_stop(device, C.byref(mean), C.byref(mn), C.byref(secs)) != 0:
        raise RuntimeError("rvc_clock_monitor_stop failed")
    return {"sclk_mhz_mean": mean.value, "sclk_mhz_min": mn.value, "seconds": secs.value}
