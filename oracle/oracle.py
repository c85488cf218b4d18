"""ctypes binding of the CPU oracle (oracle/rvc_oracle.c).  TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

ERR_NAMES = {0: "Ok", 1: "ModelNotLoaded", 2: "ContentvecNotLoaded", 3: "F0NotLoaded", 4: "Backend", 5: "Shape", 6: "Panic"}


class OracleError(RuntimeError):
    def __init__(self, code, msg=""):
        super().__init__("%s: %s" % (ERR_NAMES.get(code, code), msg))
        self.code = code


def build(force: bool = False) -> str:
    so = os.path.join(_HERE, "librvc_oracle.so")
    src = os.path.join(_HERE, "rvc_oracle.c")
    if force or not os.path.exists(so) or (os.path.exists(src) and os.path.getmtime(src) > os.path.getmtime(so)):
        subprocess.check_call(["make", "-C", _HERE, "-B", "librvc_oracle.so"], stdout=subprocess.DEVNULL)
    return so


def lib():
    global _LIB
    if _LIB is None:
        so = os.path.join(_HERE, "librvc_oracle.so")
        if not os.path.exists(so):
            build()
        L = C.CDLL(so)
        fp, sz, u32, i32p = C.POINTER(C.c_float), C.c_size_t, C.c_uint32, C.POINTER(C.c_int32)
        L.ora_new.restype = C.c_void_p
        L.ora_new.argtypes = [C.c_char_p]
        L.ora_free.argtypes = [C.c_void_p]
        L.ora_load_contentvec.argtypes = [C.c_void_p, C.c_int]
        L.ora_load_model.argtypes = [C.c_void_p, C.c_char_p]
        L.ora_load_f0.argtypes = [C.c_void_p, C.c_int]
        L.ora_unload_model.argtypes = [C.c_void_p]
        L.ora_hubert.argtypes = [C.c_void_p, fp, sz, fp, sz, C.POINTER(sz)]
        L.ora_extract_feature.argtypes = [C.c_void_p, fp, sz, fp, sz, C.POINTER(sz)]
        L.ora_pitch.argtypes = [C.c_void_p, fp, sz, C.c_int, sz, fp, sz, C.POINTER(sz)]
        L.ora_infer.argtypes = [C.c_void_p, fp, sz, sz, C.c_int, C.c_int, u32, u32, fp, sz, C.POINTER(sz)]
        L.ora_last_error.restype = C.c_char_p
        L.ora_last_error.argtypes = [C.c_void_p]
        L.ora_load_index.argtypes = [C.c_void_p, fp, sz, sz]
        L.ora_set_index_rate.argtypes = [C.c_void_p, C.c_float]
        L.ora_set_noise_seed.argtypes = [C.c_void_p, u32, u32]
        L.ora_reset_state.argtypes = [C.c_void_p]
        L.ora_get_pitch_cache.argtypes = [C.c_void_p, fp]
        L.ora_get_knn.argtypes = [C.c_void_p, i32p, fp, sz, C.POINTER(sz)]
        L.ora_get_tap.argtypes = [C.c_void_p, C.c_char_p, C.POINTER(fp), C.POINTER(sz)]
        L.ora_enable_taps.argtypes = [C.c_void_p, C.c_int]
        L.ora_set_threads.argtypes = [C.c_int]
        L.ora_hann_periodic.argtypes = [sz, fp]
        L.ora_pad_reflect.argtypes = [fp, sz, sz, fp]
        L.ora_stft.restype = sz
        L.ora_stft.argtypes = [fp, sz, sz, sz, fp, C.c_int, fp]
        L.ora_mel_filterbank.argtypes = [C.c_double, sz, sz, C.c_double, C.c_double, fp]
        L.ora_mel_extract.restype = sz
        L.ora_mel_extract.argtypes = [fp, sz, fp]
        L.ora_decode.argtypes = [fp, sz, C.c_float, fp]
        L.ora_get_f0_post.argtypes = [fp, sz, i32p]
        L.ora_uppower.restype = C.c_float
        L.ora_uppower.argtypes = [C.c_int]
        L.ora_f0_extractor_frame.restype = sz
        L.ora_f0_extractor_frame.argtypes = [sz]
        L.ora_knn_search.argtypes = [fp, sz, sz, fp, sz, C.c_int, i32p, fp]
        L.ora_philox_normal.argtypes = [u32, u32, u32, u32, sz, fp]
        L.ora_rms.restype = sz
        L.ora_rms.argtypes = [fp, sz, sz, sz, fp]
        L.ora_lerp_align_corners.argtypes = [fp, sz, sz, fp]
        L.ora_envelop_mixing.argtypes = [fp, fp, sz, sz, C.c_double]
        L.ora_sola_offset.restype = sz
        L.ora_sola_offset.argtypes = [fp, fp, sz, sz]
        L.ora_sola_step.restype = sz
        L.ora_sola_step.argtypes = [fp, sz, fp, sz, sz, sz, fp]
        L.ora_set_threads(max(1, min(int(os.environ.get('RVC_ORACLE_THREADS', '16')), os.cpu_count() or 1)))
        _LIB = L
    return _LIB


def _f(a):
    a = np.ascontiguousarray(a, dtype=np.float32)
    return a, a.ctypes.data_as(C.POINTER(C.c_float))


# ---- stage-level functions -------------------------------------------------------------
def hann_periodic(n):
    out = np.empty(n, np.float32)
    lib().ora_hann_periodic(n, out.ctypes.data_as(C.POINTER(C.c_float)))
    return out


def pad_reflect(x, pad):
    x, xp = _f(x)
    out = np.empty(len(x) + 2 * pad, np.float32)
    lib().ora_pad_reflect(xp, len(x), pad, out.ctypes.data_as(C.POINTER(C.c_float)))
    return out


def stft(sig, fft_size, hop, window, center=True):
    sig, sp = _f(sig)
    window, wp = _f(window)
    T = 1 + len(sig) // hop
    out = np.empty((fft_size // 2 + 1, T), np.float32)
    lib().ora_stft(sp, len(sig), fft_size, hop, wp, int(center), out.ctypes.data_as(C.POINTER(C.c_float)))
    return out


def mel_filterbank(sr=16000.0, n_fft=1024, n_mels=128, fmin=30.0, fmax=8000.0):
    out = np.empty((n_mels, n_fft // 2 + 1), np.float32)
    lib().ora_mel_filterbank(sr, n_fft, n_mels, fmin, fmax, out.ctypes.data_as(C.POINTER(C.c_float)))
    return out


def mel_extract(sig):
    sig, sp = _f(sig)
    T = 1 + len(sig) // 160
    out = np.empty((128, T), np.float32)
    lib().ora_mel_extract(sp, len(sig), out.ctypes.data_as(C.POINTER(C.c_float)))
    return out


def decode(salience, threshold=0.03):
    s, sp = _f(salience)
    T = s.shape[0]
    f0 = np.empty(T, np.float32)
    rc = lib().ora_decode(sp, T, threshold, f0.ctypes.data_as(C.POINTER(C.c_float)))
    return rc, f0


def get_f0_post(f0):
    f0, fp = _f(f0)
    out = np.empty(len(f0), np.int32)
    lib().ora_get_f0_post(fp, len(f0), out.ctypes.data_as(C.POINTER(C.c_int32)))
    return out, f0


def uppower(shift):
    return float(lib().ora_uppower(int(shift)))


def f0_extractor_frame(sf):
    return int(lib().ora_f0_extractor_frame(int(sf)))


def knn_search(index, q, k=4):
    index, ip = _f(index)
    q, qp = _f(q)
    nq = q.shape[0]
    idx = np.empty((nq, k), np.int32)
    dist = np.empty((nq, k), np.float32)
    lib().ora_knn_search(ip, index.shape[0], index.shape[1], qp, nq, k, idx.ctypes.data_as(C.POINTER(C.c_int32)),
                         dist.ctypes.data_as(C.POINTER(C.c_float)))
    return idx, dist


def philox_normal(seed, stream, chunk, purpose, n):
    out = np.empty(n, np.float32)
    lib().ora_philox_normal(seed, stream, chunk, purpose, n, out.ctypes.data_as(C.POINTER(C.c_float)))
    return out


def rms(y, frame_length, hop_length):
    y, yp = _f(y)
    out = np.empty(len(y) // hop_length + 8, np.float32)
    n = lib().ora_rms(yp, len(y), frame_length, hop_length, out.ctypes.data_as(C.POINTER(C.c_float)))
    return out[:n].copy()


def lerp_align_corners(x, size):
    x, xp = _f(x)
    out = np.empty(size, np.float32)
    lib().ora_lerp_align_corners(xp, len(x), size, out.ctypes.data_as(C.POINTER(C.c_float)))
    return out


def envelop_mixing(inp, out, sample_rate, mix_rate):
    inp, ip = _f(inp)
    o = np.array(out, dtype=np.float32, copy=True)
    lib().ora_envelop_mixing(ip, o.ctypes.data_as(C.POINTER(C.c_float)), len(o), sample_rate, float(mix_rate))
    return o


def sola_offset(input_buffer, sola_buffer, buffer_frame_size, search_frame_size):
    a, ap = _f(input_buffer)
    b, bp = _f(sola_buffer)
    return int(lib().ora_sola_offset(ap, bp, buffer_frame_size, search_frame_size))


def sola_step(output, sola_buffer, search, frame):
    o = np.array(output, dtype=np.float32, copy=True)
    sb = np.array(sola_buffer, dtype=np.float32, copy=True)
    fr = np.empty(frame, np.float32)
    fp_ = C.POINTER(C.c_float)
    off = lib().ora_sola_step(o.ctypes.data_as(fp_), len(o), sb.ctypes.data_as(fp_), len(sb), search, frame, fr.ctypes.data_as(fp_))
    return int(off), fr, sb


def set_threads(n):
    lib().ora_set_threads(int(n))


# ---- RvcInfer mirror -------------------------------------------------------------------
class OracleRvcInfer:
    """Same method names / argument meaning as rvc::RvcInfer (rvc/src/rvc.rs:30-220)."""

    def __init__(self, data_path: str):
        self._h = lib().ora_new(str(data_path).encode())

    def close(self):
        if self._h:
            lib().ora_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _chk(self, rc):
        if rc != 0:
            raise OracleError(rc, (lib().ora_last_error(self._h) or b"").decode())

    def load_contentvec(self, version=2):
        self._chk(lib().ora_load_contentvec(self._h, int(version)))

    def load_model(self, path):
        self._chk(lib().ora_load_model(self._h, str(path).encode()))

    def load_f0(self, algorithm=1):
        self._chk(lib().ora_load_f0(self._h, int(algorithm)))

    def unload_model(self):
        lib().ora_unload_model(self._h)

    def load_index(self, vecs):
        v, vp = _f(vecs)
        self._chk(lib().ora_load_index(self._h, vp, v.shape[0], v.shape[1]))

    def set_index_rate(self, r):
        lib().ora_set_index_rate(self._h, float(r))

    def set_noise_seed(self, seed, stream=0):
        lib().ora_set_noise_seed(self._h, int(seed), int(stream))

    def reset_state(self):
        lib().ora_reset_state(self._h)

    def enable_taps(self, on=True):
        lib().ora_enable_taps(self._h, int(on))

    def tap(self, name):
        p = C.POINTER(C.c_float)()
        n = C.c_size_t()
        rc = lib().ora_get_tap(self._h, name.encode(), C.byref(p), C.byref(n))
        if rc != 0:
            raise KeyError(name)
        return np.ctypeslib.as_array(p, shape=(n.value,)).copy()

    def pitch_cache(self):
        out = np.empty(1024, np.float32)
        lib().ora_get_pitch_cache(self._h, out.ctypes.data_as(C.POINTER(C.c_float)))
        return out

    def knn(self, rows_cap=4096):
        idx = np.empty((rows_cap, 4), np.int32)
        dist = np.empty((rows_cap, 4), np.float32)
        rows = C.c_size_t()
        self._chk(lib().ora_get_knn(self._h, idx.ctypes.data_as(C.POINTER(C.c_int32)), dist.ctypes.data_as(C.POINTER(C.c_float)),
                                    rows_cap, C.byref(rows)))
        return idx[:rows.value].copy(), dist[:rows.value].copy()

    # caller-side post-processing with the engine's method names (lets tests drive obs_rvc_amd.streaming with the oracle)
    def envelop_mixing(self, inp, out, sample_rate, mix_rate):
        return envelop_mixing(np.asarray(inp)[:len(out)], out, sample_rate, mix_rate)

    def sola_step(self, output, sola_buffer, search, frame):
        return sola_step(output, sola_buffer, search, frame)

    def hubert(self, x):
        x, xp = _f(x)
        cap = 1024 * (len(x) // 320 + 8)
        out = np.empty(cap, np.float32)
        dims = (C.c_size_t * 3)()
        self._chk(lib().ora_hubert(self._h, xp, len(x), out.ctypes.data_as(C.POINTER(C.c_float)), cap, dims))
        return out[:dims[0] * dims[1] * dims[2]].reshape(dims[0], dims[1], dims[2]).copy()

    def extract_feature(self, x):
        x, xp = _f(x)
        cap = 1024 * (2 * (len(x) // 320) + 16)
        out = np.empty(cap, np.float32)
        dims = (C.c_size_t * 3)()
        self._chk(lib().ora_extract_feature(self._h, xp, len(x), out.ctypes.data_as(C.POINTER(C.c_float)), cap, dims))
        return out[:dims[0] * dims[1] * dims[2]].reshape(dims[0], dims[1], dims[2]).copy()

    def pitch(self, x, pitch_shift, sample_frame_16k_size):
        x, xp = _f(x)
        out = np.empty(4096, np.float32)
        n = C.c_size_t()
        self._chk(lib().ora_pitch(self._h, xp, len(x), int(pitch_shift), int(sample_frame_16k_size),
                                  out.ctypes.data_as(C.POINTER(C.c_float)), 4096, C.byref(n)))
        return out[:n.value].copy()

    def infer(self, x, sample_frame_16k_size, pitch_shift, skip_head, return_length):
        x, xp = _f(x)
        cap = int(return_length) * 1024 + 16
        out = np.empty(cap, np.float32)
        n = C.c_size_t()
        has = 0 if pitch_shift is None else 1
        self._chk(lib().ora_infer(self._h, xp, len(x), int(sample_frame_16k_size), has, int(pitch_shift or 0), int(skip_head),
                                  int(return_length), out.ctypes.data_as(C.POINTER(C.c_float)), cap, C.byref(n)))
        return out[:n.value].copy()
