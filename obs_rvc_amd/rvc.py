"""Python mirror of the reference's `rvc` crate API (rvc/src/lib.rs:5, rvc/src/rvc.rs:18-220) on top of
the C ABI (include/rvc_mi355x.h).  Same method names, argument meaning and error behaviour as
`rvc::RvcInfer`; numpy arrays stand in for ndarray views.  All compute happens in the HIP library."""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

from . import _native
from .rvc_common import PitchAlgorithm, RvcInferError, RvcModelVersion

_FP = C.POINTER(C.c_float)


def _f32(a):
    a = np.ascontiguousarray(a, dtype=np.float32)
    return a, a.ctypes.data_as(_FP)


class RvcInfer:
    def __init__(self, data_path, device: int = -1):
        """RvcInfer::new (rvc.rs:30-44)."""
        self._L = _native.lib()
        h = C.c_void_p()
        rc = self._L.rvc_create(os.fspath(data_path).encode(), int(device), C.byref(h))
        if rc != 0:
            raise RvcInferError(rc, "rvc_create failed (no HIP device?)")
        self._h = h
        self.n_streams = 1

    # -- lifetime -----------------------------------------------------------------------
    def close(self):
        if getattr(self, "_h", None):
            self._L.rvc_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _chk(self, rc):
        if rc != 0:
            raise RvcInferError(rc, (self._L.rvc_last_error_message(self._h) or b"").decode())

    # -- loading (rvc.rs:46-79) -------------------------------------------------------------
    def load_contentvec(self, model_version=RvcModelVersion.V2):
        self._chk(self._L.rvc_load_contentvec(self._h, int(RvcModelVersion.from_value(model_version))))

    def load_model(self, model_path):
        self._chk(self._L.rvc_load_model(self._h, os.fspath(model_path).encode()))

    def load_f0(self, pitch_algorithm=PitchAlgorithm.Rmvpe):
        self._chk(self._L.rvc_load_f0(self._h, int(PitchAlgorithm.from_value(pitch_algorithm))))

    def unload_model(self):
        self._L.rvc_unload_model(self._h)

    # -- per-chunk API (rvc.rs:81-220) ----------------------------------------------------
    def hubert(self, input):
        """-> (1, C, T) float32 (rvc.rs:81-97)."""
        x, xp = _f32(input)
        dims = (C.c_size_t * 3)()
        cap = 1024 * (len(x) // 320 + 8)
        out = np.empty(cap, np.float32)
        self._chk(self._L.rvc_hubert(self._h, xp, len(x), out.ctypes.data_as(_FP), cap, dims))
        return out[: dims[0] * dims[1] * dims[2]].reshape(dims[0], dims[1], dims[2]).copy()

    def extract_feature(self, input):
        """-> (1, 2T+1, C) float32 (rvc.rs:99-109)."""
        x, xp = _f32(input)
        dims = (C.c_size_t * 3)()
        cap = 1024 * (2 * (len(x) // 320) + 16)
        out = np.empty(cap, np.float32)
        self._chk(self._L.rvc_extract_feature(self._h, xp, len(x), out.ctypes.data_as(_FP), cap, dims))
        return out[: dims[0] * dims[1] * dims[2]].reshape(dims[0], dims[1], dims[2]).copy()

    def pitch(self, input, pitch_shift: int, sample_frame_16k_size: int):
        """-> f0 in Hz, one value per RMVPE frame (rvc.rs:111-131)."""
        x, xp = _f32(input)
        n = C.c_size_t()
        out = np.empty(4096, np.float32)
        self._chk(self._L.rvc_pitch(self._h, xp, len(x), int(pitch_shift), int(sample_frame_16k_size), out.ctypes.data_as(_FP), 4096, C.byref(n)))
        return out[: n.value].copy()

    def infer(self, input, sample_frame_16k_size: int, pitch_shift, skip_head: int, return_length: int):
        """-> float PCM at the model rate (rvc.rs:133-220).  pitch_shift=None mirrors Option::None."""
        x, xp = _f32(input)
        n = C.c_size_t()
        cap = int(return_length) * 1024 + 16
        out = np.empty(cap, np.float32)
        self._chk(self._L.rvc_infer(self._h, xp, len(x), int(sample_frame_16k_size), 0 if pitch_shift is None else 1, int(pitch_shift or 0),
                                    int(skip_head), int(return_length), out.ctypes.data_as(_FP), cap, C.byref(n)))
        return out[: n.value].copy()

    # -- extensions ---------------------------------------------------------------------
    def load_index(self, vectors):
        """`vectors`: an (n, dim) float32 array, or the path of a Faiss `.index` file (IndexFlat / IndexIVFFlat: the stored
        vectors are reconstructed in id order, obs_rvc_amd.faiss_index) or of a `.npy` matrix (upstream's total_fea.npy)."""
        if isinstance(vectors, (str, os.PathLike)):
            path = os.fspath(vectors)
            if path.endswith(".npy"):
                vectors = np.load(path)
            else:
                from .faiss_index import read_index
                vectors = read_index(path)
        v, vp = _f32(vectors)
        self._chk(self._L.rvc_load_index(self._h, vp, v.shape[0], v.shape[1]))

    def rccl_unique_id(self) -> bytes:
        """rvc_rccl_unique_id: rank 0 creates the 128-byte ncclUniqueId the host then hands to the other ranks."""
        buf = C.create_string_buffer(128)
        self._chk(self._L.rvc_rccl_unique_id(buf))
        return buf.raw

    def index_broadcast(self, unique_id: bytes, rank: int, world: int, vectors=None, expect=None):
        """rvc_index_broadcast: ONE ncclBroadcast of the shared retrieval index from rank 0 into this rank's HBM (RCCL over xGMI).
        Rank 0 passes the (n, dim) matrix (or None to send the index it already holds); the other ranks pass None, optionally with the
        shape they expect (`expect=(n, dim)`: a mismatch with what rank 0 sends makes EVERY rank fail, together)."""
        assert len(unique_id) == 128
        if vectors is not None:
            v, vp = _f32(vectors)
            self._chk(self._L.rvc_index_broadcast(self._h, unique_id, int(rank), int(world), vp, v.shape[0], v.shape[1]))
        else:
            n, dim = expect if expect else (0, 0)
            self._chk(self._L.rvc_index_broadcast(self._h, unique_id, int(rank), int(world), None, int(n), int(dim)))

    def rccl_available(self) -> bool:
        """rvc_rccl_available: can this process load librccl?  (no communicator is created)"""
        return int(self._L.rvc_rccl_available()) == 0

    def index_broadcast_info(self):
        """rvc_index_broadcast_info -> {ms_comm_init, ms_broadcast, ms_repack, ranks} of this engine's last broadcast"""
        ms = (C.c_double * 3)()
        ranks = C.c_int(0)
        self._chk(self._L.rvc_index_broadcast_info(self._h, ms, C.byref(ranks)))
        return {"ms_comm_init": round(ms[0], 3), "ms_broadcast": round(ms[1], 3), "ms_repack": round(ms[2], 3), "ranks": int(ranks.value)}

    def set_index_rate(self, rate: float):
        self._L.rvc_set_index_rate(self._h, float(rate))

    def knn(self, rows_cap: int = 4096):
        """hits of the last infer: (rows, 4) indices and squared distances; rows = return_length per stream, stream-major, for as many streams
        of a batched call as rows_cap holds whole"""
        idx = np.empty((rows_cap, 4), np.int32)
        dist = np.empty((rows_cap, 4), np.float32)
        rows = C.c_size_t()
        self._chk(self._L.rvc_get_knn(self._h, idx.ctypes.data_as(C.POINTER(C.c_int32)), dist.ctypes.data_as(_FP), rows_cap, C.byref(rows)))
        return idx[: rows.value].copy(), dist[: rows.value].copy()

    def set_noise_seed(self, seed: int, stream_id: int = 0):
        self._L.rvc_set_noise_seed(self._h, int(seed), int(stream_id))

    def reset_state(self):
        self._L.rvc_reset_state(self._h)

    def set_streams(self, n: int):
        self._chk(self._L.rvc_set_streams(self._h, int(n)))
        self.n_streams = int(n)

    def infer_batch(self, inputs, sample_frame_16k_size: int, pitch_shift, skip_head: int, return_length: int):
        """inputs (n_streams, n) -> (n_streams, N).  pitch_shift: one int for every stream, or a sequence of n_streams ints (every
        stream of a batch is a caller of its own: rvc_infer_batch_v)."""
        x, xp = _f32(inputs)
        assert x.ndim == 2 and x.shape[0] == self.n_streams
        n = C.c_size_t()
        cap = int(return_length) * 1024 + 16
        out = np.empty((self.n_streams, cap), np.float32)
        if np.ndim(pitch_shift) == 0:
            self._chk(self._L.rvc_infer_batch(self._h, xp, x.shape[1], int(sample_frame_16k_size), int(pitch_shift), int(skip_head),
                                              int(return_length), out.ctypes.data_as(_FP), cap, C.byref(n)))
        else:
            sh = np.ascontiguousarray(pitch_shift, dtype=np.int32)
            assert sh.shape == (self.n_streams,)
            self._chk(self._L.rvc_infer_batch_v(self._h, xp, x.shape[1], int(sample_frame_16k_size), sh.ctypes.data_as(C.POINTER(C.c_int32)), int(skip_head),
                                                int(return_length), out.ctypes.data_as(_FP), cap, C.byref(n)))
        return out[:, : n.value].copy()

    def infer_batch_g(self, inputs, sample_frame_16k_size, pitch_shift, skip_head, return_length):
        """rvc_infer_batch_g: one call for n_streams callers that do NOT share a geometry (each argument is a sequence with one entry per
        stream; inputs[s] is that stream's 16 kHz buffer).  -> list of n_streams output arrays"""
        S = self.n_streams
        assert len(inputs) == len(sample_frame_16k_size) == len(skip_head) == len(return_length) == S
        xs = [np.ascontiguousarray(x, dtype=np.float32) for x in inputs]
        caps = [int(r) * 1024 + 16 for r in return_length]
        outs = [np.empty(c, np.float32) for c in caps]
        in_p = (_FP * S)(*[x.ctypes.data_as(_FP) for x in xs])
        out_p = (_FP * S)(*[o.ctypes.data_as(_FP) for o in outs])
        sz = C.c_size_t
        n_a = (sz * S)(*[x.shape[0] for x in xs]); f_a = (sz * S)(*[int(v) for v in sample_frame_16k_size]); cap_a = (sz * S)(*caps); len_a = (sz * S)()
        sh_a = (C.c_uint32 * S)(*[int(v) for v in skip_head]); rl_a = (C.c_uint32 * S)(*[int(v) for v in return_length])
        ps_a = None if pitch_shift is None else (C.c_int32 * S)(*[int(v) for v in pitch_shift])
        self._chk(self._L.rvc_infer_batch_g(self._h, in_p, n_a, f_a, ps_a, sh_a, rl_a, out_p, cap_a, len_a))
        return [o[: len_a[i]].copy() for i, o in enumerate(outs)]

    def infer_device(self, d_in_ptr: int, n: int, sample_frame_16k_size: int, pitch_shift: int, skip_head: int, return_length: int,
                     d_out_ptr: int, cap_per_stream: int, sync: bool = False) -> int:
        nn = C.c_size_t()
        self._chk(self._L.rvc_infer_device(self._h, C.c_void_p(d_in_ptr), int(n), int(sample_frame_16k_size), int(pitch_shift), int(skip_head),
                                           int(return_length), C.c_void_p(d_out_ptr), int(cap_per_stream), C.byref(nn), 1 if sync else 0))
        return nn.value

    def set_pipeline(self, on: bool = True):
        """Offline throughput mode: unsynchronised infer_device calls overlap across chunks (rvc_set_pipeline)."""
        self._L.rvc_set_pipeline(self._h, 1 if on else 0)

    def set_gemm_precision(self, mode: int):
        """EXPLORATORY: 1 = the wide 1-D layers as three bf16 matrix-core products per fp32 product at many streams; 0 = fp32 (default)."""
        self._chk(self._L.rvc_set_gemm_precision(self._h, int(mode)))

    def set_plan_cache(self, n_plans: int):
        """Plans (one per call geometry) the engine keeps; least recently used evicted first (rvc_set_plan_cache)."""
        self._chk(self._L.rvc_set_plan_cache(self._h, int(n_plans)))

    def set_plan_autotune(self, on: bool = True):
        """plans of more than 4 streams pick among the eligible kernels / tiles of every layer by timing them at plan build (default on); False: the rules only"""
        self._chk(self._L.rvc_set_plan_autotune(self._h, 1 if on else 0))

    def plan_autotune_info(self) -> dict:
        """the last plan build: layers tuned by trials / changed against the rules / served from the process cache, ms in trials, ms in all"""
        t, c, h = C.c_int(), C.c_int(), C.c_int()
        tm, bm = C.c_double(), C.c_double()
        self._chk(self._L.rvc_plan_autotune_info(self._h, C.byref(t), C.byref(c), C.byref(h), C.byref(tm), C.byref(bm)))
        return {"tuned": t.value, "changed": c.value, "cache_hits": h.value, "tune_ms": tm.value, "build_ms": bm.value}

    def plan_cache_info(self) -> dict:
        cap, cached, builds = C.c_int(), C.c_int(), C.c_longlong()
        self._L.rvc_plan_cache_info(self._h, C.byref(cap), C.byref(cached), C.byref(builds))
        return {"capacity": cap.value, "cached": cached.value, "builds": builds.value}

    def retrieval_recoveries(self) -> int:
        """Chunks whose retrieval was recomputed through the exhaustive scan after a hand-off time-out (they returned normally)."""
        return int(self._L.rvc_retrieval_recoveries(self._h))

    def synchronize(self):
        self._chk(self._L.rvc_synchronize(self._h))

    def set_use_graph(self, on: bool = True):
        self._L.rvc_set_use_graph(self._h, 1 if on else 0)

    def set_profile(self, on: bool = True):
        self._L.rvc_set_profile(self._h, 1 if on else 0)

    def last_gpu_ms(self) -> float:
        return float(self._L.rvc_last_gpu_ms(self._h))

    def profile_last(self):
        n, ms, fl = C.c_int(), C.c_double(), C.c_double()
        self._chk(self._L.rvc_profile_last(self._h, C.byref(n), C.byref(ms), C.byref(fl)))
        return n.value, ms.value, fl.value

    def profile_last_knn(self):
        n, ms, by = C.c_int(), C.c_double(), C.c_double()
        self._chk(self._L.rvc_profile_last_knn(self._h, C.byref(n), C.byref(ms), C.byref(by)))
        return n.value, ms.value, by.value

    def enable_taps(self, on=True):
        """True / 1: taps on the explicit plan; 2: taps on the production plan (folded LayerNorms, composed WaveNets); False: off"""
        self._L.rvc_enable_taps(self._h, 2 if on == 2 else (1 if on else 0))

    def plan_ops(self) -> int:
        """kernel launches / copies queued per chunk by the plan of the last call (test aid: which plan ran)"""
        n = C.c_int(0)
        self._L.rvc_debug_last_plan.argtypes = [C.c_void_p, C.POINTER(C.c_int)]
        self._L.rvc_debug_last_plan.restype = C.c_int
        assert self._L.rvc_debug_last_plan(self._h, C.byref(n)) == 1
        return int(n.value)

    def tap(self, name: str):
        n = C.c_size_t()
        cap = 1 << 24
        out = np.empty(cap, np.float32)
        self._chk(self._L.rvc_get_tap(self._h, name.encode(), out.ctypes.data_as(_FP), cap, C.byref(n)))
        return out[: n.value].copy()

    def pitch_cache(self, stream: int = 0):
        out = np.zeros(1024, np.float32)
        self._L.rvc_get_pitch_cache(self._h, int(stream), out.ctypes.data_as(_FP))
        return out

    # -- caller-side post-processing (obs-rvc/src/rt_utils.rs) ---------------------------------
    def envelop_mixing(self, input, output, sample_rate: int, mix_rate: float):
        """rt_utils.rs:119-132; returns the mixed copy of `output`."""
        x, xp = _f32(input)
        o = np.array(output, dtype=np.float32, copy=True)
        self._chk(self._L.rvc_envelop_mixing(self._h, xp, o.ctypes.data_as(_FP), len(o), int(sample_rate), float(mix_rate)))
        return o

    def sola_step(self, output, sola_buffer, search: int, frame: int):
        """rt_utils.rs:60-90 + lib.rs:768-794; returns (offset, frame samples, new sola buffer)."""
        o = np.array(output, dtype=np.float32, copy=True)
        sb = np.array(sola_buffer, dtype=np.float32, copy=True)
        fr = np.empty(frame, np.float32)
        off = C.c_size_t()
        self._chk(self._L.rvc_sola_step(self._h, o.ctypes.data_as(_FP), len(o), sb.ctypes.data_as(_FP), len(sb), int(search), int(frame),
                                        fr.ctypes.data_as(_FP), C.byref(off)))
        return off.value, fr, sb

    def index_device_ptr(self):
        b = C.c_size_t()
        p = self._L.rvc_index_device_ptr(self._h, C.byref(b))
        return p, b.value
