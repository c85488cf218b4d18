"""Caller-side streaming state machine (SURVEY.md section 8 rows f1 + f3): the plugin's `process_one_frame`
(/root/reference/obs-rvc/src/lib.rs:659-795) around `RvcInfer::infer`, including its two rubato resamplers
(lib.rs:236-242): host rate -> 16 kHz in front of the engine and model rate -> host rate behind it.

Backend-agnostic: `engine` needs infer / envelop_mixing / sola_step with the signatures of obs_rvc_amd.rvc.RvcInfer and
`resampler` is a factory `(rate_in, rate_out, chunk_size_in) -> object with process()` -- for the HIP engine
`lambda ri, ro, n: obs_rvc_amd.resample.FftFixedInOut(engine, ri, ro, n)`.  With `resampler=None` both converters are
bypassed: the caller then supplies every chunk at the host rate AND at 16 kHz and the model rate must equal the host rate."""
from __future__ import annotations

import numpy as np

from .geometry import Geometry


class StreamingSession:
    def __init__(self, engine, geom: Geometry, pitch_shift: int = 12, rms_mix_rate: float = 1.0,
                 model_output_sample_rate: int | None = None, resampler=None, skip_inference: bool = False):
        self.e, self.g = engine, geom
        self.pitch_shift, self.rms_mix_rate, self.skip_inference = pitch_shift, rms_mix_rate, skip_inference
        g = geom
        host_hop = g.sample_rate // 100
        self.model_return_size = g.model_return_size
        if model_output_sample_rate is None:
            model_output_sample_rate = g.model_return_size // g.model_return_length * 100
        if skip_inference:                                                          # lib.rs:224-227
            model_output_sample_rate, self.model_return_size = 16000, g.model_return_length * 160
        self.downsampler = self.upsampler = None
        if resampler is not None:
            self.downsampler = resampler(g.sample_rate, 16000, g.sample_frame_size + 2 * g.zc)       # lib.rs:236-237
            self.upsampler = resampler(model_output_sample_rate, g.sample_rate, self.model_return_size)   # lib.rs:240-242
        elif self.model_return_size != g.model_return_length * host_hop:
            raise ValueError("without resamplers the model rate must equal the host rate")
        self.input_buffer = np.zeros(g.input_buffer_size, np.float32)            # lib.rs:215
        self.input_buffer_16k = np.zeros(g.input_buffer_16k_size, np.float32)    # lib.rs:218
        self.sola_buffer = np.zeros(g.sola_buffer_frame_size, np.float32)        # lib.rs:229
        self.last_sola_offset = 0

    def process_one_frame(self, input_sample: np.ndarray, chunk_16k: np.ndarray | None = None) -> np.ndarray:
        g = self.g
        assert len(input_sample) == g.sample_frame_size
        # lib.rs:661-665: move and append the last n samples
        self.input_buffer[:-g.sample_frame_size] = self.input_buffer[g.sample_frame_size:]
        self.input_buffer[-g.sample_frame_size:] = input_sample
        # lib.rs:669-683: resample and set to 16k.  The converter is fed the new chunk plus the 2*zc samples before it and its
        # first 160 output samples are dropped: the write covers the new 16 kHz chunk and re-writes the 160 samples before it.
        self.input_buffer_16k[:-g.sample_frame_16k] = self.input_buffer_16k[g.sample_frame_16k:]
        if self.downsampler is not None:
            start = len(self.input_buffer) - g.sample_frame_size - 2 * g.sample_rate // 100
            result = self.downsampler.process(self.input_buffer[start:])
            copy_begin = len(self.input_buffer_16k) - (g.sample_frame_size // (g.sample_rate // 100) + 1) * 160
            self.input_buffer_16k[copy_begin:] = result[160:]
        else:
            assert chunk_16k is not None and len(chunk_16k) == g.sample_frame_16k
            self.input_buffer_16k[-g.sample_frame_16k:] = chunk_16k
        # lib.rs:694-707
        if self.skip_inference:
            out = self.input_buffer_16k[len(self.input_buffer_16k) - self.model_return_size:].copy()
        else:
            out = self.e.infer(self.input_buffer_16k, g.sample_frame_16k, self.pitch_shift, g.skip_head, g.model_return_length)
        # lib.rs:742-756: model rate -> host rate
        if self.upsampler is not None:
            out = self.upsampler.process(out)
        # lib.rs:758-765
        if self.rms_mix_rate < 1.0:
            out = self.e.envelop_mixing(self.input_buffer[g.extra_frame_size:], out, g.sample_rate, self.rms_mix_rate)
        # lib.rs:768-794
        off, frame, self.sola_buffer = self.e.sola_step(out, self.sola_buffer, g.sola_search_frame_size, g.sample_frame_size)
        self.last_sola_offset = off
        return frame


class NativeStreamingSession:
    """The same state machine as one native call per chunk with every buffer resident in HBM (`rvc_session_*`,
    csrc/session.hip.h): one H2D copy, one D2H copy and one synchronisation per chunk."""

    def __init__(self, engine, sample_rate: int = 48000, sample_length: float = 0.30, crossfade_length: float = 0.07,
                 extra_inference_time: float = 2.0, model_output_sample_rate: int = 40000, pitch_shift: int = 12,
                 rms_mix_rate: float = 1.0, skip_inference: bool = False):
        import ctypes as C
        from . import _native
        from .rvc_common import RvcInferError
        self._C, self._err, self._L, self._engine = C, RvcInferError, _native.lib(), engine
        h = C.c_void_p()
        rc = self._L.rvc_session_create(engine._h, sample_rate, sample_length, crossfade_length, extra_inference_time, model_output_sample_rate,
                                        pitch_shift, rms_mix_rate, 1 if skip_inference else 0, C.byref(h))
        if rc != 0:
            raise RvcInferError(rc, (self._L.rvc_last_error_message(engine._h) or b"").decode())
        self._h = h
        self.n_streams = int(getattr(engine, "n_streams", 1))
        g = (C.c_int32 * 10)()
        self._L.rvc_session_geometry(h, g)
        (self.sample_frame_size, self.sample_frame_16k, self.input_buffer_size, self.input_buffer_16k_size, self.model_return_length,
         self.model_return_size, self.skip_head, self.sola_buffer_frame_size, self.sola_search_frame_size, self.extra_frame_size) = [int(v) for v in g]
        self.last_sola_offset = 0

    def __del__(self):
        h, self._h = getattr(self, "_h", None), None
        if h and getattr(self._engine, "_h", None):
            self._L.rvc_session_destroy(h)

    def set_params(self, pitch_shift: int, rms_mix_rate: float, stream: int = None) -> None:
        """Settings of every stream, or (stream given) of one stream only (rvc_session_set_params_stream)."""
        if stream is None:
            self._L.rvc_session_set_params(self._h, pitch_shift, rms_mix_rate)
        elif int(self._L.rvc_session_set_params_stream(self._h, int(stream), int(pitch_shift), float(rms_mix_rate))) != 0:
            raise ValueError("stream out of range")

    def process_one_frame(self, input_sample: np.ndarray) -> np.ndarray:
        """One chunk of one stream (shape (sample_frame_size,)) or of every stream of the engine ((streams, sample_frame_size))."""
        C = self._C
        x = np.ascontiguousarray(input_sample, dtype=np.float32)
        single = x.ndim == 1
        x = x.reshape(1, -1) if single else x
        if x.shape != (self.n_streams, self.sample_frame_size):
            raise self._err(5, "expected %d stream(s) of %d samples" % (self.n_streams, self.sample_frame_size))
        out = np.empty((x.shape[0], self.sample_frame_size), np.float32)
        off = (C.c_size_t * x.shape[0])()
        fp = C.POINTER(C.c_float)
        rc = self._L.rvc_session_process(self._h, x.ctypes.data_as(fp), x.shape[1], out.ctypes.data_as(fp), out.shape[1], off)
        if rc != 0:
            raise self._err(rc, (self._L.rvc_last_error_message(self._engine._h) or b"").decode())
        self.last_sola_offsets = [int(v) for v in off]
        self.last_sola_offset = self.last_sola_offsets[0]
        return out[0] if single else out
