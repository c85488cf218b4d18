"""Host-side mirror of the resampler the plugin uses on both sides of `RvcInfer::infer`:
`rubato::FftFixedInOut::<f32>::new(rate_in, rate_out, chunk_size_in, 1)` at /root/reference/obs-rvc/src/lib.rs:236-242,
`process` at lib.rs:675 and `process_into_buffer` at lib.rs:747-749.  Same method names and error behaviour (a wrong input
length raises, where rubato returns ResampleError::WrongNumberOfInputFrames); the arithmetic runs in the HIP library
(csrc/resample.hip.h) on the engine's device -- there is no CPU fallback."""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _native
from .rvc_common import RvcInferError


class FftFixedInOut:
    def __init__(self, engine, sample_rate_input: int, sample_rate_output: int, chunk_size_in: int, nbr_channels: int = 1):
        if nbr_channels != 1:
            raise ValueError("the plugin resamples one channel (lib.rs:237,241)")
        self._L = _native.lib()
        self._engine = engine                      # keeps the engine (device, stream) alive
        h = C.c_void_p()
        rc = self._L.rvc_resampler_create(engine._h, sample_rate_input, sample_rate_output, chunk_size_in, C.byref(h))
        if rc != 0:
            raise RvcInferError(rc, (self._L.rvc_last_error_message(engine._h) or b"").decode())
        self._h = h

    def __del__(self):
        h, self._h = getattr(self, "_h", None), None
        if h and getattr(self._engine, "_h", None):
            self._L.rvc_resampler_destroy(h)

    def input_frames_next(self) -> int:
        return int(self._L.rvc_resampler_input_frames_next(self._h))

    def output_frames_max(self) -> int:
        return int(self._L.rvc_resampler_output_frames_max(self._h))

    def reset(self) -> None:
        self._L.rvc_resampler_reset(self._h)

    def process(self, wave_in) -> np.ndarray:
        x = np.ascontiguousarray(wave_in, dtype=np.float32).reshape(-1)
        out = np.empty(self.output_frames_max(), np.float32)
        n = C.c_size_t(0)
        fp = C.POINTER(C.c_float)
        rc = self._L.rvc_resampler_process(self._h, x.ctypes.data_as(fp), x.size, out.ctypes.data_as(fp), out.size, C.byref(n))
        if rc != 0:
            raise RvcInferError(rc, (self._L.rvc_last_error_message(self._engine._h) or b"").decode())
        return out[: n.value]

    def process_into_buffer(self, wave_in, wave_out: np.ndarray):
        """-> (frames consumed, frames written), as rubato's process_into_buffer (lib.rs:747-758)."""
        y = self.process(wave_in)
        if wave_out.size < y.size:
            raise RvcInferError(5, "output buffer too small")
        wave_out[: y.size] = y
        return len(np.asarray(wave_in).reshape(-1)), y.size

    def process_device(self, d_in: int, d_out: int, sync: bool = False) -> None:
        rc = self._L.rvc_resampler_process_device(self._h, C.c_void_p(d_in), C.c_void_p(d_out), 1 if sync else 0)
        if rc != 0:
            raise RvcInferError(rc, (self._L.rvc_last_error_message(self._engine._h) or b"").decode())
