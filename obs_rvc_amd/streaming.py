"""Caller-side streaming state machine: the plugin's `process_one_frame`
(/root/reference/obs-rvc/src/lib.rs:659-795) around `RvcInfer::infer`, minus the two rubato resamplers (row f3,
not built yet): the caller supplies each chunk at BOTH the host rate (for the RMS envelope) and 16 kHz (for the
engine), and the model rate must equal the host rate.  Backend-agnostic: `engine` only needs
infer / envelop_mixing / sola_step with the signatures of obs_rvc_amd.rvc.RvcInfer."""
from __future__ import annotations

import numpy as np

from .geometry import Geometry


class StreamingSession:
    def __init__(self, engine, geom: Geometry, pitch_shift: int = 12, rms_mix_rate: float = 1.0):
        self.e, self.g = engine, geom
        self.pitch_shift, self.rms_mix_rate = pitch_shift, rms_mix_rate
        if geom.model_return_size != geom.model_return_length * (geom.sample_rate // 100):
            raise ValueError("model rate must equal the host rate until the resamplers (row f3) exist")
        self.input_buffer = np.zeros(geom.input_buffer_size, np.float32)            # lib.rs:215
        self.input_buffer_16k = np.zeros(geom.input_buffer_16k_size, np.float32)    # lib.rs:218
        self.sola_buffer = np.zeros(geom.sola_buffer_frame_size, np.float32)        # lib.rs:229
        self.last_sola_offset = 0

    def process_one_frame(self, chunk_host_rate: np.ndarray, chunk_16k: np.ndarray) -> np.ndarray:
        g = self.g
        assert len(chunk_host_rate) == g.sample_frame_size and len(chunk_16k) == g.sample_frame_16k
        # lib.rs:661-665: move and append the last n samples
        self.input_buffer[:-g.sample_frame_size] = self.input_buffer[g.sample_frame_size:]
        self.input_buffer[-g.sample_frame_size:] = chunk_host_rate
        # lib.rs:669-683 (resampler output replaced by the caller's 16 kHz chunk)
        self.input_buffer_16k[:-g.sample_frame_16k] = self.input_buffer_16k[g.sample_frame_16k:]
        self.input_buffer_16k[-g.sample_frame_16k:] = chunk_16k
        # lib.rs:694-707
        out = self.e.infer(self.input_buffer_16k, g.sample_frame_16k, self.pitch_shift, g.skip_head, g.model_return_length)
        # lib.rs:758-765
        if self.rms_mix_rate < 1.0:
            out = self.e.envelop_mixing(self.input_buffer[g.extra_frame_size:], out, g.sample_rate, self.rms_mix_rate)
        # lib.rs:768-794
        off, frame, self.sola_buffer = self.e.sola_step(out, self.sola_buffer, g.sola_search_frame_size, g.sample_frame_size)
        self.last_sola_offset = off
        return frame
