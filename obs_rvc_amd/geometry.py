"""Buffer geometry of the caller, restated from the plugin's `create` / `update`
(/root/reference/obs-rvc/src/lib.rs:200-227, 694): given the host sample rate and the three
length settings it yields the exact arguments the plugin passes to `RvcInfer::infer`."""
from __future__ import annotations

import math
from dataclasses import dataclass


def _round_half_away(x: float) -> int:
    """f64::round() of the reference (and llround() of the native session): halves away from zero -- Python's round() goes to even
    (sample_length 0.125 s at 48 kHz: 12.5 hops -> 13 in the plugin, 12 with round())."""
    return int(math.floor(x + 0.5)) if x >= 0 else -int(math.floor(-x + 0.5))


@dataclass(frozen=True)
class Geometry:
    sample_rate: int
    zc: int
    sample_frame_size: int
    sample_frame_16k: int
    crossfade_frame_size: int
    sola_buffer_frame_size: int
    sola_search_frame_size: int
    extra_frame_size: int
    input_buffer_size: int
    input_buffer_16k_size: int
    model_return_length: int
    model_return_size: int
    skip_head: int


def derive(sample_rate: int = 48000, sample_length: float = 0.30, crossfade_length: float = 0.07,
           extra_inference_time: float = 2.0, model_output_sample_rate: int = 40000) -> Geometry:
    zc = sample_rate // 100                                                              # lib.rs:200
    sample_frame_time = _round_half_away(sample_length * sample_rate / zc)                     # lib.rs:202
    sample_frame_size = sample_frame_time * zc
    sample_frame_16k = sample_frame_time * 160                                           # lib.rs:205
    crossfade_frame_size = _round_half_away(crossfade_length * sample_rate / zc) * zc          # lib.rs:206-207
    sola_buffer_frame_size = min(crossfade_frame_size, 4 * zc)                           # lib.rs:208
    sola_search_frame_size = zc                                                          # lib.rs:209
    extra_frame_size = _round_half_away(extra_inference_time * sample_rate / zc) * zc          # lib.rs:210-211
    input_buffer_size = extra_frame_size + crossfade_frame_size + sola_search_frame_size + sample_frame_size   # lib.rs:213-214
    input_buffer_16k_size = 160 * input_buffer_size // zc                                # lib.rs:217
    model_return_length = (sample_frame_size + sola_buffer_frame_size + sola_search_frame_size) // zc          # lib.rs:220-221
    model_return_size = model_return_length * (model_output_sample_rate // 100)          # lib.rs:222
    skip_head = extra_frame_size // (sample_rate // 100)                                 # lib.rs:694
    return Geometry(sample_rate, zc, sample_frame_size, sample_frame_16k, crossfade_frame_size, sola_buffer_frame_size,
                    sola_search_frame_size, extra_frame_size, input_buffer_size, input_buffer_16k_size, model_return_length,
                    model_return_size, skip_head)


# BASELINE.json configs: 160 ms chunks @16 kHz, v2-48k synthesizer
BASELINE_160MS = derive(48000, 0.16, 0.07, 2.0, 48000)
