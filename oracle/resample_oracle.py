"""CPU restatement of the plugin's two resamplers.  TEST INFRASTRUCTURE ONLY (see oracle/rvc_oracle.h).

The reference builds them at obs-rvc/src/lib.rs:236-242 as `rubato::FftFixedInOut::<f32>::new(rate_in, rate_out, chunk, 1)`
and calls `process` / `process_into_buffer` at lib.rs:675 and lib.rs:747-749.  rubato is a third-party dependency that is NOT
in /root/reference (Cargo.lock:1223 pins rubato 0.15.0); its published synchronous-FFT algorithm is restated here from the
crate's documented design (src/synchro.rs: FftResampler / FftFixedInOut; src/sinc.rs: make_sincs; src/windows.rs:
BlackmanHarris2).  PARITY UNPINNED: the reference holds no test or vector for the resamplers, so this
restatement is anchored on its own mathematical properties (tests/test_resample.py) and on the call sites above.

Algorithm of one `process` call (one channel):
  fft_size_in/out = chunk sizes rounded so that fft_in / fft_out = rate_in / rate_out exactly;
  filter  = windowed sinc (Blackman-Harris^2 window, fft_in taps, cutoff 0.4^(16/fft_in) [* fft_out/fft_in when decimating])
            scaled by 1/(2 fft_in), zero-padded to 2 fft_in, real-FFT'd once;
  chunk   -> zero-pad to 2 fft_in -> real FFT -> * filter spectrum -> keep the first new_len bins -> inverse real FFT of
            length 2 fft_out -> first half + saved overlap is the output, second half is the new overlap.
All arithmetic in f32 (scipy.fft keeps single precision).
"""
from math import gcd

import numpy as np
import scipy.fft as sfft


def blackman_harris(npoints: int) -> np.ndarray:
    """4-term Blackman-Harris, periodic form (x / npoints)."""
    x = np.arange(npoints, dtype=np.float32)
    n = np.float32(npoints)
    pi = np.float32(np.pi)
    a, b, c, d = np.float32(0.35875), np.float32(0.48829), np.float32(0.14128), np.float32(0.01168)
    return (a - b * np.cos(np.float32(2) * pi * x / n) + c * np.cos(np.float32(4) * pi * x / n) - d * np.cos(np.float32(6) * pi * x / n)).astype(np.float32)


def make_sinc(npoints: int, f_cutoff: float) -> np.ndarray:
    """make_sincs(npoints, factor = 1, f_cutoff, BlackmanHarris2)[0]: window * sinc((x - npoints/2) * cutoff), normalised to unit sum."""
    w = blackman_harris(npoints) ** 2
    x = (np.arange(npoints, dtype=np.float32) - np.float32(npoints // 2)) * np.float32(f_cutoff)
    px = x * np.float32(np.pi)
    with np.errstate(invalid="ignore", divide="ignore"):
        s = np.where(x == 0, np.float32(1), np.sin(px) / px).astype(np.float32)
    y = (w * s).astype(np.float32)
    total = np.float32(0)
    for v in y:                      # sequential f32 sum as in the crate's loop
        total = np.float32(total + v)
    return (y / total).astype(np.float32)


def fft_sizes(rate_in: int, rate_out: int, chunk_size_in: int):
    g = gcd(rate_in, rate_out)
    min_chunk_in = rate_in // g
    fft_chunks = int(np.ceil(np.float32(chunk_size_in) / np.float32(min_chunk_in)))
    return fft_chunks * rate_in // g, fft_chunks * rate_out // g


def cutoff(fft_size_in: int, fft_size_out: int) -> float:
    c = np.float32(0.4) ** (np.float32(16.0) / np.float32(fft_size_in))
    if fft_size_in > fft_size_out:
        c = c * np.float32(fft_size_out) / np.float32(fft_size_in)
    return float(np.float32(c))


class FftFixedInOut:
    """rubato::FftFixedInOut<f32> with one channel (lib.rs:236-242)."""

    def __init__(self, rate_in: int, rate_out: int, chunk_size_in: int):
        self.fft_size_in, self.fft_size_out = fft_sizes(rate_in, rate_out, chunk_size_in)
        fi = self.fft_size_in
        sinc = make_sinc(fi, cutoff(fi, self.fft_size_out))
        filt = np.zeros(2 * fi, np.float32)
        filt[:fi] = sinc / np.float32(2 * fi)
        self.filter_f = sfft.rfft(filt).astype(np.complex64)
        self.overlap = np.zeros(self.fft_size_out, np.float32)

    def input_frames_next(self) -> int:
        return self.fft_size_in

    def output_frames_max(self) -> int:
        return self.fft_size_out

    def reset(self):
        self.overlap[:] = 0

    def process(self, wave_in: np.ndarray) -> np.ndarray:
        fi, fo = self.fft_size_in, self.fft_size_out
        wave_in = np.asarray(wave_in, np.float32)
        if wave_in.shape != (fi,):
            raise ValueError("wrong number of input frames: expected %d, got %d" % (fi, wave_in.size))
        buf = np.zeros(2 * fi, np.float32)
        buf[:fi] = wave_in
        spec = sfft.rfft(buf).astype(np.complex64) * self.filter_f
        new_len = fi + 1 if fi < fo else fo
        out_f = np.zeros(fo + 1, np.complex64)
        out_f[:new_len] = spec[:new_len]
        # realfft's inverse is unnormalised (scipy's irfft divides by n)
        y = (sfft.irfft(out_f, n=2 * fo).astype(np.float32) * np.float32(2 * fo)).astype(np.float32)
        out = (y[:fo] + self.overlap).astype(np.float32)
        self.overlap = y[fo:].copy()
        return out
