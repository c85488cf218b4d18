"""Shared helpers for the test-suite (inputs of BASELINE.md section 2, zoo set-up, comparisons)."""
from __future__ import annotations

import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from obs_rvc_amd import weights as W  # noqa: E402
from obs_rvc_amd.geometry import BASELINE_160MS, derive  # noqa: E402

GOLDEN = os.path.join(ROOT, "tests", "golden")


def voice_signal(n: int, seed: int = 0, sr: int = 16000) -> np.ndarray:
    """110->220 Hz glide, 8 harmonics (amp 0.1) + white noise sigma 0.003."""
    rng = np.random.default_rng(seed)
    t = np.arange(n) / sr
    f = 110.0 * (2.0 ** (t / max(t[-1], 1e-9))) * (1.0 + 0.01 * seed)
    phase = 2 * np.pi * np.cumsum(f) / sr
    x = sum(np.sin((h + 1) * phase) / (h + 1) for h in range(8)) * 0.1 / 1.7
    return (x + 0.003 * rng.standard_normal(n)).astype(np.float32)


def chunk_stream(audio: np.ndarray, ring_len: int, chunk: int):
    """Feed `audio` chunk by chunk through a zero-initialised ring of ring_len samples (lib.rs:669-683)."""
    ring = np.zeros(ring_len, np.float32)
    for i in range(0, len(audio) - chunk + 1, chunk):
        ring[:-chunk] = ring[chunk:]
        ring[-chunk:] = audio[i:i + chunk]
        yield ring.copy()


def zoo(preset: str = "tiny", version: int = 2, synth_preset=None):
    return W.build_model_zoo(W.default_zoo_root(preset), preset, version, synth_preset)


def rms(a):
    a = np.asarray(a, np.float64)
    return float(np.sqrt(np.mean(a * a))) if a.size else 0.0


def rel_rms(a, b):
    return rms(np.asarray(a, np.float64) - np.asarray(b, np.float64)) / max(rms(b), 1e-12)


# oracle tap name -> (engine tap name, transpose shape or None)
TAP_MAP = [
    ("cv.conv0", "cv.conv0", None), ("cv.feat", "cv.feat", None), ("cv.proj", "cv.proj", None), ("cv.pos", "cv.pos", None),
    ("cv.l0", "cv.l0", None), ("cv.out", "cv.out", None),
    ("rm.mel", "rm.mel", None), ("rm.enc0", "rm.enc0", None), ("rm.enc4", "rm.enc4", None), ("rm.int", "rm.int", None),
    ("rm.dec0", "rm.dec0", None), ("rm.dec4", "rm.dec4", None),
    ("rm.cnn", "rm.cnn_ct", "T"), ("rm.gru", "rm.gru_ct", "T"), ("rm.sal", "rm.sal_ct", "T"), ("f0", "f0", None),
    ("phone", "phone_ct", "T"), ("sy.emb", "sy.emb", None), ("sy.enc", "sy.enc", None), ("sy.stats", "sy.stats", None),
    ("sy.zp", "sy.zp", None), ("sy.z", "sy.z", None), ("sy.src", "sy.src", None), ("sy.pre", "sy.pre", None),
    ("sy.up0", "sy.up0", None), ("sy.rb0", "sy.rb0", None), ("sy.up3", "sy.up3", None), ("sy.rb3", "sy.rb3", None),
]


def compare_taps(ora, eng, rows_hint, skip=()):
    """Yield (name, rel_rms_error, n).  rows_hint maps an oracle tap name to its leading dimension for transposes."""
    for oname, ename, tr in TAP_MAP:
        if oname in skip:
            continue
        try:
            a = ora.tap(oname)
        except KeyError:
            continue
        b = eng.tap(ename)
        if a.size != b.size:
            yield oname, float("inf"), a.size
            continue
        if tr == "T":
            rows = rows_hint[oname]
            b = b.reshape(-1, rows).T.reshape(-1)
        yield oname, rel_rms(b, a), a.size


def set_opt(name: str, value=None) -> None:
    """Set (or clear, value=None) one of the library's test hooks -- an explicit call on the loaded library, not an environment
    variable (csrc/engine.hip "switches": the product reads no tuning variable from the environment)."""
    import ctypes as C
    from obs_rvc_amd import _native
    L = _native.lib()
    L.rvc_debug_option.argtypes = [C.c_char_p, C.c_char_p]
    L.rvc_debug_option.restype = C.c_int
    rc = L.rvc_debug_option(name.encode(), None if value is None else str(value).encode())
    assert rc == 0, "unknown test hook %s" % name
