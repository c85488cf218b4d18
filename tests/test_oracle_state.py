"""Stateful pieces of RvcInfer::infer restated in numpy and checked against the oracle:
pitch cache (Q6, rvc.rs:167-179), slicing (rvc.rs:153-155), committed golden chain."""
import os

import numpy as np

from common import BASELINE_160MS as g, GOLDEN, chunk_stream, derive, voice_signal, zoo
from oracle import oracle as O


def _mk(seed=(1, 0)):
    z = zoo("tiny")
    ora = O.OracleRvcInfer(z["data"]); ora.load_contentvec(2); ora.load_f0(1); ora.load_model(z["model"]); ora.set_noise_seed(*seed)
    ora.enable_taps(True)
    return ora


def test_pitch_cache_q6():
    ora = _mk()
    audio = voice_signal(g.sample_frame_16k * 20, seed=5)
    cache = np.zeros(1024, np.float32)
    for ring in list(chunk_stream(audio, g.input_buffer_16k_size, g.sample_frame_16k))[-5:]:
        ora.infer(ring, g.sample_frame_16k, 12, g.skip_head, g.model_return_length)
        f0 = ora.tap("f0")
        shift = g.sample_frame_16k // 160
        cache[:1024 - shift] = cache[shift:].copy()
        cache[1024 + 4 - len(f0):] = f0[3:len(f0) - 1]
        assert np.array_equal(ora.pitch_cache(), cache)
        hubert_length = min(g.input_buffer_16k_size // 160, 2 * 111 + 1)
        start = 1024 - hubert_length + g.skip_head
        assert np.array_equal(ora.tap("pitchf"), cache[start:start + g.model_return_length])
    # never reset by the reference; reset_state is an extension
    ora.reset_state()
    assert not ora.pitch_cache().any()


def test_default_300ms_geometry_runs():
    gg = derive(48000, 0.30, 0.07, 2.0, 48000)
    ora = _mk()
    x = voice_signal(gg.input_buffer_16k_size, seed=2)
    y = ora.infer(x, gg.sample_frame_16k, 0, gg.skip_head, gg.model_return_length)
    assert y.shape == (gg.model_return_length * 48,)      # tiny synth: upp = 4*3*2*2
    assert len(ora.tap("f0")) == 64


def test_noise_is_explicit_and_counter_based():
    x = voice_signal(g.input_buffer_16k_size, seed=2)
    a, b = _mk((5, 0)), _mk((5, 0))
    ya1, ya2 = a.infer(x, 2560, 12, 200, 21), a.infer(x, 2560, 12, 200, 21)
    yb1 = b.infer(x, 2560, 12, 200, 21)
    assert np.array_equal(ya1, yb1)
    assert not np.array_equal(ya1, ya2)                    # chunk counter advanced
    c = _mk((5, 1))
    assert not np.array_equal(c.infer(x, 2560, 12, 200, 21), ya1)


def test_golden_chain_reproduces():
    d = np.load(os.path.join(GOLDEN, "tiny_chain.npz"))
    ora = _mk((99, 5))
    rings = list(chunk_stream(d["audio"], g.input_buffer_16k_size, g.sample_frame_16k))[-4:]
    for i, r in enumerate(rings):
        y = ora.infer(r, g.sample_frame_16k, 7 if i % 2 else -12, g.skip_head, g.model_return_length)
        assert np.abs(y - d["outs"][i]).max() < 1e-5
    assert np.allclose(ora.pitch_cache(), d["cache"], rtol=1e-5, atol=1e-4)


def test_retrieval_blend_definition():
    z = zoo("tiny")
    ora = _mk()
    rng = np.random.default_rng(7)
    index = (rng.standard_normal((2000, 48)) * 0.35).astype(np.float32)
    x = voice_signal(g.input_buffer_16k_size, seed=2)
    ora.infer(x, 2560, 12, 200, 21)
    plain = ora.tap("phone").reshape(21, 48)
    ora2 = _mk(); ora2.load_index(index); ora2.set_index_rate(0.75)
    ora2.infer(x, 2560, 12, 200, 21)
    blended = ora2.tap("phone").reshape(21, 48)
    idx, dist = ora2.knn()
    ridx, rdist = O.knn_search(index, plain, 4)
    assert np.array_equal(idx, ridx) and np.array_equal(dist, rdist)
    w = (1.0 / dist.astype(np.float64)) ** 2
    w /= w.sum(1, keepdims=True)
    ref = 0.75 * (w[..., None] * index[idx]).sum(1) + 0.25 * plain
    assert np.abs(blended - ref).max() < 1e-5
    # Q2: duplicated frames get identical hits
    assert np.array_equal(idx[0], idx[1]) and np.array_equal(idx[2], idx[3])
