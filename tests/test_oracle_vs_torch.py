"""Pins the oracle's three networks against independent implementations: HuggingFace's HuBERT class
(ContentVec-base config) and torch.nn.functional convs / GRU, with identical seeded weights."""
import numpy as np
import pytest

import torch_ref as TR
from common import BASELINE_160MS as g, rel_rms, voice_signal, zoo
from obs_rvc_amd import weights as W
from oracle import oracle as O

torch = pytest.importorskip("torch")


def _ora(preset, version=2):
    z = zoo(preset, version)
    ora = O.OracleRvcInfer(z["data"]); ora.load_contentvec(version); ora.load_f0(1); ora.load_model(z["model"]); ora.set_noise_seed(3, 1)
    ora.enable_taps(True)
    return z, ora


@pytest.mark.parametrize("preset,version", [("tiny", 2), ("tiny", 1), ("full", 2)])
def test_contentvec_matches_hf_hubert(preset, version):
    z, ora = _ora(preset, version)
    cfg, tens = W.read_blob("%s/contentvec/%s" % (z["data"], W.cv_blob_name(version)))
    wav = voice_signal(16000 if preset == "full" else 35840, seed=1)
    got = ora.hubert(wav)[0]
    ref = TR.contentvec_hf(cfg, tens, wav)
    assert got.shape == ref.shape
    assert rel_rms(got, ref) < 2e-5


@pytest.mark.parametrize("preset", ["tiny", "full"])
def test_rmvpe_matches_torch(preset):
    z, ora = _ora(preset)
    cfg, tens = W.read_blob("%s/f0/rmvpe.rvcw" % z["data"])
    wav = voice_signal(35840, seed=2)
    ora.pitch(wav, 0, 2560)
    mel = ora.tap("rm.mel").reshape(128, 32)
    got = ora.tap("rm.sal").reshape(32, 360)
    ref = TR.rmvpe_salience(cfg, tens, mel)
    assert rel_rms(got, ref) < 2e-5


@pytest.mark.parametrize("preset", ["tiny", "full"])
def test_synth_matches_torch(preset):
    z, ora = _ora(preset)
    cfg, tens = W.read_blob(z["model"])
    wav = voice_signal(35840, seed=4)
    audio = ora.infer(wav, g.sample_frame_16k, 12, g.skip_head, g.model_return_length)
    R, I = g.model_return_length, int(cfg["inter"])
    phone = ora.tap("phone").reshape(R, -1)
    pitch = ora.tap("pitch").astype(np.int64)
    eps = O.philox_normal(3, 1, 0, 0, I * R).reshape(I, R)
    enc, stats, zz = TR.synth_until_z(cfg, tens, phone, pitch, eps)
    assert rel_rms(ora.tap("sy.enc").reshape(enc.shape), enc) < 2e-5
    assert rel_rms(ora.tap("sy.stats").reshape(stats.shape), stats) < 2e-5
    assert rel_rms(ora.tap("sy.z").reshape(zz.shape), zz) < 2e-5
    ref_audio = TR.synth_decoder(cfg, tens, ora.tap("sy.z").reshape(zz.shape), ora.tap("sy.src"))
    assert ref_audio.shape == audio.shape
    assert np.sqrt(np.mean((ref_audio - audio) ** 2)) < 2e-5
