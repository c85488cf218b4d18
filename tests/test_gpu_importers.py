"""Row f4 on the GPU: a model in the form users have it (PyTorch state dict / ONNX initializers of the public architectures)
-> obs_rvc_amd.importers -> .rvcw blob -> the HIP engine, compared with the forward pass of the torch module the weights came from.
The CPU suite (tests/test_importers.py) makes the same comparison through the oracle; here the imported blob runs on the product path."""
import os

import numpy as np
import pytest

from common import BASELINE_160MS as g, rel_rms, rms, voice_signal, zoo
from obs_rvc_amd import importers as IM, weights as W
from test_importers import _hf_model, _upstream_rmvpe, _upstream_synth

pytestmark = pytest.mark.gpu


def test_imported_contentvec_runs_on_the_engine_like_hf_hubert(tmp_path):
    import torch
    from obs_rvc_amd.rvc import RvcInfer
    m = _hf_model()
    named = {k: v.detach().numpy() for k, v in m.state_dict().items()}
    cfg, tens = IM.import_contentvec(named, version=2, heads=4, pos_groups=4)
    d = tmp_path / "data"
    os.makedirs(d / "contentvec")
    W.write_blob(str(d / "contentvec" / W.cv_blob_name(2)), cfg, tens)
    eng = RvcInfer(str(d)); eng.load_contentvec(2)
    for n in (12000, 35840):
        wav = voice_signal(n, seed=2)
        got = eng.hubert(wav)[0]                                                   # (C, T)
        with torch.no_grad():
            ref = m(torch.from_numpy(wav)[None]).last_hidden_state[0].T.numpy()
        assert got.shape == ref.shape and rel_rms(got, ref) < 1e-4, (n, rel_rms(got, ref))
        feat = eng.extract_feature(wav)[0]                                         # (2T+1, C): frame k = raw frame min(k // 2, T - 1) (rvc.rs:99-109)
        T = ref.shape[1]
        assert feat.shape == (2 * T + 1, ref.shape[0]) and rel_rms(feat[2 * (T - 1)], ref[:, T - 1]) < 1e-4 and np.array_equal(feat[2 * T], feat[2 * T - 1])


def test_imported_rmvpe_runs_on_the_engine_like_the_upstream_module(tmp_path):
    import torch
    from obs_rvc_amd.rvc import RvcInfer
    from obs_rvc_amd.rvc_common import RvcInferError
    m = _upstream_rmvpe()
    named = {k: v.detach().numpy() for k, v in m.state_dict().items() if "num_batches_tracked" not in k}
    cfg, tens = IM.import_rmvpe(named)
    d = tmp_path / "data"
    os.makedirs(d / "f0")
    W.write_blob(str(d / "f0" / "rmvpe.rvcw"), cfg, tens)
    eng = RvcInfer(str(d)); eng.load_f0(1); eng.enable_taps(True)
    try:
        eng.pitch(voice_signal(35840, seed=4), 0, 2560)
    except RvcInferError as ex:                  # an untrained head may decode to a bin where the reference panics: the taps are still there
        assert "Panic" in str(ex)
    mel = eng.tap("rm.mel").reshape(128, 32)
    got = eng.tap("rm.sal_ct").reshape(360, 32).T
    with torch.no_grad():
        ref = m(torch.from_numpy(np.ascontiguousarray(mel))[None])[0].numpy()
    assert ref.shape == got.shape and rel_rms(got, ref) < 1e-4, rel_rms(got, ref)


def _coarse_pitch(f0):
    # get_f0_post, rvc/src/f0/mod.rs:7-12 (f32 arithmetic, round half away from zero)
    f0 = f0.astype(np.float32)
    mn, mx = np.float32(1127.0) * np.log(np.float32(1.0) + np.float32(50.0 / 700.0)), np.float32(1127.0) * np.log(np.float32(1.0) + np.float32(500.0 / 700.0))
    mel = np.float32(1127.0) * np.log(np.float32(1.0) + f0 / np.float32(700.0))
    pos = mel > 0
    mel = np.where(pos, (mel - mn) * np.float32(254.0) / (mx - mn) + np.float32(1.0), mel)
    mel = np.clip(mel, 1.0, 255.0)
    return np.floor(mel + 0.5).astype(np.int64)


def test_imported_synthesizer_runs_on_the_engine_like_the_upstream_module(tmp_path):
    import torch
    from oracle import oracle as O          # the counter-based noise definition (Philox) lives with the oracle: the checker, not the product
    from obs_rvc_amd.rvc import RvcInfer
    m = _upstream_synth()
    sd = {k: v.detach().numpy() for k, v in m.state_dict().items()}
    cfg, tens = IM.import_synth(sd, sid=1, sr=4800, up_rates=[4, 3, 2, 2], heads=2)
    model = str(tmp_path / "upstream.rvcw")
    W.write_blob(model, cfg, tens)
    z = zoo("tiny")
    eng = RvcInfer(z["data"]); eng.load_contentvec(2); eng.load_f0(1); eng.load_model(model); eng.set_noise_seed(3, 1); eng.enable_taps(True)
    R = 21
    audio = eng.infer(voice_signal(35840, seed=4), 2560, 12, 200, R)
    C = int(cfg["phone_dim"])
    phone = torch.from_numpy(np.ascontiguousarray(eng.tap("phone_ct").reshape(C, R).T))[None]
    pitchf = eng.pitch_cache()[1024 - 223 + 200:1024 - 223 + 200 + R]               # rvc.rs:176-177 at the 160 ms geometry
    pitch = torch.from_numpy(_coarse_pitch(pitchf))[None]
    I = 16
    with torch.no_grad():
        gvec = m.emb_g(torch.tensor([1])).unsqueeze(-1)
        mp, logs = m.enc_p(phone, pitch)
        stats = torch.cat([mp, logs], 1)[0].numpy()
        assert rel_rms(eng.tap("sy.stats").reshape(stats.shape), stats) < 1e-4
        eps = torch.from_numpy(O.philox_normal(3, 1, 0, 0, I * R).reshape(I, R))[None]
        zp = mp + torch.exp(logs) * eps * 0.66666
        zz = m.flow(zp, torch.ones(1, 1, R), gvec, reverse=True)
        assert rel_rms(eng.tap("sy.z").reshape(I, R), zz[0].numpy()) < 1e-4
        src = torch.from_numpy(np.ascontiguousarray(eng.tap("sy.src")).reshape(1, 1, -1))
        ref = m.dec(zz, src, gvec)[0, 0].numpy()
    assert ref.shape == audio.shape and rms(ref - audio) < 1e-4, rms(ref - audio)
