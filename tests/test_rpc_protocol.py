"""Wire format of the process boundary."""
import os
import struct

import numpy as np
import pytest

from common import BASELINE_160MS as g, rms, voice_signal, zoo
from obs_rvc_amd import _native, rpc


def test_request_bytes_match_appendix_b():
    pcm = np.zeros(35840, np.float32)
    req = rpc.encode_request(pcm, 2560, 12, 200, 21)
    assert req[:4] == bytes.fromhex("00300200") and len(req) == 4 + 143360 + 16
    assert req[-16:] == bytes.fromhex("000a0000" "0c000000" "c8000000" "15000000")
    p, fr, sh, sk, rl = rpc.decode_request(req)
    assert (len(p), fr, sh, sk, rl) == (35840, 2560, 12, 200, 21)
    assert rpc.encode_request(pcm, 2560, -12, 200, 21)[-12:-8] == struct.pack("<i", -12)
    assert rpc.encode_reply(np.zeros(10080, np.float32))[:4] == bytes.fromhex("809d0000")


def test_rpc_binary_usage_line():
    if not os.path.exists(_native.RPC_PATH):
        _native.build()
    import subprocess
    r = subprocess.run([_native.RPC_PATH, "v2", "rmvpe"], capture_output=True, timeout=30)
    assert r.returncode == 0 and b"Usage: rvc-rpc <version> <f0_algorithm> <model> <data>" in r.stderr
    # the reference checks `args.len() < 4` but reads args[4] (rvc-rpc/src/main.rs:14,22): three arguments pass the usage check
    # and then panic on the index -> exit status 101, no usage line
    r = subprocess.run([_native.RPC_PATH, "v2", "rmvpe", "model.onnx"], capture_output=True, timeout=30)
    assert r.returncode == 101 and b"Usage" not in r.stderr


@pytest.mark.gpu
def test_rpc_server_end_to_end():
    from oracle import oracle as O
    z = zoo("tiny")
    env = dict(os.environ, RVC_NOISE_SEED="77")
    cli = rpc.RpcEngine("v2", "rmvpe", z["model"], z["data"], env=env)
    try:
        assert "Ready to receive input" in cli.wait_ready()
        ora = O.OracleRvcInfer(z["data"]); ora.load_contentvec(2); ora.load_f0(1); ora.load_model(z["model"]); ora.set_noise_seed(77, 0)
        for i in range(3):
            x = voice_signal(g.input_buffer_16k_size, seed=20 + i)
            y = cli.infer(x, g.sample_frame_16k, 12, g.skip_head, g.model_return_length)
            yo = ora.infer(x, g.sample_frame_16k, 12, g.skip_head, g.model_return_length)
            assert y.shape == yo.shape and rms(y - yo) < 1e-3
    finally:
        cli.close()
    # a missing model file kills the server (the reference panics, rvc-rpc/src/main.rs:49-54)
    bad = rpc.RpcEngine("v2", "rmvpe", "/nonexistent/model.rvcw", z["data"])
    assert bad.p.wait(timeout=120) != 0
