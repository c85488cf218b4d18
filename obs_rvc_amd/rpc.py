"""Client side of the rvc-rpc stdio protocol, as the plugin's adapter speaks it
(reference: obs-rvc/src/rvcadapter.rs:34-119; server: rvc-rpc/src/main.rs:64-101)."""
from __future__ import annotations

import struct
import subprocess
from typing import Optional

import numpy as np

from . import _native


def encode_request(pcm16k: np.ndarray, sample_frame_16k_size: int, pitch_shift: int, skip_head: int, return_length: int) -> bytes:
    body = np.ascontiguousarray(pcm16k, dtype="<f4").tobytes()
    return struct.pack("<I", len(body)) + body + struct.pack("<IiII", sample_frame_16k_size, pitch_shift, skip_head, return_length)


def decode_request(buf: bytes):
    (nbytes,) = struct.unpack_from("<I", buf, 0)
    pcm = np.frombuffer(buf, dtype="<f4", count=nbytes // 4, offset=4)
    frame, shift, skip, ret = struct.unpack_from("<IiII", buf, 4 + nbytes)
    return pcm, frame, shift, skip, ret


def encode_reply(pcm: np.ndarray) -> bytes:
    body = np.ascontiguousarray(pcm, dtype="<f4").tobytes()
    return struct.pack("<I", len(body)) + body


class RpcEngine:
    """Same role as obs-rvc's rvcadapter::RvcInfer: spawn the child, one blocking request per chunk, kill on drop."""

    def __init__(self, version: str, f0_algorithm: str, model_path: str, data_path: str, exe: Optional[str] = None, env=None):
        self.p = subprocess.Popen([exe or _native.RPC_PATH, version, f0_algorithm, str(model_path), str(data_path)],
                                  stdin=subprocess.PIPE, stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env)

    def wait_ready(self) -> str:
        line = self.p.stderr.readline().decode()
        while line and "Ready to receive input" not in line:
            line = self.p.stderr.readline().decode()
        return line

    def infer(self, pcm16k, sample_frame_16k_size, pitch_shift, skip_head, return_length) -> np.ndarray:
        self.p.stdin.write(encode_request(pcm16k, sample_frame_16k_size, pitch_shift, skip_head, return_length))
        self.p.stdin.flush()
        hdr = self.p.stdout.read(4)
        if len(hdr) != 4:
            raise IOError("rvc-rpc died (exit %s): %s" % (self.p.poll(), self.p.stderr.read().decode()[-400:]))
        (n,) = struct.unpack("<I", hdr)
        return np.frombuffer(self.p.stdout.read(n), dtype="<f4").copy()

    def close(self):
        if self.p.poll() is None:
            self.p.kill()
        self.p.wait()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
