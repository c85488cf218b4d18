"""world_size-2 gloo test of the only exchange step of the multi-GPU path: the index broadcast at load,
plus the round-robin stream sharding.  (On the GPU box the same code runs over RCCL.)"""
import os
import socket
import subprocess
import sys

import numpy as np

from common import ROOT

WORKER = r"""
import os, sys
import numpy as np
sys.path.insert(0, sys.argv[1])
import torch.distributed as dist
from obs_rvc_amd import dist as rd, weights as W
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo", rank=rank, world_size=world)
vecs = W.make_index(500, 48, seed=7) if rank == 0 else None
t = rd.broadcast_index(vecs, 500, 48, rank, world, device="cpu")
ref = W.make_index(500, 48, seed=7)
assert np.array_equal(t.numpy(), ref), "broadcast mismatch on rank %d" % rank
mine = rd.local_streams(9, rank, world)
import torch
cnt = torch.tensor([len(mine)]); dist.all_reduce(cnt)
assert int(cnt) == 9
print("rank", rank, "ok", mine)
dist.barrier(); dist.destroy_process_group()
"""


def test_gloo_broadcast_and_sharding(tmp_path):
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    script = tmp_path / "w.py"
    script.write_text(WORKER)
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, str(script), ROOT], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT))
    outs = [p.communicate(timeout=240)[0].decode() for p in procs]
    assert all(p.returncode == 0 for p in procs), outs
    assert "rank 0 ok [0, 2, 4, 6, 8]" in outs[0] and "rank 1 ok [1, 3, 5, 7]" in outs[1]
