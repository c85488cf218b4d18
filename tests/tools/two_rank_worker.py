#!/usr/bin/env python
"""TEST INFRASTRUCTURE: one rank of a world-N run of the C-ABI index broadcast on ONE GPU (tests/test_gpu_multi.py starts N of these
with RVC_RCCL_LIB = tests/tools/fake_rccl.cpp's library).  Rank 0 creates the unique id through rvc_rccl_unique_id and leaves it in a
file (the "any host-side means" of include/rvc_mi355x.h); every rank calls rvc_index_broadcast, runs one chunk with retrieval on the
same input and writes what it got as JSON.

usage: two_rank_worker.py <rank> <world> <workdir> <scenario> [port]
scenario: ok | mismatch (rank 1 expects another shape) | root_bad (rank 0 passes 2 vectors) | dist (the host-side path bench.py --gpus N takes:
obs_rvc_amd.dist.load_shared_index over a torch.distributed group -- gloo here -- which agrees on librccl, on the arguments, hands the id round)"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from common import BASELINE_160MS as g, voice_signal, zoo  # noqa: E402
from obs_rvc_amd import weights as W  # noqa: E402
from obs_rvc_amd.rvc import RvcInfer  # noqa: E402

rank, world, work, scenario = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], sys.argv[4]
z = zoo("tiny")
eng = RvcInfer(z["data"], device=0)
eng.load_contentvec(2); eng.load_f0(1); eng.load_model(z["model"]); eng.set_noise_seed(21, 0)
dim = eng.hubert(voice_signal(g.input_buffer_16k_size, seed=1)).shape[1]
N = 3000
index = W.make_index(N, dim, seed=5)
uid_path = os.path.join(work, "uid.bin")
res = {"rank": rank, "scenario": scenario}
if scenario == "dist":
    import torch.distributed as tdist
    from obs_rvc_amd import dist as rdist
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = sys.argv[5]
    tdist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        rdist.load_shared_index(eng, index if rank == 0 else None, N, dim, rank, world)
        res["error"] = None
    except Exception as ex:      # noqa: BLE001
        res["error"] = "%s: %s" % (type(ex).__name__, ex)
    tdist.barrier(); tdist.destroy_process_group()
elif rank == 0:
    uid = eng.rccl_unique_id()
    with open(uid_path + ".tmp", "wb") as f:
        f.write(uid)
    os.replace(uid_path + ".tmp", uid_path)
else:
    t0 = time.time()
    while not os.path.exists(uid_path):
        time.sleep(0.01)
        assert time.time() - t0 < 120, "rank 0 never published the unique id"
    uid = open(uid_path, "rb").read()
if scenario != "dist":
    try:
        if rank == 0:
            eng.index_broadcast(uid, 0, world, index[:2] if scenario == "root_bad" else index)
        else:
            expect = (N + 1, dim) if (scenario == "mismatch" and rank == 1) else (N, dim)
            eng.index_broadcast(uid, rank, world, None, expect=expect)
        res["error"] = None
    except Exception as ex:      # noqa: BLE001 -- the failure IS the result
        res["error"] = "%s: %s" % (type(ex).__name__, ex)
if res["error"] is None:
    info = eng.index_broadcast_info()
    ptr, nbytes = eng.index_device_ptr()
    eng.set_index_rate(0.75)
    x = voice_signal(g.input_buffer_16k_size, seed=3)
    y = eng.infer(x, g.sample_frame_16k, 12, g.skip_head, g.model_return_length)
    idx, dist = eng.knn()
    res.update(ranks=info["ranks"], index_bytes=int(nbytes), hits=idx.tolist(), dist=dist.tolist(), pcm=[float(v) for v in y[:256]], pcm_rms=float(np.sqrt(np.mean(y * y))))
with open(os.path.join(work, "rank%d.json" % rank), "w") as f:
    json.dump(res, f)
eng.close()
