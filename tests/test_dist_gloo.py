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


def _last_json(text):
    import json
    for ln in reversed(text.strip().splitlines()):
        if ln.startswith("{"):
            return json.loads(ln)
    raise AssertionError("no JSON line in: " + text)


def test_bench_launcher_spawns_ranks_dry():
    # `bench.py --gpus 2` with no torchrun environment spawns the two ranks itself (one process per GPU, 127.0.0.1 rendezvous);
    # --dry-launch replaces the GPU work by a stand-in so the launcher, the rendezvous, the round-robin sharding (stream s of the
    # job on rank s mod N) and the reductions run here.  BASELINE configs[4] shape: 64 streams per rank.
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dry-launch", "--steps", "4", "--streams", "64"],
                       env=env, capture_output=True, timeout=300)
    assert r.returncode == 0, r.stderr.decode()
    j = _last_json(r.stdout.decode())
    assert j["dry_launch"] is True and j["n_gpus"] == 2 and j["streams_total"] == 128 and j["streams_covered"] is True
    assert j["per_rank_streams"][0][:3] == [0, 2, 4] and j["per_rank_streams"][1][:3] == [1, 3, 5] and j["latency_samples"] == 8
    # more than one rank: BASELINE configs[4] is also a top-level key of the line (the driver's parser cannot miss it)
    assert j["config4"]["n_gpus"] == 2 and j["config4"]["streams_total"] == 128 and j["config4"]["per_rank_streams"] == [64, 64]


def test_bench_launcher_keeps_rank_logs_and_reports_the_failing_rank(tmp_path):
    # every rank's stdout / stderr go to files; a rank that fails takes the run down with ITS last lines (round 2 sent the children's
    # output to /dev/null: a failing rank 5 of 8 would have been a bare exit code)
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env["RVC_BENCH_LOGDIR"] = str(tmp_path)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dry-launch", "--steps", "2"], env=env, capture_output=True, timeout=300)
    assert r.returncode == 0, r.stderr.decode()
    for rk in (0, 1):
        assert (tmp_path / ("rank%d.out" % rk)).exists() and (tmp_path / ("rank%d.err" % rk)).exists()
    assert _last_json((tmp_path / "rank0.out").read_text())["n_gpus"] == 2
    env["RVC_BENCH_FAIL_RANK"] = "1"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dry-launch", "--steps", "2"], env=env, capture_output=True, timeout=300)
    assert r.returncode != 0
    err = r.stderr.decode()
    assert "rank 1 exited with 3" in err and "simulated failure of rank 1" in err and "rank exit codes [0, 3]" in err, err
    assert _last_json(r.stdout.decode())["n_gpus"] == 2          # rank 0's line is still replayed


def test_bench_under_torchrun_dry():
    # the driver's own launch line: python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ...
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dry-launch", "--steps", "3"],
                       env=env, capture_output=True, timeout=300)
    assert r.returncode == 0, r.stderr.decode()
    j = _last_json(r.stdout.decode())
    assert j["n_gpus"] == 2 and j["streams_covered"] is True and j["per_rank_streams"] == [[0], [1]]
