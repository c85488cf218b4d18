"""Multi-GPU layer: streams are independent units, so the per-chunk path has NO collective.

* stream s -> rank s mod G (round-robin, BASELINE config 5: 512 streams -> 64 per GPU);
* the only exchange step is at load: the shared retrieval index is broadcast from rank 0 with ONE ncclBroadcast issued by the
  engine itself (rvc_index_broadcast in the C ABI: librccl over xGMI, no Python needed by a Rust / C host).  The host's only job
  is to hand rank 0's 128-byte unique id to the other ranks; here that goes through torch.distributed's object broadcast.
  `broadcast_index` is the host-side equivalent on torch tensors ("gloo" in the CPU tests).
The reference has no multi-stream or multi-GPU mode (one RvcInfer per process, rvc.rs:133-134)."""
from __future__ import annotations

import ctypes as C
from typing import List, Optional

import numpy as np


def shard_streams(n_streams: int, world: int) -> List[List[int]]:
    """Round-robin stream -> rank assignment: rank r owns streams r, r+G, r+2G, ..."""
    return [list(range(r, n_streams, world)) for r in range(world)]


def local_streams(n_streams: int, rank: int, world: int) -> List[int]:
    return shard_streams(n_streams, world)[rank]


def broadcast_index(vecs: Optional[np.ndarray], n: int, dim: int, rank: int, world: int, device: str = "cpu"):
    """Rank 0 passes the (n, dim) fp32 index, the other ranks pass None; every rank gets a tensor on `device`."""
    import torch
    import torch.distributed as dist

    t = torch.empty((n, dim), dtype=torch.float32, device=device)
    if rank == 0:
        t.copy_(torch.from_numpy(np.ascontiguousarray(vecs, dtype=np.float32)))
    if world > 1:
        dist.broadcast(t, src=0)
    return t


def exchange_unique_id(eng, rank: int, world: int) -> bytes:
    """Rank 0 asks the engine (librccl) for a unique id; every rank returns the same 128 bytes."""
    box = [eng.rccl_unique_id() if rank == 0 else None]
    if world > 1:
        import torch.distributed as dist
        dist.broadcast_object_list(box, src=0)
    return box[0]


def load_shared_index(eng, vecs: Optional[np.ndarray], n: int, dim: int, rank: int, world: int) -> None:
    """Every rank ends up with the same HBM-resident index: rank 0 uploads it once and the engine broadcasts it over RCCL / xGMI
    (with world == 1 the call still goes through RCCL, a one-rank communicator)."""
    uid = exchange_unique_id(eng, rank, world)
    eng.index_broadcast(uid, rank, world, vecs if rank == 0 else None)
    p, nbytes = eng.index_device_ptr()
    if nbytes != n * dim * 4:
        raise RuntimeError("index broadcast delivered %d bytes, expected %d" % (nbytes, n * dim * 4))
