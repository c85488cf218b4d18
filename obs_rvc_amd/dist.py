"""Multi-GPU layer: streams are independent units, so the per-chunk path has NO collective.

* stream s -> rank s mod G (round-robin, BASELINE config 5: 512 streams -> 64 per GPU);
* the only exchange step is at load: the shared retrieval index is broadcast from rank 0 with
  torch.distributed (backend "nccl" = RCCL over xGMI on the GPU box, "gloo" in the CPU tests).
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


def load_shared_index(eng, vecs: Optional[np.ndarray], n: int, dim: int, rank: int, world: int) -> None:
    """Broadcast over RCCL (device to device) and hand the HBM-resident copy to the engine."""
    import torch

    t = broadcast_index(vecs, n, dim, rank, world, device="cuda")
    torch.cuda.synchronize()
    rc = eng._L.rvc_load_index_device(eng._h, C.c_void_p(t.data_ptr()), n, dim)
    if rc != 0:
        from .rvc_common import RvcInferError
        raise RvcInferError(rc, "rvc_load_index_device")
