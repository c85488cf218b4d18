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


def _all_agree(ok: bool, world: int) -> bool:
    """True only if every rank reports success (one tiny all-reduce on the default process group)."""
    if world == 1:
        return bool(ok)
    import torch
    import torch.distributed as dist
    dev = "cuda" if dist.get_backend() == "nccl" else "cpu"
    t = torch.tensor([1.0 if ok else 0.0], dtype=torch.float64, device=dev)
    dist.all_reduce(t, op=dist.ReduceOp.MIN)
    return bool(t.item() > 0.5)


def exchange_unique_id(eng, rank: int, world: int) -> bytes:
    """Rank 0 asks the engine (librccl) for a unique id; every rank returns the same 128 bytes.  A failure on rank 0 still travels
    through the object broadcast (as None), so that every rank raises in step instead of waiting for an id that never comes."""
    box = [None]
    err = None
    if rank == 0:
        try:
            box[0] = eng.rccl_unique_id()
        except Exception as ex:                       # librccl refused: tell the others
            err = ex
    if world > 1:
        import torch.distributed as dist
        dist.broadcast_object_list(box, src=0)
    if box[0] is None:
        raise RuntimeError("rank 0 could not create an RCCL unique id" + (": %s" % err if err else ""))
    return box[0]


def load_shared_index(eng, vecs: Optional[np.ndarray], n: int, dim: int, rank: int, world: int) -> None:
    """Every rank ends up with the same HBM-resident index.  One rank: a plain upload (no RCCL, no librccl needed).  More: rank 0
    uploads it once and the engine broadcasts it over RCCL / xGMI (rvc_index_broadcast); the ranks first agree that all of them can
    load librccl, so that none waits in the communicator set-up for a peer that never joins."""
    if world == 1:
        eng.load_index(vecs)
    else:
        if not _all_agree(eng.rccl_available(), world):
            raise RuntimeError("librccl is not loadable on every rank")
        # every rank's own arguments, agreed on BEFORE any communicator exists (the engine also fails together from inside the
        # collective, but a host that can tell earlier should not start one)
        mine = (vecs is not None and np.ndim(vecs) == 2 and np.shape(vecs) == (n, dim) and n >= 4) if rank == 0 else (n >= 4 and dim >= 1)
        if not _all_agree(bool(mine), world):
            raise RuntimeError("index arguments rejected on at least one rank (rank 0 needs the (n, dim) matrix with n >= 4)")
        uid = exchange_unique_id(eng, rank, world)
        eng.index_broadcast(uid, rank, world, vecs if rank == 0 else None, expect=None if rank == 0 else (n, dim))
    p, nbytes = eng.index_device_ptr()
    if nbytes != n * dim * 4:
        raise RuntimeError("index load delivered %d bytes, expected %d" % (nbytes, n * dim * 4))
