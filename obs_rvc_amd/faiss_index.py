"""Reader for the Faiss `.index` files RVC users have next to their model (the plugin plumbs
`index_path` / `index_rate` through its settings, obs-rvc/src/lib.rs:78,81,331,337, and the reference leaves the search as
`// TODO: index search`, rvc/src/rvc.rs:159).  Upstream RVC builds `index_factory(dim, "IVF<n>,Flat")`, adds every training
feature and, for real-time use, reconstructs the stored vectors (`index.reconstruct_n(0, ntotal)`): the engine's flat-L2
retrieval needs exactly that matrix, so `read_index` returns the (ntotal, d) float32 vectors in id order.

Faiss is not installed here and the reference holds no index file: the layout below restates faiss/impl/index_write.cpp /
index_read.cpp (IndexFlat "IxF2" / "IxFI" / "IxFl", IndexIVFFlat "IwFl" with "ilar" inverted lists, IndexIDMap "IxMp").
PARITY UNPINNED against a file written by Faiss itself; pinned at byte level against tests/golden/faiss_*.index, which a separate
C restatement of index_write.cpp's WRITE1 / WRITEVECTOR / WRITEXBVECTOR sequence produces (tests/golden/make_faiss_fixture.c);
`write_flat` / `write_ivf_flat` emit the same layout for the round-trip tests."""
from __future__ import annotations

import struct
from typing import BinaryIO, List, Optional, Tuple

import numpy as np


class IndexFormatError(ValueError):
    pass


class _R:
    def __init__(self, f: BinaryIO):
        self.f = f

    def raw(self, n: int) -> bytes:
        b = self.f.read(n)
        if len(b) != n:
            raise IndexFormatError("truncated index file")
        return b

    def fourcc(self) -> str:
        return self.raw(4).decode("latin1")

    def i32(self) -> int: return struct.unpack("<i", self.raw(4))[0]
    def i64(self) -> int: return struct.unpack("<q", self.raw(8))[0]
    def u64(self) -> int: return struct.unpack("<Q", self.raw(8))[0]
    def u8(self) -> int: return self.raw(1)[0]
    def f32(self) -> float: return struct.unpack("<f", self.raw(4))[0]

    def vec(self, dtype, max_items: int = 1 << 34) -> np.ndarray:
        n = self.u64()
        if n > max_items:
            raise IndexFormatError("implausible vector length %d" % n)
        return np.frombuffer(self.raw(n * np.dtype(dtype).itemsize), dtype=np.dtype(dtype).newbyteorder("<")).astype(dtype)


def _header(r: _R) -> Tuple[int, int, int]:
    d, ntotal = r.i32(), r.i64()
    r.i64(); r.i64()                                   # two dummies (1 << 20)
    r.u8()                                             # is_trained
    metric = r.i32()
    if metric > 1:
        r.f32()                                        # metric_arg
    if d <= 0 or d > 1 << 16 or ntotal < 0:
        raise IndexFormatError("bad index header (d=%d, ntotal=%d)" % (d, ntotal))
    return d, ntotal, metric


def _read(r: _R) -> Tuple[np.ndarray, Optional[np.ndarray]]:
    """-> (vectors (n, d) in storage order, ids or None when storage order is id order)."""
    cc = r.fourcc()
    if cc in ("IxF2", "IxFI", "IxFl"):
        d, ntotal, _ = _header(r)
        xb = r.vec(np.float32)                          # codes: ntotal * d floats (size is stored in 4-byte units)
        if xb.size != ntotal * d:
            raise IndexFormatError("flat index holds %d floats for ntotal=%d, d=%d" % (xb.size, ntotal, d))
        return xb.reshape(ntotal, d), None
    if cc == "IxMp" or cc == "IxM2":                    # IndexIDMap(2): sub-index + id_map
        _header(r)
        v, ids = _read(r)
        idmap = r.vec(np.int64)
        return v, (idmap if ids is None else idmap[ids])
    if cc == "IwFl":
        d, ntotal, _ = _header(r)
        nlist = r.u64(); r.u64()                        # nprobe
        _read(r)                                        # coarse quantizer (IndexFlat of the nlist centroids): not needed
        dm_type = r.u8()                                # direct map
        r.vec(np.int64)
        if dm_type == 2:
            n = r.u64(); r.raw(n * 16)
        # write_index for IndexIVFFlat emits write_ivf_header + write_InvertedLists only: unlike IwSq / IwPQ there is NO code_size
        # field here (read_index sets code_size = d * sizeof(float)); the value inside the "ilar" header is checked against it
        code_size = 4 * d
        il = r.fourcc()
        if il == "il00":
            return np.zeros((0, d), np.float32), np.zeros(0, np.int64)
        if il != "ilar":
            raise IndexFormatError("unsupported inverted-list container %r" % il)
        nl2, cs2 = r.u64(), r.u64()
        if nl2 != nlist or cs2 != code_size:
            raise IndexFormatError("inverted lists (nlist %d, code size %d) disagree with the IVF header (nlist %d, 4 * d = %d)" % (nl2, cs2, nlist, code_size))
        lt = r.fourcc()
        if lt == "full":
            sizes = r.vec(np.uint64).astype(np.int64)
        elif lt == "sprs":
            pairs = r.vec(np.uint64).astype(np.int64).reshape(-1, 2)
            sizes = np.zeros(nlist, np.int64); sizes[pairs[:, 0]] = pairs[:, 1]
        else:
            raise IndexFormatError("unsupported list-size encoding %r" % lt)
        if sizes.size != nlist or int(sizes.sum()) != ntotal:
            raise IndexFormatError("inverted-list sizes do not add up to ntotal")
        vs: List[np.ndarray] = []; ids: List[np.ndarray] = []
        for n in sizes:
            n = int(n)
            if n:
                vs.append(np.frombuffer(r.raw(n * code_size), dtype="<f4").astype(np.float32).reshape(n, d))
                ids.append(np.frombuffer(r.raw(n * 8), dtype="<i8").astype(np.int64))
        if not vs:
            return np.zeros((0, d), np.float32), np.zeros(0, np.int64)
        return np.concatenate(vs), np.concatenate(ids)
    raise IndexFormatError("unsupported Faiss index type %r (supported: IndexFlat, IndexIVFFlat, IndexIDMap)" % cc)


def read_index(path: str) -> np.ndarray:
    """(ntotal, d) float32, row i = the vector stored under id i (upstream's `big_npy = index.reconstruct_n(0, index.ntotal)`)."""
    with open(path, "rb") as f:
        v, ids = _read(_R(f))
    if ids is None:
        return np.ascontiguousarray(v)
    if ids.size != v.shape[0] or (ids.size and (ids.min() < 0 or ids.max() >= ids.size or np.unique(ids).size != ids.size)):
        raise IndexFormatError("ids are not a permutation of 0..ntotal-1: cannot reconstruct in id order")
    out = np.empty_like(v)
    out[ids] = v
    return out


# ----------------------------------------------------------------------------- writers (tests / export)
def _w_header(f: BinaryIO, d: int, ntotal: int, metric: int = 1):
    f.write(struct.pack("<iqqqBi", d, ntotal, 1 << 20, 1 << 20, 1, metric))


def write_flat(path_or_file, vectors: np.ndarray):
    v = np.ascontiguousarray(vectors, dtype="<f4")
    f = open(path_or_file, "wb") if isinstance(path_or_file, str) else path_or_file
    f.write(b"IxF2"); _w_header(f, v.shape[1], v.shape[0])
    f.write(struct.pack("<Q", v.size)); f.write(v.tobytes())
    if isinstance(path_or_file, str):
        f.close()


def write_ivf_flat(path: str, vectors: np.ndarray, centroids: np.ndarray, sparse_sizes: bool = False):
    """IndexIVFFlat with the given coarse centroids; vectors are assigned to their nearest centroid, ids = row numbers."""
    v = np.ascontiguousarray(vectors, dtype=np.float32)
    c = np.ascontiguousarray(centroids, dtype=np.float32)
    n, d = v.shape
    assign = np.argmin(((v[:, None, :] - c[None]) ** 2).sum(-1), axis=1) if n else np.zeros(0, np.int64)
    with open(path, "wb") as f:
        f.write(b"IwFl"); _w_header(f, d, n)
        f.write(struct.pack("<QQ", c.shape[0], 1))
        write_flat(f, c)
        f.write(struct.pack("<B", 0)); f.write(struct.pack("<Q", 0))            # direct map: NoMap, empty array
        f.write(b"ilar"); f.write(struct.pack("<QQ", c.shape[0], 4 * d))
        sizes = np.bincount(assign, minlength=c.shape[0]).astype(np.uint64)
        if sparse_sizes:
            nz = np.nonzero(sizes)[0]
            f.write(b"sprs"); f.write(struct.pack("<Q", 2 * nz.size))
            f.write(np.stack([nz.astype(np.uint64), sizes[nz]], 1).astype("<u8").tobytes())
        else:
            f.write(b"full"); f.write(struct.pack("<Q", sizes.size)); f.write(sizes.astype("<u8").tobytes())
        for l in range(c.shape[0]):
            ids = np.nonzero(assign == l)[0].astype("<i8")
            if ids.size:
                f.write(v[ids].astype("<f4").tobytes()); f.write(ids.tobytes())
