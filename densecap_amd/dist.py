"""Image sharding + the single end-of-run gather of the multi-GPU path.

The reference is single-device (densecap/utils.lua:22-36; run_model.lua:160-180 loops over images on it).  The
path shards by image with no data-path collective; the only exchange is ONE gather of fixed-capacity typed records

    {int32 K, T, capacity, 0;  float32 boxes[P][4];  float32 scores[P];  int32 tokens[P][T]}

per image to rank 0.  Two carriers of the same record:
  * `Comm` -- the C ABI's dc_gather_results (densecap_amd/csrc/comm.hip): RCCL ncclSend/ncclRecv over xGMI, what a
    LuaJIT host uses too (lua/DenseCapModelHIP.lua);
  * `gather_records` -- one torch.distributed gather of the packed bytes (backend "nccl" = RCCL on the GPU box, "gloo"
    in the CPU tests and when two ranks share one GPU).
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib
from ._lib import DcResult

HEADER_BYTES = 16


def shard_range(n_images, world, rank):
    """Contiguous block partition: image i -> rank floor(i / ceil(n/world))."""
    per = (n_images + world - 1) // world
    lo = min(rank * per, n_images)
    return lo, min(lo + per, n_images)


def record_bytes(P, T):
    return HEADER_BYTES + P * (16 + 4 + 4 * T)


def pack_records(results, P, T):
    """results: list of (boxes (K,4) f32, scores (K,) f32, tokens (K,T) i32) -> uint8 (n, record_bytes(P,T)).
    Typed fields at fixed offsets; tokens stay int32 (same layout as comm.hip)."""
    n = len(results)
    rb = record_bytes(P, T)
    buf = np.zeros((n, rb), np.uint8)
    for i, (b, s, t) in enumerate(results):
        k = len(b)
        if k > P:
            raise ValueError("record %d has %d rows, capacity is %d" % (i, k, P))
        buf[i, :HEADER_BYTES].view(np.int32)[:] = (k, T, P, 0)
        buf[i, 16:16 + 16 * k] = np.ascontiguousarray(b, np.float32).reshape(-1).view(np.uint8)
        o = 16 + 16 * P
        buf[i, o:o + 4 * k] = np.ascontiguousarray(s, np.float32).reshape(-1).view(np.uint8)
        o = 16 + 20 * P
        buf[i, o:o + 4 * T * k] = np.ascontiguousarray(t, np.int32).reshape(-1).view(np.uint8)
    return buf


def unpack_records(buf):
    out = []
    for row in np.ascontiguousarray(buf, np.uint8):
        k, T, P, _ = (int(v) for v in row[:HEADER_BYTES].view(np.int32))
        if not (0 <= k <= P) or len(row) != record_bytes(P, T):
            raise ValueError("corrupt record header K=%d T=%d P=%d" % (k, T, P))
        b = row[16:16 + 16 * k].view(np.float32).reshape(k, 4).copy()
        o = 16 + 16 * P
        s = row[o:o + 4 * k].view(np.float32).copy()
        o = 16 + 20 * P
        t = row[o:o + 4 * T * k].view(np.int32).reshape(k, T).copy()
        out.append((b, s, t))
    return out


def gather_records(dist, results, P, T, rank, world, device=None):
    """ONE torch.distributed gather of every rank's packed records on rank 0.  Returns a list (per rank) of lists of
    (boxes, scores, tokens) on rank 0, None elsewhere.  `device`: torch device for the collective's tensor (cuda for
    RCCL; None = CPU for gloo)."""
    import torch
    rec = torch.from_numpy(pack_records(results, P, T))
    if device is not None:
        rec = rec.to(device)
    recs = [torch.empty_like(rec) for _ in range(world)] if rank == 0 else None
    dist.gather(rec, recs, dst=0)
    if rank != 0:
        return None
    return [unpack_records(r.cpu().numpy()) for r in recs]


class Comm:
    """dc_comm of the C ABI: RCCL communicator bound to a ctx; gather() is the path's single collective."""

    SELF_TRANSPORT = 1          # DC_COMM_SELF_TRANSPORT (include/densecap.h)

    def __init__(self, ctx, rank=0, world=1, unique_id=None, self_transport=False):
        """self_transport (world == 1 only): build the carrier for the single rank too, so that gather() travels through
        ncclSend / ncclRecv to itself -- the multi-GPU code path, executable on one GPU."""
        self.lib = ctx.lib
        self.ctx = ctx
        self.rank, self.world = int(rank), int(world)
        if self.world > 1 and (unique_id is None or len(unique_id) != 128):
            raise ValueError("world > 1 needs the 128-byte id made by Comm.unique_id() on rank 0")
        h = C.c_void_p()
        idbuf = C.create_string_buffer(bytes(unique_id), 128) if unique_id is not None else None
        rc = self.lib.dc_comm_create_ex(C.byref(h), ctx.h, idbuf, self.rank, self.world,
                                        self.SELF_TRANSPORT if self_transport else 0)
        if rc < 0:
            msg = self.lib.dc_comm_last_error(None)
            raise _lib.DenseCapError("dc_comm_create failed (%d): %s" % (rc, msg.decode() if msg else "?"))
        self.h = h

    @property
    def transport(self):
        t = self.lib.dc_comm_transport(self.h)
        return t.decode() if t else ""

    @staticmethod
    def unique_id(lib=None):
        lib = lib or _lib.lib()
        buf = C.create_string_buffer(128)
        rc = lib.dc_comm_unique_id(buf)
        if rc < 0:
            msg = lib.dc_comm_last_error(None)
            raise _lib.DenseCapError("dc_comm_unique_id failed (%d): %s" % (rc, msg.decode() if msg else "?"))
        return buf.raw

    def close(self):
        if getattr(self, "h", None):
            self.lib.dc_comm_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @staticmethod
    def _structs(n, P, T):
        arr = (DcResult * n)()
        keep = []
        for i in range(n):
            b = np.zeros((P, 4), np.float32); s = np.zeros((P,), np.float32); t = np.zeros((P, T), np.int32)
            arr[i].capacity = P
            arr[i].T = T
            arr[i].boxes = b.ctypes.data_as(_lib.c_float_p)
            arr[i].scores = s.ctypes.data_as(_lib.c_float_p)
            arr[i].tokens = t.ctypes.data_as(_lib.c_int32_p)
            keep.append((b, s, t))
        return arr, keep

    def gather(self, results, P, T):
        """results: this rank's list of (boxes, scores, tokens).  Rank 0 returns [rank][image] -> (boxes, scores,
        tokens); other ranks None.  Every rank must pass the same number of images, P and T."""
        n = len(results)
        loc, keep = self._structs(n, P, T)
        for i, (b, s, t) in enumerate(results):
            k = len(b)
            loc[i].K = k
            keep[i][0][:k] = b; keep[i][1][:k] = np.asarray(s).reshape(-1); keep[i][2][:k] = t
        if self.rank == 0:
            allr, akeep = self._structs(n * self.world, P, T)
        else:
            allr, akeep = None, None
        rc = self.lib.dc_gather_results(self.h, loc, n, allr)
        if rc < 0:
            msg = self.lib.dc_comm_last_error(self.h)
            raise _lib.DenseCapError("dc_gather_results failed (%d): %s" % (rc, msg.decode() if msg else "?"))
        if self.rank != 0:
            return None
        out = []
        for r in range(self.world):
            shard = []
            for i in range(n):
                j = r * n + i
                k = allr[j].K
                b, s, t = akeep[j]
                shard.append((b[:k].copy(), s[:k].copy(), t[:k].copy()))
            out.append(shard)
        return out
