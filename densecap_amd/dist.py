"""Image sharding + the single end-of-run gather of the multi-GPU path.

The path shards by image with no data-path collective; the only exchange is one gather of
fixed-capacity padded records {K; boxes[P][4]; scores[P]; tokens[P][T]} to rank 0
(torch.distributed: backend "nccl" = RCCL over xGMI on the GPU box, "gloo" in CPU tests).
"""
from __future__ import annotations

import numpy as np


def shard_range(n_images, world, rank):
    """Contiguous block partition: image i -> rank floor(i / ceil(n/world))."""
    per = (n_images + world - 1) // world
    lo = min(rank * per, n_images)
    return lo, min(lo + per, n_images)


def pack_records(results, P, T):
    """results: list of (boxes (K,4), scores (K,), tokens (K,T)) -> (rec (n,P,5+T) f32, cnt (n,) i32).
    Token ids (<= V+1 ~ 1e4) are exactly representable in fp32."""
    n = len(results)
    rec = np.zeros((n, P, 5 + T), np.float32)
    cnt = np.zeros((n,), np.int32)
    for i, (b, s, t) in enumerate(results):
        k = len(b)
        cnt[i] = k
        rec[i, :k, :4] = b
        rec[i, :k, 4] = np.asarray(s).reshape(-1)
        rec[i, :k, 5:] = t
    return rec, cnt


def unpack_records(rec, cnt):
    out = []
    for i in range(rec.shape[0]):
        k = int(cnt[i])
        out.append((rec[i, :k, :4].copy(), rec[i, :k, 4].copy(), rec[i, :k, 5:].astype(np.int32)))
    return out


def gather_records(dist, rec, cnt, rank, world, device=None):
    """One gather of every rank's records on rank 0.  Returns list (per rank) of unpacked results on
    rank 0, None elsewhere.  `device`: torch device for the collective's tensors (cuda for RCCL)."""
    import torch
    rec_t = torch.from_numpy(rec)
    cnt_t = torch.from_numpy(cnt)
    if device is not None:
        rec_t = rec_t.to(device)
        cnt_t = cnt_t.to(device)
    if rank == 0:
        recs = [torch.empty_like(rec_t) for _ in range(world)]
        cnts = [torch.empty_like(cnt_t) for _ in range(world)]
    else:
        recs = cnts = None
    dist.gather(rec_t, recs, dst=0)
    dist.gather(cnt_t, cnts, dst=0)
    if rank != 0:
        return None
    return [unpack_records(r.cpu().numpy(), c.cpu().numpy()) for r, c in zip(recs, cnts)]
