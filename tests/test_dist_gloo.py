"""world_size-2 gloo test of the N>1 path's host logic: contiguous image sharding and the single
end-of-run gather of padded (boxes, scores, tokens) records (bench.py uses the same functions over
RCCL).  The per-image compute is replaced by a deterministic stand-in: no GPU here."""
import os
import socket

import numpy as np


def _fake_forward(img_id, P, T):
    rng = np.random.default_rng(img_id)
    k = int(rng.integers(0, P + 1))
    return (rng.standard_normal((k, 4)).astype(np.float32), np.sort(rng.standard_normal(k).astype(np.float32))[::-1].copy(),
            rng.integers(1, 10499, (k, T)).astype(np.int32))


def _worker(rank, world, port, n_images, P, T, q):
    import torch.distributed as dist
    from densecap_amd import dist as D
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    lo, hi = D.shard_range(n_images, world, rank)
    results = [_fake_forward(i, P, T) for i in range(lo, hi)]
    per = (n_images + world - 1) // world
    while len(results) < per:                      # pad the last shard so gather shapes agree
        results.append((np.zeros((0, 4), np.float32), np.zeros((0,), np.float32), np.zeros((0, T), np.int32)))
    rec, cnt = D.pack_records(results, P, T)
    dist.barrier()
    out = D.gather_records(dist, rec, cnt, rank, world)
    if rank == 0:
        flat = [r for shard in out for r in shard][:n_images]
        ok = True
        for i, (b, s, t) in enumerate(flat):
            eb, es, et = _fake_forward(i, P, T)
            ok &= np.array_equal(b, eb) and np.array_equal(s, es) and np.array_equal(t, et)
        q.put(bool(ok))
    dist.destroy_process_group()


def test_shard_range_covers_all_images():
    from densecap_amd.dist import shard_range
    for n, w in [(512, 8), (7, 2), (3, 4), (1, 1), (64, 8)]:
        seen = []
        for r in range(w):
            lo, hi = shard_range(n, w, r)
            seen += list(range(lo, hi))
        assert seen == list(range(n))
    assert shard_range(512, 8, 3) == (192, 256)    # BASELINE config 4: 64 images per GPU


def test_gather_world2_gloo():
    import torch.multiprocessing as mp
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, 7, 20, 15, q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    assert q.get(timeout=5) is True
