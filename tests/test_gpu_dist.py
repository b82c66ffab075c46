"""Multi-GPU readiness on a one-GPU box: the C ABI's communicator at world = 1, bench.py's real world = 2 branch with
two ranks sharing GPU 0 (gloo carrier: RCCL refuses two ranks on one device), two contexts in one process, and -- only
where two GPUs are visible -- dc_gather_results over RCCL between two processes."""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _ngpus():
    import torch
    return torch.cuda.device_count()


def test_comm_world1_gather_is_identity():
    from densecap_amd import dist as D
    from densecap_amd.ops import Context
    ctx = Context(0)
    comm = D.Comm(ctx, 0, 1)
    rng = np.random.default_rng(0)
    res = []
    for i in range(3):
        k = int(rng.integers(0, 20))
        res.append((rng.standard_normal((k, 4)).astype(np.float32), rng.standard_normal(k).astype(np.float32),
                    rng.integers(1, 999, (k, 15)).astype(np.int32)))
    out = comm.gather(res, 20, 15)
    assert len(out) == 1 and len(out[0]) == 3
    for (b, s, t), (b2, s2, t2) in zip(res, out[0]):
        np.testing.assert_array_equal(b, b2); np.testing.assert_array_equal(s, s2); np.testing.assert_array_equal(t, t2)
    with pytest.raises(Exception):
        comm.gather([(np.zeros((30, 4), np.float32), np.zeros(30, np.float32), np.zeros((30, 15), np.int32))], 20, 15)
    comm.close()
    ctx.close()


def _random_records(rng, n, P, T):
    out = []
    for _ in range(n):
        k = int(rng.integers(0, P + 1))
        out.append((rng.standard_normal((k, 4)).astype(np.float32), rng.standard_normal(k).astype(np.float32),
                    rng.integers(1, 9999, (k, T)).astype(np.int32)))
    return out


def _assert_same_records(a, b):
    assert len(a) == len(b)
    for (x0, x1, x2), (y0, y1, y2) in zip(a, b):
        np.testing.assert_array_equal(x0, y0); np.testing.assert_array_equal(x1, y1); np.testing.assert_array_equal(x2, y2)


@pytest.mark.parametrize("carrier", ["rccl", "loopback"])
def test_comm_self_transport_carries_records_byte_for_byte(carrier):
    """Round-4 verdict, item 1: the RCCL carrier had never executed.  DC_COMM_SELF_TRANSPORT builds it for ONE rank --
    ncclCommInitRank(world = 1) -- and every gather goes through device staging and one ncclGroupStart / ncclRecv /
    ncclSend / ncclGroupEnd with rank 0 as its own peer: the calls of the multi-GPU gather.  The records that come back
    are the ones that travelled (the host staging is overwritten before the receive lands), compared byte for byte; sizes
    grow between gathers (buffers are re-made), and bad arguments still fail."""
    from densecap_amd import dist as D
    from densecap_amd.ops import Context
    ctx = Context(0)
    ident = b"dc-loopback:self".ljust(128, b"\0") if carrier == "loopback" else None
    comm = D.Comm(ctx, 0, 1, ident, self_transport=True)
    try:
        assert comm.transport == carrier + ", self"
        rng = np.random.default_rng(5)
        for n, P, T in ((1, 7, 3), (3, 20, 15), (64, 1000, 15), (2, 20, 15)):
            res = _random_records(rng, n, P, T)
            out = comm.gather(res, P, T)
            assert len(out) == 1
            _assert_same_records(res, out[0])
        with pytest.raises(Exception):
            comm.gather([(np.zeros((30, 4), np.float32), np.zeros(30, np.float32), np.zeros((30, 15), np.int32))], 20, 15)
        out = comm.gather(res, 20, 15)                      # still usable after a refused call
        _assert_same_records(res, out[0])
    finally:
        comm.close()
        ctx.close()


def test_env_switch_forces_the_rccl_carrier_at_world1():
    """DC_COMM_FORCE_RCCL=1: an unmodified host's dc_comm_create(world = 1) builds the carrier too."""
    from densecap_amd import dist as D
    from densecap_amd.ops import Context
    ctx = Context(0)
    plain = D.Comm(ctx, 0, 1)
    assert plain.transport == "host copy"
    plain.close()
    os.environ["DC_COMM_FORCE_RCCL"] = "1"
    try:
        import ctypes as C
        h = C.c_void_p()
        assert ctx.lib.dc_comm_create(C.byref(h), ctx.h, None, 0, 1) == 0
        assert ctx.lib.dc_comm_transport(h) == b"rccl, self"
        ctx.lib.dc_comm_destroy(h)
    finally:
        del os.environ["DC_COMM_FORCE_RCCL"]
    ctx.close()


def test_model_results_through_the_rccl_self_gather():
    """A real forward's results through the one-rank RCCL gather equal the results themselves."""
    from densecap_amd import DenseCapModel, dist as D
    from densecap_amd.weights import make_synthetic_image, make_synthetic_weights
    W = make_synthetic_weights(seed=1234, vocab_size=300, seq_length=6)
    m = DenseCapModel(W, device=0)
    try:
        m.setTestArgs(rpn_nms_thresh=0.7, final_nms_thresh=0.3, num_proposals=100)
        res = m.forward_batch(np.stack([make_synthetic_image(224, 288, i) for i in range(4)]))
        comm = D.Comm(m.ctx, 0, 1, None, self_transport=True)
        out = comm.gather(res, 100, 6)
        comm.close()
        _assert_same_records(res, out[0])
        assert sum(len(b) for b, _, _ in res) > 0
    finally:
        m.ctx.close()


def test_rccl_unique_id_is_available():
    """librccl is dlopen()ed on demand; the id is the 128-byte ncclUniqueId rank 0 hands to its peers."""
    from densecap_amd import dist as D
    a, b = D.Comm.unique_id(), D.Comm.unique_id()
    assert len(a) == len(b) == 128 and a != b


def test_two_contexts_in_one_process_give_identical_results():
    """N contexts in one process (include/densecap.h threading note): the second ctx -- same device here, device
    1..7 on a node -- must find every kernel attribute it needs (the >64 KiB dynamic-LDS limit is per device and is
    tracked per (device, kernel)), and both give bit-identical results."""
    from densecap_amd import DenseCapModel
    from densecap_amd.weights import make_synthetic_image, make_synthetic_weights
    W = make_synthetic_weights(seed=1234, vocab_size=300, seq_length=6)
    img = make_synthetic_image(224, 288, 2)
    a = DenseCapModel(W, device=0)
    b = DenseCapModel(W, device=_ngpus() - 1)           # another device when there is one
    try:
        for m in (a, b):
            m.setTestArgs(rpn_nms_thresh=0.7, final_nms_thresh=0.3, num_proposals=100)
        ra, rb = a.forward_raw(img), b.forward_raw(img)
        for x, y in zip(ra, rb):
            np.testing.assert_array_equal(x, y)
        assert len(ra[0]) > 0
    finally:
        a.ctx.close(); b.ctx.close()


def _run_bench(nproc, extra):
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    p = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(nproc),
                        "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "bench.py"),
                        "--gpus", str(nproc), "--steps", "4", "--warmup", "1", "--repeats", "2", "--height", "224",
                        "--width", "288", "--proposals", "100"] + extra, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stdout[-3000:] + p.stderr[-3000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    return json.loads(lines[0])


def test_bench_world2_branch_on_one_gpu():
    """bench.py launched exactly as the driver launches it for N = 2, both ranks on GPU 0 (gloo carrier): the HIP model
    runs in both ranks, rank 0 receives both shards through the one gather and prints the line."""
    d = _run_bench(2, ["--dist-backend", "gloo"])
    assert d["n_gpus"] == 2 and d["data"] == "synthetic" and d["config"]["gather"].startswith("torch.distributed.gather")
    assert d["config"]["total_output_boxes"] > 0 and d["value"] > 0
    assert "roofline" in d and d["roofline"]["frac"] > 0


def test_bench_one_rank_under_torchrun_gathers_over_rccl():
    """bench.py as the driver launches it for N > 1, with ONE rank: torch's own RCCL process group (backend nccl) and the
    librccl that libdensecap_hip.so dlopen()s coexist in one process, and the gather of every timed region runs over the
    one-rank RCCL communicator."""
    d = _run_bench(1, ["--dist-backend", "nccl", "--gather", "abi", "--no-cpu-baseline", "--sustain-seconds", "0"])
    assert d["n_gpus"] == 1 and d["config"]["gather"] == "dc_gather_results (rccl, self)", d["config"]["gather"]
    assert d["config"]["total_output_boxes"] > 0 and d["value"] > 0


def test_bench_line_has_roofline_and_cpu_baseline():
    """Default single-GPU invocation shape (small workload so the CPU leg is quick): the contract's `roofline` and
    `cpu_baseline` objects, the per-stage fractions and the repeats are all in the one line."""
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "4", "--warmup", "1", "--repeats", "2",
                        "--height", "224", "--width", "288", "--proposals", "100"], capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    # the JSON line is the LAST line of stdout: the banner RCCL prints through C stdio (the gather runs over a one-rank RCCL
    # communicator now) is flushed before it, nothing native reaches stdout after it
    assert p.stdout.strip().splitlines()[-1] == lines[0], p.stdout[-600:]
    d = json.loads(lines[0])
    r = d["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "frac_serial", "achieved_serial", "traffic", "kernel", "trunk_frac",
              "fc_frac", "decode_frac", "serial_ms_per_image", "measured_on", "traffic_from_profile"):
        assert k in r, k
    assert r["bound"] == "mfma" and r["unit"] == "TFLOP/s" and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9
    # HBM traffic per MFMA launch measured by the run itself (two rocprofv3 --pmc child passes), not only quoted from profiles/
    live = r["traffic_live"]
    assert "error" not in live, live
    assert live["launches_counted"] > 30 and live["hbm_bytes_per_launch"] > 0 and r["traffic"] == live["hbm_bytes_per_launch"]
    assert 0.5 < r["traffic_over_algorithmic"] < 20, r["traffic_over_algorithmic"]
    assert 0.05 < live["mfma_util_family"] < 1.0 and 0.05 < live["mfma_util_conv_kernels"] < 1.0, live
    c = d["cpu_baseline"]
    assert c["kind"] == "port" and c["unit"] == "images/s" and c["value"] > 0 and c["cores"] >= 1 and "sample" in c
    assert d["n_gpus"] == 1 and len(d["repeats"]["images_per_s"]) == 2 and d["value"] > c["value"]
    assert d["value_split_bf16"] > 0 and d["roofline_split_bf16"]["peak"] > 400 and 0 < d["roofline_split_bf16"]["frac"] < 1
    assert abs(r["frac"] - r["frac_timed_region"]) < 1e-12 and len(d["warm_up_regions_images_per_s"]) >= 2
    # one GPU, no launcher: the end-of-region gather ran over the one-rank RCCL communicator
    assert d["config"]["gather"] == "dc_gather_results (rccl, self)", d["config"]["gather"]


@pytest.mark.skipif("_ngpus() < 2")
def test_bench_world2_rccl_gather():
    """Two GPUs visible: the default carrier -- dc_gather_results over RCCL send/recv."""
    d = _run_bench(2, [])
    assert d["n_gpus"] == 2 and d["config"]["gather"].startswith("dc_gather_results")
    assert d["config"]["total_output_boxes"] > 0


def _loopback_gather(world, n_local, P, T, shapes=None, seed=0):
    """`world` dc_comm objects on an in-process loopback hub (id "dc-loopback:<name>"), one thread per rank, all on
    GPU 0: drives dc_gather_results' whole control flow -- shape handshake, device staging, per-peer offsets block*peer,
    unpack order r*n_local+i -- without RCCL (which refuses two ranks on one device)."""
    import threading
    from densecap_amd import dist as D
    from densecap_amd.ops import Context
    ident = ("dc-loopback:test-%d-%d-%d" % (world, n_local, seed)).encode().ljust(128, b"\0")
    rng = np.random.default_rng(seed)
    data = []
    for r in range(world):
        nl, Pr, Tr = shapes[r] if shapes else (n_local, P, T)
        shard = []
        for i in range(nl):
            k = int(rng.integers(0, Pr + 1))
            shard.append((rng.standard_normal((k, 4)).astype(np.float32), rng.standard_normal(k).astype(np.float32),
                          rng.integers(1, 9999, (k, Tr)).astype(np.int32)))
        data.append(shard)
    ctxs = [Context(0) for _ in range(world)]
    out, errs = [None] * world, [None] * world

    def run(r):
        try:
            comm = D.Comm(ctxs[r], r, world, ident)
            try:
                _, Pr, Tr = shapes[r] if shapes else (n_local, P, T)
                out[r] = comm.gather(data[r], Pr, Tr)
            finally:
                comm.close()
        except Exception as e:       # noqa: BLE001 -- reported per rank below
            errs[r] = e

    th = [threading.Thread(target=run, args=(r,)) for r in range(world)]
    for t in th: t.start()
    for t in th: t.join(120)
    assert not any(t.is_alive() for t in th), "a rank hung in dc_gather_results"
    for c in ctxs: c.close()
    return data, out, errs


@pytest.mark.parametrize("world,n_local", [(2, 3), (4, 2), (8, 4)])
def test_gather_results_loopback_world_gt1(world, n_local):
    data, out, errs = _loopback_gather(world, n_local, P=37, T=15, seed=world)
    assert errs == [None] * world, errs
    assert all(o is None for o in out[1:]) and len(out[0]) == world
    for r in range(world):
        assert len(out[0][r]) == n_local
        for (b, s, t), (b2, s2, t2) in zip(data[r], out[0][r]):
            np.testing.assert_array_equal(b, b2); np.testing.assert_array_equal(s, s2); np.testing.assert_array_equal(t, t2)


def test_gather_results_refuses_unequal_shards_without_hanging():
    """dist.shard_range gives uneven shards when n % world != 0: the shape handshake must turn that into an error on
    EVERY rank (under RCCL, mismatched byte counts would hang or corrupt)."""
    shapes = [(3, 20, 15), (3, 20, 15), (2, 20, 15), (3, 20, 15)]           # rank 2 brings one image less
    _, out, errs = _loopback_gather(4, 3, 20, 15, shapes=shapes, seed=9)
    assert all(e is not None for e in errs), errs
    assert "rank 2" in str(errs[0]) and "disagree" in str(errs[1])
    shapes = [(2, 20, 15), (2, 24, 15)]                                      # another capacity
    _, out, errs = _loopback_gather(2, 2, 20, 15, shapes=shapes, seed=10)
    assert all(e is not None for e in errs), errs


def test_gather_results_local_validation_failure_reaches_every_rank():
    """Advisor finding (round 3): a rank whose own arguments are bad used to return BEFORE the handshake, leaving the other
    ranks blocked in it.  The failure now travels in the handshake: all ranks return an error together, nobody hangs --
    whether the bad rank is a peer or rank 0 itself."""
    for shapes, bad in (([(2, 20, 15), (0, 20, 15)], 1), ([(0, 20, 15), (2, 20, 15)], 0),
                        ([(2, 20, 15), (2, 20, 15), (0, 20, 15), (2, 20, 15)], 2)):
        _, out, errs = _loopback_gather(len(shapes), 2, 20, 15, shapes=shapes, seed=30 + bad)
        assert all(e is not None for e in errs), errs
        assert "bad arguments" in str(errs[bad])
        for r, e in enumerate(errs):
            if r != bad:
                assert "rejected its arguments" in str(e), (r, str(e))
