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


def test_bench_line_has_roofline_and_cpu_baseline():
    """Default single-GPU invocation shape (small workload so the CPU leg is quick): the contract's `roofline` and
    `cpu_baseline` objects, the per-stage fractions and the repeats are all in the one line."""
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "4", "--warmup", "1", "--repeats", "2",
                        "--height", "224", "--width", "288", "--proposals", "100"], capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    r = d["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic", "kernel", "trunk_frac", "fc_frac", "decode_frac",
              "serial_ms_per_image", "measured_on", "traffic_from_profile"):
        assert k in r, k
    assert r["bound"] == "mfma" and r["unit"] == "TFLOP/s" and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9
    c = d["cpu_baseline"]
    assert c["kind"] == "port" and c["unit"] == "images/s" and c["value"] > 0 and c["cores"] >= 1 and "sample" in c
    assert d["n_gpus"] == 1 and len(d["repeats"]["images_per_s"]) == 2 and d["value"] > c["value"]


@pytest.mark.skipif("_ngpus() < 2")
def test_bench_world2_rccl_gather():
    """Two GPUs visible: the default carrier -- dc_gather_results over RCCL send/recv."""
    d = _run_bench(2, [])
    assert d["n_gpus"] == 2 and d["config"]["gather"].startswith("dc_gather_results")
    assert d["config"]["total_output_boxes"] > 0
