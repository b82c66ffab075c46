"""Greedy decode alone (LanguageModel:sample through dc_op_lm_sample) at the path's row counts: wall time per call on one
stream, and the MFMA family's share.   usage (GPU box): python tools/decode_bench.py [reps] [rows ...]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from densecap_amd import DenseCapModel
from densecap_amd._lib import check
from densecap_amd.weights import make_synthetic_weights

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
rows = [int(a) for a in sys.argv[2:] if a.lstrip("-").isdigit()] or [1000, 300, 50]
W = make_synthetic_weights(seed=1234)
m = DenseCapModel(W, device=0)
ctx = m.ctx
rng = np.random.default_rng(0)
beam = 0
for a in sys.argv:
    if a.startswith("--stages="):         # LDS ring depth of the 128x64-tile kernel (2 = three workgroups per CU)
        check(ctx.h, ctx.lib.dc_debug_set(ctx.h, b"v2_stages", int(a.split("=")[1])))
    if a.startswith("--force-cfg="):      # measurement hook: 2 = 128x64 tiles, 3 = 64x64 tiles for the step GEMM
        check(ctx.h, ctx.lib.dc_debug_set(ctx.h, b"force_cfg", int(a.split("=")[1])))
    if a.startswith("--beam="):          # LM:beamsearch with that many beams instead of the greedy LM:sample
        beam = int(a.split("=")[1])
        m.setBeamSize(beam)
if beam:
    print("beam search, %d beams (all proposals advance together: rows = proposals x beams)" % beam)
for n in rows:
    codes = np.maximum(rng.standard_normal((n, 4096)), 0).astype(np.float32)
    cd = ctx.to_device(codes); td = ctx.empty((n, 15), np.int32)
    for _ in range(3):
        check(ctx.h, ctx.lib.dc_op_lm_sample(ctx.h, cd.ptr, n, td.ptr), "dc_op_lm_sample")
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter()
        check(ctx.h, ctx.lib.dc_op_lm_sample(ctx.h, cd.ptr, n, td.ptr), "dc_op_lm_sample")
        ts.append(time.perf_counter() - t0)
    m.mfma_profile(reset=1)
    check(ctx.h, ctx.lib.dc_op_lm_sample(ctx.h, cd.ptr, n, td.ptr), "dc_op_lm_sample")
    p = m.mfma_profile(reset=-1)
    ts = np.array(ts) * 1e3
    print("rows=%d  decode call ms: median %.3f min %.3f (incl. scratch malloc/free + sync)   MFMA launches %d, %.3f ms, %.1f TF  tokens[0]=%s" %
          (n, np.median(ts), ts.min(), p["launches"], p["ms"], p["flops"] / max(p["ms"], 1e-9) / 1e9, td.numpy()[0, :6].tolist()))
    cd.free(); td.free()
