"""Micro-benchmark of the MFMA contraction kernel on hot-path shapes (run on the GPU box).
usage: python tools/gemm_bench.py [reps]"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctypes as C
import numpy as np
from densecap_amd.ops import Context
from densecap_amd._lib import check

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 5
CUSTOM = [tuple(int(v) for v in a.split(",")) for a in sys.argv[2:] if a.count(",") == 2]     # extra "M,N,K" shapes only
ctx = Context(0)
lib = ctx.lib
if "--serial" in sys.argv:          # single-image mode: tail plans (K-split last round) are active, as in `bench.py --lanes 1`
    check(ctx.h, lib.dc_set_lanes(ctx.h, 1))
for a in sys.argv:
    if a.startswith("--math-mode="):      # 1 = split-bf16 contraction mode (dc_set_math_mode)
        check(ctx.h, lib.dc_set_math_mode(ctx.h, int(a.split("=")[1])))
    if a.startswith("--force-cfg="):      # dc_debug_set force_cfg (tile configuration of plain launches)
        check(ctx.h, lib.dc_debug_set(ctx.h, b"force_cfg", int(a.split("=")[1])))
    if a.startswith("--stages="):         # LDS ring depth of the 128x64-tile kernel (2 = three workgroups per CU)
        check(ctx.h, ctx.lib.dc_debug_set(ctx.h, b"v2_stages", int(a.split("=")[1])))
    if a.startswith("--tail-mode="):      # 0 stream-K (default), 1 K-split tail plan, 2 whole tiles
        check(ctx.h, lib.dc_debug_set(ctx.h, b"tail_mode", int(a.split("=")[1])))
print("mode:", "serial (dc_set_lanes(1): partial last rounds are shared along K)" if "--serial" in sys.argv else "multi-lane planning (whole tiles)",
      [a for a in sys.argv if a.startswith("--tail-mode")])

def prof(reset):
    l = C.c_int64(0); ms = C.c_double(0); fl = C.c_double(0)
    lib.dc_mfma_profile(ctx.h, reset, C.byref(l), C.byref(ms), C.byref(fl))
    return l.value, ms.value, fl.value

rng = np.random.default_rng(0)
def dev_rand(n):
    return ctx.to_device(rng.standard_normal(n).astype(np.float32))

LIN = [("fc6", 1000, 4096, 25088), ("fc7", 1000, 4096, 4096), ("lm_enc", 1000, 512, 4096),
       ("gates", 1000, 2048, 512), ("vocab", 1000, 10498, 512), ("rpn_heads", 1710, 72, 256),
       ("dense_c3_2", 27000, 256, 2304), ("dense_c4_2", 6750, 512, 4608), ("dense_c2_2", 108000, 128, 1152),
       ("fc6_b32", 9600, 4096, 25088) if len(sys.argv) > 2 else None]
CONV = [("conv1_2", 600, 720, 64, 64), ("conv2_1", 300, 360, 64, 128), ("conv2_2", 300, 360, 128, 128),
        ("conv3_1", 150, 180, 128, 256), ("conv3_2", 150, 180, 256, 256), ("conv4_1", 75, 90, 256, 512),
        ("conv4_2", 75, 90, 512, 512), ("conv5_1", 38, 45, 512, 512), ("rpn_conv", 38, 45, 512, 256)]
for a in sys.argv:
    if a.startswith("--hw="):             # another image size, e.g. --hw=320,480 (webcam regime): the conv list follows the ceil-mode pools
        H0, W0 = (int(v) for v in a.split("=")[1].split(","))
        hs = [(H0, W0)]
        for _ in range(4): hs.append(((hs[-1][0] + 1) // 2, (hs[-1][1] + 1) // 2))
        CONV = [("conv1_2", *hs[0], 64, 64), ("conv2_1", *hs[1], 64, 128), ("conv2_2", *hs[1], 128, 128),
                ("conv3_1", *hs[2], 128, 256), ("conv3_2", *hs[2], 256, 256), ("conv4_1", *hs[3], 256, 512),
                ("conv4_2", *hs[3], 512, 512), ("conv5_1", *hs[4], 512, 512), ("rpn_conv", *hs[4], 512, 256)]
POOLED = {"conv1_2", "conv2_2", "conv3_2", "conv4_2"}       # (conv3_3 / conv4_3 have conv3_2 / conv4_2's shapes) also timed with the fused 2x2 ceil-mode pool epilogue (as the trunk runs them)
if CUSTOM:
    LIN = [("%dx%dx%d" % c,) + c for c in CUSTOM]
    CONV = []
print("%-10s %10s %10s %8s" % ("op", "GFLOP", "us", "TF"))
for item in LIN:
    if item is None: continue
    name, M, N, K = item
    A = dev_rand(M * K); W = dev_rand(N * K); b = dev_rand(N); Cc = ctx.empty((M, N))
    check(ctx.h, lib.dc_op_linear(ctx.h, A.ptr, W.ptr, b.ptr, Cc.ptr, M, N, K, 1))
    prof(1)
    for _ in range(reps):
        check(ctx.h, lib.dc_op_linear(ctx.h, A.ptr, W.ptr, b.ptr, Cc.ptr, M, N, K, 1))
    l, ms, fl = prof(-1)
    print("%-10s %10.2f %10.1f %8.1f" % (name, fl / l / 1e9, ms / l * 1e3, fl / ms / 1e9))
    for x in (A, W, b, Cc): x.free()
for name, H, Wd, Cin, Cout in CONV:
    A = dev_rand(H * Wd * Cin); W = dev_rand(Cout * 9 * Cin); b = dev_rand(Cout); Cc = ctx.empty((H, Wd, Cout))
    check(ctx.h, lib.dc_op_conv3x3(ctx.h, A.ptr, W.ptr, b.ptr, Cc.ptr, 1, H, Wd, Cin, Cout, 1))
    prof(1)
    for _ in range(reps):
        check(ctx.h, lib.dc_op_conv3x3(ctx.h, A.ptr, W.ptr, b.ptr, Cc.ptr, 1, H, Wd, Cin, Cout, 1))
    l, ms, fl = prof(-1)
    print("%-10s %10.2f %10.1f %8.1f" % (name, fl / l / 1e9, ms / l * 1e3, fl / ms / 1e9))
    if name in POOLED:
        Cp = ctx.empty(((H + 1) // 2, (Wd + 1) // 2, Cout))
        check(ctx.h, lib.dc_op_conv3x3_relu_pool(ctx.h, A.ptr, W.ptr, b.ptr, Cp.ptr, H, Wd, Cin, Cout))
        prof(1)
        for _ in range(reps):
            check(ctx.h, lib.dc_op_conv3x3_relu_pool(ctx.h, A.ptr, W.ptr, b.ptr, Cp.ptr, H, Wd, Cin, Cout))
        l, ms, fl = prof(-1)
        print("%-10s %10.2f %10.1f %8.1f" % (name + "+pool", fl / l / 1e9, ms / l * 1e3, fl / ms / 1e9))
        Cp.free()
    for x in (A, W, b, Cc): x.free()
