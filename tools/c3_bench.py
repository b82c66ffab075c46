"""conv1_1 (Cin = 3, CHW image -> HWC activations) alone: run under `rocprofv3 --kernel-trace --stats` for the kernel time.
usage: python tools/c3_bench.py [reps] [H W]"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from densecap_amd.ops import Context
from densecap_amd._lib import check

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 30
H, W = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (600, 720)
ctx = Context(0)
rng = np.random.default_rng(0)
x = ctx.to_device(rng.standard_normal((3, H, W)).astype(np.float32))
w = ctx.to_device((rng.standard_normal((64, 3, 3, 3)) * 0.2).astype(np.float32))
b = ctx.to_device(rng.standard_normal(64).astype(np.float32))
o = ctx.empty((H, W, 64))
ts = []
for _ in range(reps):
    t0 = time.perf_counter()
    check(ctx.h, ctx.lib.dc_op_conv3x3_c3(ctx.h, x.ptr, w.ptr, b.ptr, o.ptr, H, W, 64, 1), "dc_op_conv3x3_c3")
    ts.append(time.perf_counter() - t0)
print("conv1_1 %dx%d: host call (launch + sync) median %.1f us; output %.1f MB" % (H, W, np.median(ts) * 1e6, H * W * 64 * 4 / 1e6))
