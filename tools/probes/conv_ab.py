"""One conv3x3 shape, many calls, one tile configuration (dc_debug_set force_cfg): run under `rocprofv3 --kernel-trace --stats`
to compare kernel time (contraction + split-K reduce) between routes.
usage: python tools/probes/conv_ab.py H W Cin Cout force_cfg [reps] [serial]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from densecap_amd.ops import Context
from densecap_amd._lib import check
H, W, Cin, Cout, cfg = (int(v) for v in sys.argv[1:6])
reps = int(sys.argv[6]) if len(sys.argv) > 6 else 100
serial = int(sys.argv[7]) if len(sys.argv) > 7 else 1
ctx = Context(0)
if serial:
    check(ctx.h, ctx.lib.dc_set_lanes(ctx.h, 1), "dc_set_lanes")
check(ctx.h, ctx.lib.dc_debug_set(ctx.h, b"force_cfg", cfg), "dc_debug_set")
rng = np.random.default_rng(0)
x = ctx.to_device(rng.standard_normal((H, W, Cin)).astype(np.float32))
w = ctx.to_device((rng.standard_normal((Cout, 9 * Cin)) * 0.02).astype(np.float32))
b = ctx.to_device(rng.standard_normal(Cout).astype(np.float32))
o = ctx.empty((H, W, Cout))
for _ in range(reps):
    check(ctx.h, ctx.lib.dc_op_conv3x3(ctx.h, x.ptr, w.ptr, b.ptr, o.ptr, 1, H, W, Cin, Cout, 1), "dc_op_conv3x3")
print("done", H, W, Cin, Cout, cfg)
