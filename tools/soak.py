"""Soak: continuous batches through the product path for a few minutes, both caption orders and several lane / group settings in
turn, every batch compared with the first one of its setting (bit-identical) -- looks for rare hangs or races (round 6: the NMS band
scan hands chunks between waves through LDS flags; the packed decode reads a device-side row count).
usage (GPU box): python tools/soak.py [seconds]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from densecap_amd import DenseCapModel
from densecap_amd.weights import make_synthetic_weights, make_synthetic_image

seconds = float(sys.argv[1]) if len(sys.argv) > 1 else 180.0
m = DenseCapModel(make_synthetic_weights(seed=1234), device=0)
H, W, K, P = 600, 720, 32, 1000
if os.environ.get("SOAK_SMALL") == "1":                       # many short forwards: ~10x the NMS runs and decodes per second
    H, W, K, P = 224, 288, 64, 100
m.setTestArgs(rpn_nms_thresh=0.7, final_nms_thresh=0.3, num_proposals=P)
imgs = np.stack([make_synthetic_image(H, W, s) for s in range(16)])
dev = m.ctx.to_device(np.concatenate([imgs] * (K // 16)))
settings = [(2, 8, True), (4, 4, True), (3, 1, False), (4, 8, False), (2, 3, True), (1, 1, True)]
if os.environ.get("SOAK_FIXED"):                             # "lanes,group,order": one setting throughout
    l_, g_, o_ = (int(v) for v in os.environ["SOAK_FIXED"].split(","))
    settings = [(l_, g_, bool(o_))]
if os.environ.get("SOAK_ORDER") in ("0", "1"):               # one caption order only
    settings = [(l, g, bool(int(os.environ["SOAK_ORDER"]))) for l, g, _ in settings]
one_stream = os.environ.get("SOAK_ONE_STREAM") == "1"     # ONE stream, but the multi-lane planning (the kernels of the other runs)
if one_stream:
    from densecap_amd._lib import check
    check(m.ctx.h, m.ctx.lib.dc_debug_set(m.ctx.h, b"plan_mode", 0), "dc_debug_set")
    settings = [(1, 1, False)]
ref, t_end, batches, images, bad = None, time.time() + seconds, 0, 0, 0
while time.time() < t_end:
    for lanes, group, order in settings:
        m.setLanes(lanes); m.setGroup(group); m.setCaptionOrder(order)
        for _ in range(6):
            r = m.forward_batch_device(dev.ptr, K, H, W)
            if lanes >= 2 or one_stream:                     # (single-image mode plans some layers differently: its own bits)
                if ref is None:
                    ref = r
                for ii, (a, b) in enumerate(zip(r, ref)):
                    for name, x, y in zip(("boxes", "scores", "tokens"), a, b):
                        if not np.array_equal(x, y):
                            bad += 1
                            if x.shape != y.shape:
                                what = "shape %s vs %s" % (x.shape, y.shape)
                            else:
                                rows = np.nonzero((x != y).reshape(len(x), -1).any(axis=1))[0]
                                what = "%d rows differ, first %s: got %s want %s" % (len(rows), rows[:5].tolist(),
                                                                                     x[rows[0]].tolist(), y[rows[0]].tolist())
                            print("MISMATCH batch %d lanes %d group %d captions_after_nms %d image %d %s: %s" % (
                                batches, lanes, group, order, ii, name, what), flush=True)
            batches += 1; images += K
print("soak %s: %d batches, %d images, %d NMS runs in %.0f s, %d mismatching arrays" % ("ok" if bad == 0 else "FAILED", batches, images, 2 * images, seconds, bad))
sys.exit(1 if bad else 0)
