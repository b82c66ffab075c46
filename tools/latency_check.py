"""Single-image latency (lanes = 1, what run_model / the daemon use) of forward_test, both caption orders
(dc_set_caption_order: 0 = the reference's, 1 = captions after the final NMS, the CLIs' default), at 720x600 / 1000
proposals and in the webcam regime (480x320 / 50 proposals).
usage: python tools/latency_check.py [reps]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from densecap_amd import DenseCapModel
from densecap_amd.weights import make_synthetic_weights, make_synthetic_image

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 30
m = DenseCapModel(make_synthetic_weights(seed=1234), device=0)
for (H, W, P) in ((600, 720, 1000), (600, 720, 300), (320, 480, 50)):
    m.setTestArgs(rpn_nms_thresh=0.7, final_nms_thresh=0.3, num_proposals=P)
    img = make_synthetic_image(H, W, 0)
    dev = m.ctx.to_device(img[None])
    for lanes in (1, 3):
        for order in (False, True):
            m.setLanes(lanes); m.setCaptionOrder(order)
            for _ in range(3):
                m.forward_batch_device(dev.ptr, 1, H, W)
            ts = []
            for _ in range(reps):
                t0 = time.perf_counter()
                r = m.forward_batch_device(dev.ptr, 1, H, W)
                ts.append(time.perf_counter() - t0)
            ts = np.array(ts) * 1e3
            print("%dx%d P=%d lanes=%d captions_after_final_nms=%d  latency ms: median %.3f  min %.3f  max %.3f   K=%d  stages %s" %
                  (W, H, P, lanes, order, np.median(ts), ts.min(), ts.max(), len(r[0][0]),
                   {k: round(v, 3) for k, v in m.stage_times().items()}), flush=True)
    dev.free()
m.setCaptionOrder(False)
