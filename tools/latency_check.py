"""Single-image latency (lanes = 1, what run_model / the daemon use) of forward_test at 720x600 / 1000 proposals.
usage: python tools/latency_check.py [reps]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from densecap_amd import DenseCapModel
from densecap_amd.weights import make_synthetic_weights, make_synthetic_image

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 30
m = DenseCapModel(make_synthetic_weights(seed=1234), device=0)
m.setTestArgs(rpn_nms_thresh=0.7, final_nms_thresh=0.3, num_proposals=1000)
img = make_synthetic_image(600, 720, 0)
dev = m.ctx.to_device(img[None])
for lanes in (1, 3):
    m.setLanes(lanes)
    for _ in range(3):
        m.forward_batch_device(dev.ptr, 1, 600, 720)
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter()
        r = m.forward_batch_device(dev.ptr, 1, 600, 720)
        ts.append(time.perf_counter() - t0)
    ts = np.array(ts) * 1e3
    print("lanes=%d  latency ms: median %.3f  min %.3f  max %.3f   K=%d  stages %s" %
          (lanes, np.median(ts), ts.min(), ts.max(), len(r[0][0]),
           {k: round(v, 3) for k, v in m.stage_times().items()}))
