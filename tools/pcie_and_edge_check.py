import os, sys, time, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from densecap_amd import DenseCapModel
from densecap_amd.weights import make_synthetic_weights, make_synthetic_image
W = make_synthetic_weights(); m = DenseCapModel(W, 0); m.setTestArgs(num_proposals=1000)
imgs = np.stack([make_synthetic_image(600, 720, i) for i in range(12)])
m.forward_batch(imgs[:3])
t0 = time.perf_counter(); m.forward_batch(imgs); t1 = time.perf_counter()
print("host-resident images: %.1f img/s" % (12 / (t1 - t0)))
dev = m.ctx.to_device(imgs)
m.forward_batch_device(dev.ptr, 3, 600, 720)
t0 = time.perf_counter(); m.forward_batch_device(dev.ptr, 12, 600, 720); t1 = time.perf_counter()
print("HBM-resident images: %.1f img/s" % (12 / (t1 - t0)))
# tiny image edge case
m.setTestArgs(num_proposals=50)
b, s, t = m.forward_raw(make_synthetic_image(48, 64, 1))
print("tiny 48x64:", b.shape, s[:3], t.shape)
m.setTestArgs(num_proposals=5000)
b, s, t = m.forward_raw(make_synthetic_image(96, 128, 2))
print("P=5000 > anchors(%d):" % (12*6*8), b.shape, t.shape)
