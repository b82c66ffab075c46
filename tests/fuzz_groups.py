"""Randomised check that lanes and image groups are pure scheduling: a batch call with any (lanes >= 2, group) setting
must give every image exactly the bits it gets alone in a one-image multi-lane call, over random image sizes, proposal
counts, thresholds and batch sizes (small vocabulary so that a case takes milliseconds).  One case in five runs in
single-image mode (lanes = 1) with a group setting: include/densecap.h says the group is ignored there, i.e. every image
gets the bits of a one-image single-lane call (round-4 advisor finding: the fuzz only drew lanes >= 2).
FUZZ_CROSS_ORDER=1 (round 6): the batch call runs with captions AFTER the final NMS (one packed decode per group) and is
compared with the REFERENCE caption order image by image -- the order, too, must be invisible in the outputs.
usage: python tests/fuzz_groups.py [n_cases] [seed]"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402

from densecap_amd import DenseCapModel  # noqa: E402
from densecap_amd.weights import make_synthetic_image, make_synthetic_weights  # noqa: E402


def main(n_cases, seed):
    rng = np.random.default_rng(seed)
    m = DenseCapModel(make_synthetic_weights(seed=77, vocab_size=257, seq_length=6), device=0)
    bad = 0
    for case in range(n_cases):
        H = int(rng.integers(33, 500)); Wd = int(rng.integers(33, 640))
        if rng.integers(0, 5) == 0:
            H = int(rng.integers(500, 800)); Wd = int(rng.integers(640, 1100))
        P = int(rng.choice([1, 3, 50, 64, 100, 128, 300, 1000, -1]))
        n = int(rng.integers(2, 10)); G = int(rng.choice([1, 2, 3, 4, 4, 6, 8])); lanes = int(rng.integers(2, 5))
        if rng.integers(0, 5) == 0:
            lanes = 1
        order = bool(rng.integers(0, 2))
        cross = os.environ.get("FUZZ_CROSS_ORDER") == "1"
        if cross:
            order = True
        m.setTestArgs(rpn_nms_thresh=float(rng.choice([0.3, 0.7, 1.0])), final_nms_thresh=float(rng.choice([-1.0, 0.0, 0.3, 0.5])),
                      num_proposals=P)
        m.setCaptionOrder(order)
        imgs = np.stack([make_synthetic_image(H, Wd, 5000 + 16 * case + s) for s in range(n)])
        rec = dict(case=case, H=H, W=Wd, P=P, n=n, group=G, lanes=lanes, caption_after_nms=order, cross_order=cross)
        try:
            m.setLanes(lanes); m.setGroup(G)
            got = m.forward_batch(imgs)
            m.setLanes(1 if lanes == 1 else 2); m.setGroup(1)
            if cross:
                m.setCaptionOrder(False)
            for i in range(n):
                ref = m.forward_batch(imgs[i:i + 1])[0]
                for x, y, name in zip(got[i], ref, ("boxes", "scores", "tokens")):
                    assert x.shape == y.shape and np.array_equal(x, y), "image %d %s differ" % (i, name)
            rec["ok"] = True; rec["K"] = [int(len(g[0])) for g in got]
        except AssertionError as e:
            rec["ok"] = False; rec["why"] = str(e)[:300]; bad += 1
        print(json.dumps(rec), flush=True)
    m.setGroup(0)
    print("GROUP FUZZ %s: %d/%d cases ok" % ("OK" if bad == 0 else "FAILED", n_cases - bad, n_cases))
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main(int(sys.argv[1]) if len(sys.argv) > 1 else 40, int(sys.argv[2]) if len(sys.argv) > 2 else 0))
