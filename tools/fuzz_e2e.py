"""Randomised end-to-end comparison HIP vs CPU oracle over image sizes, proposal counts and thresholds
(small vocabulary so the oracle is quick).  usage: python tools/fuzz_e2e.py [n_cases] [seed]"""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from densecap_amd import DenseCapModel
from densecap_amd.weights import make_synthetic_weights, make_synthetic_image
from oracle import densecap_oracle as O

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 30
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
W = make_synthetic_weights(seed=99, vocab_size=333, seq_length=7)
m = DenseCapModel(W, device=0)
bad = 0
for case in range(n_cases):
    H = int(rng.integers(33, 420)); Wd = int(rng.integers(33, 520))
    P = int(rng.choice([1, 2, 7, 50, 64, 65, 128, 300, 1000, -1]))
    rthr = float(rng.choice([0.0, 0.3, 0.7, 1.0])); fthr = float(rng.choice([-1.0, 0.0, 0.3, 0.5, 1.0]))
    lanes = int(rng.choice([1, 3])); order = bool(rng.integers(0, 2))
    img = make_synthetic_image(H, Wd, 1000 + case)
    m.setLanes(lanes); m.setCaptionOrder(order)
    m.setTestArgs(rpn_nms_thresh=rthr, final_nms_thresh=fthr, num_proposals=P)
    b, s, t = m.forward_raw(img)
    ob, os_, oseq = O.forward_test(img, W, rthr, fthr, P, 7)
    # every oracle box must exist in the HIP output (identity match), scores in decreasing order, tokens identical on matches
    matched = tok_same = 0
    for i, bx in enumerate(ob):
        if len(b) == 0:
            break
        d = np.abs(b - bx).max(axis=1); j = int(np.argmin(d))
        if d[j] <= 1e-3 * max(1.0, np.abs(bx).max()):
            matched += 1
            tok_same += int((t[j] == oseq[i]).all())
    ok = (abs(len(b) - len(ob)) <= max(1, len(ob) // 50) and matched >= len(ob) - max(1, len(ob) // 50)
          and tok_same >= matched - max(1, matched // 50) and (fthr <= 0 or (np.diff(s) <= 0).all()))
    bad += not ok
    print(json.dumps(dict(case=case, H=H, W=Wd, P=P, rpn_thr=rthr, final_thr=fthr, lanes=lanes, caption_after_nms=order,
                          K_hip=len(b), K_oracle=len(ob), matched=matched, tok_same=tok_same, ok=bool(ok))), flush=True)
print("FUZZ %s: %d/%d cases ok" % ("OK" if bad == 0 else "FAILED", n_cases - bad, n_cases))
sys.exit(1 if bad else 0)
