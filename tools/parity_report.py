"""Prints the end-to-end parity statistics of the HIP path vs the CPU oracle (run on the GPU box)."""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from densecap_amd import DenseCapModel
from densecap_amd.weights import make_synthetic_weights, make_synthetic_image
from oracle import densecap_oracle as O

W = make_synthetic_weights(seed=1234)
m = DenseCapModel(W, device=0)
rows = []
SETTINGS = [(600, 720, 1000, 0), (600, 720, 1000, 1), (600, 720, 300, 2), (720, 1080, 2000, 5), (480, 720, 1000, 7)]
SETTINGS += [(600, 720, 1000, sd) for sd in range(10, 10 + int(os.environ.get('PARITY_EXTRA', '0')))]
for (H, Wd, P, seed) in SETTINGS:
    img = make_synthetic_image(H, Wd, seed)
    m.setTestArgs(rpn_nms_thresh=0.7, final_nms_thresh=0.3, num_proposals=P)
    b, s, t = m.forward_raw(img)
    st = {}
    ob, os_, oseq = O.forward_test(img, W, 0.7, 0.3, P, 15, stages=st)
    fh, fw = st["feat"].shape[1:]
    feat, _ = m.debug_fetch("feat_hwc", (fh, fw, 512))
    feat_err = float(np.abs(feat.transpose(2, 0, 1) - st["feat"]).max() / np.abs(st["feat"]).max())
    A = 12 * fh * fw
    p, _ = m.debug_fetch("rpn_p", (A,))
    p_err = float(np.abs(p - st["rpn"]["p"]).max())
    idx, _ = m.debug_fetch("rpn_nms_idx", (P,), np.int32)
    cnt, _ = m.debug_fetch("rpn_nms_count", (1,), np.int32)
    same_order = int((idx[:cnt[0]] == st["rpn_nms_idx"][:cnt[0]]).sum()) if cnt[0] == len(st["rpn_nms_idx"]) else -1
    overlap = np.intersect1d(idx[:cnt[0]], st["rpn_nms_idx"]).size
    codes, _ = m.debug_fetch("codes", (P, 4096))
    seq_all, _ = m.debug_fetch("seq", (P, 15), np.int32)
    # all P rows BEFORE the final NMS, matched through the RPN pick id (the two pick lists can differ in order on near-ties)
    pos_o = {int(v): i for i, v in enumerate(st["rpn_nms_idx"])}
    common = [(i, pos_o[int(v)]) for i, v in enumerate(idx[:cnt[0]]) if int(v) in pos_o]
    hi = np.array([a for a, _ in common]); oi = np.array([b_ for _, b_ in common])
    codes_err = float(np.abs(codes[hi] - st["codes"][oi]).max() / max(1e-30, np.abs(st["codes"]).max()))
    pre_rows_same = int((seq_all[hi] == st["seq_pre_nms"][oi]).all(axis=1).sum())
    matched = tok_same = 0
    box_err = score_err = 0.0
    for i, bx in enumerate(ob):
        d = np.abs(b - bx).max(axis=1)
        j = int(np.argmin(d))
        if d[j] <= 1e-3 * max(1.0, np.abs(bx).max()):
            matched += 1
            box_err = max(box_err, float(d[j] / max(1.0, np.abs(bx).max())))
            score_err = max(score_err, float(abs(s[j] - os_[i]) / max(1.0, abs(os_[i]))))
            tok_same += int((t[j] == oseq[i]).all())
    rows.append(dict(H=H, W=Wd, P=P, seed=seed, trunk_rel_err=feat_err, rpn_p_abs_err=p_err, rpn_picks=int(cnt[0]),
                     rpn_picks_identical_position=same_order, rpn_picks_in_common=int(overlap), K_hip=len(b), K_oracle=len(ob),
                     final_boxes_matched=matched, max_box_rel_err=box_err, max_score_rel_err=score_err,
                     token_rows_identical=tok_same, fc7_codes_rel_err=codes_err,
                     pre_nms_rows_compared=len(common), pre_nms_token_rows_identical=pre_rows_same))
    print(json.dumps(rows[-1]))
json.dump(rows, open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out", "parity_report.json"), "w"), indent=1)
