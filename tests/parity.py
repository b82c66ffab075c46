"""Strict end-to-end parity check of the HIP path against the CPU oracle (test infrastructure).

north_star bar: boxes / scores / features within 1e-4 relative fp32, greedy token ids identical.

Greedy NMS and arg-max are discontinuous, so "HIP == oracle" is established the only way that is
rigorous for a pipeline with integer decisions:

  (1) every CONTINUOUS tensor (trunk features, RPN probabilities and boxes, fc7 codes, objectness,
      final boxes) agrees with the oracle within 1e-4 relative, rows paired by anchor id / RPN pick id;
  (2) every INTEGER stage is bit-exact when the oracle is fed the HIP path's own inputs of that stage
      (teacher forcing): RPN NMS picks, final NMS picks, greedy tokens.  A token row may differ only if
      the oracle's own top-2 logit margin at the first differing step is below 1e-4 relative (proof of
      an fp32 near-tie), and such rows are reported;
  (3) the FINAL outputs (what forward_test returns) equal the oracle's: same K, every oracle box
      reproduced within 1e-4 relative at the same rank, scores within 1e-4, token rows identical.  A
      departure is accepted only with a proof taken from the ORACLE's data: an IoU within 1e-4 of the NMS
      threshold, a score gap below 1e-4 relative between two boxes whose order matters, or a top-2
      logit margin below 1e-4 -- at or before the rank where the two lists first differ.

No percentage thresholds: anything that is neither identical nor proven fails.
"""
import os

import numpy as np

REL = 1e-4


def oracle_threads():
    import torch
    torch.set_num_threads(max(1, min(32, os.cpu_count() or 1)))


def rel_err(a, b):
    """max |a-b| / max |b|  (tensor-level relative error, as DESIGN.md quotes it)."""
    return float(np.abs(np.asarray(a, np.float64) - np.asarray(b, np.float64)).max()) / max(float(np.abs(b).max()), 1e-30)


def row_rel_err(a, b):
    """max over rows of max|a-b| / max(1, max|b|) -- boxes in pixels, logits O(1..10)."""
    a = np.asarray(a, np.float64).reshape(len(b), -1)
    b = np.asarray(b, np.float64).reshape(len(b), -1)
    if len(b) == 0:
        return 0.0
    return float((np.abs(a - b).max(axis=1) / np.maximum(1.0, np.abs(b).max(axis=1))).max())


def iou_plus1(b, i, j):
    """box_utils.nms inline IoU (+1 convention, box_utils.lua:219-227) of corner boxes b[i], b[j] in float64."""
    x1 = max(b[i, 0], b[j, 0]); y1 = max(b[i, 1], b[j, 1]); x2 = min(b[i, 2], b[j, 2]); y2 = min(b[i, 3], b[j, 3])
    w = max(0.0, x2 - x1 + 1.0); h = max(0.0, y2 - y1 + 1.0)
    inter = w * h
    ai = (b[i, 2] - b[i, 0] + 1.0) * (b[i, 3] - b[i, 1] + 1.0)
    aj = (b[j, 2] - b[j, 0] + 1.0) * (b[j, 3] - b[j, 1] + 1.0)
    return inter / (ai + aj - inter)


def nms_near_tie(boxes5, picks, thr, upto_rank, tol=REL):
    """Is there, in the ORACLE's NMS run over boxes5 (x1,y1,x2,y2,score) with pick list `picks`, a decision within
    `tol` of flipping at or before pick rank `upto_rank`?  Decisions: (a) IoU(candidate, pick) vs thr for every pick of
    rank <= upto_rank and every candidate; (b) score order of two boxes with IoU > thr (which one survives).
    Returns a description string or None."""
    b = np.asarray(boxes5, np.float64)
    s = b[:, 4]
    n = len(b)
    if n == 0:
        return None
    x1, y1, x2, y2 = b[:, 0], b[:, 1], b[:, 2], b[:, 3]
    area = (x2 - x1 + 1.0) * (y2 - y1 + 1.0)
    for r, i in enumerate(picks[:upto_rank + 1]):
        w = np.maximum(0.0, np.minimum(x2, x2[i]) - np.maximum(x1, x1[i]) + 1.0)
        h = np.maximum(0.0, np.minimum(y2, y2[i]) - np.maximum(y1, y1[i]) + 1.0)
        inter = w * h
        iou = inter / (area + area[i] - inter)
        close = np.nonzero(np.abs(iou - thr) < tol)[0]
        close = close[close != i]
        if close.size:
            return "IoU(%d,%d)=%.7f within %g of thr %.3f at pick rank %d" % (i, close[0], iou[close[0]], tol, thr, r)
        over = np.nonzero((iou > thr) & (np.abs(s - s[i]) <= tol * np.maximum(1.0, np.abs(s[i]))))[0]
        over = over[over != i]
        if over.size:
            return "score gap %.3g between overlapping boxes %d,%d at pick rank %d" % (abs(s[over[0]] - s[i]), i, over[0], r)
    # (c) order of the pick list itself: adjacent scores closer than tol (a cap would cut differently)
    sp = s[np.asarray(picks[:upto_rank + 2], np.int64)]
    if len(sp) > 1:
        gaps = np.abs(np.diff(sp)) / np.maximum(1.0, np.abs(sp[:-1]))
        if (gaps < tol).any():
            return "adjacent pick scores within %g (rank %d)" % (tol, int(np.argmin(gaps)))
    return None


def token_divergence_proven(oracle_mod, codes_row, weights, T, hip_row, oracle_row, tol=REL):
    """First differing step of a token row must be an oracle top-2 logit near-tie (fp32 noise)."""
    import torch
    diff = np.nonzero(np.asarray(hip_row) != np.asarray(oracle_row))[0]
    if diff.size == 0:
        return True, None
    t = int(diff[0])
    _, logits = oracle_mod.lm_sample(torch.from_numpy(np.ascontiguousarray(codes_row[None], dtype=np.float32)), weights, T,
                                     return_logits=True)
    top2 = torch.topk(logits[t][0], 2).values
    margin = float(top2[0] - top2[1]) / max(1.0, float(top2[0].abs()))
    return margin < tol, "step %d top-2 margin %.3g" % (t, margin)


def compare_final(oracle_mod, weights, hip, ora, st, final_thr, T, report):
    """(3) of the module docstring.  hip/ora = (boxes, scores, tokens); st = oracle stages."""
    boxes, scores, tokens = hip
    ob, os_, oseq = ora
    if final_thr > 0:
        assert (np.diff(scores) <= 0).all(), "scores must come back in decreasing order (box_utils.nms contract)"
    n = min(len(boxes), len(ob))
    same = np.zeros(n, bool)
    for i in range(n):
        same[i] = np.abs(boxes[i] - ob[i]).max() <= REL * max(1.0, float(np.abs(ob[i]).max()))
    first_bad = int(np.argmin(same)) if (n and not same.all()) else (n if len(boxes) == len(ob) else n)
    lists_equal = len(boxes) == len(ob) and same.all()
    if not lists_equal:
        # the two final lists part at rank first_bad: demand the proof from the oracle's own final NMS
        b5 = np.concatenate([oracle_mod.xcycwh_to_x1y1x2y2(st["final_boxes_pre_nms"]), st["obj"][:, None]], 1)
        why = nms_near_tie(b5, st["final_nms_idx"], final_thr, first_bad)
        assert why is not None, ("final boxes differ from the oracle at rank %d (K %d vs %d) and no oracle decision is "
                                 "within 1e-4 of flipping" % (first_bad, len(boxes), len(ob)))
        report.setdefault("final_list_flips", []).append(why)
    # every oracle box that IS reproduced: score and tokens
    matched = 0
    for i, bx in enumerate(ob):
        if not len(boxes):
            break
        d = np.abs(boxes - bx).max(axis=1)
        j = int(np.argmin(d))
        if d[j] <= REL * max(1.0, float(np.abs(bx).max())):
            matched += 1
            assert abs(scores[j] - os_[i]) <= REL * max(1.0, abs(float(os_[i]))), "score of final box %d" % i
            if not (tokens[j] == oseq[i]).all():
                ridx = int(st["final_nms_idx"][i])
                ok, why = token_divergence_proven(oracle_mod, st["codes"][ridx], weights, T, tokens[j], oseq[i])
                assert ok, "final box %d: token row differs without a near-tie (%s)" % (i, why)
                report.setdefault("token_near_ties", []).append(why)
    if lists_equal:
        assert matched == len(ob)
    report.update(K=len(boxes), K_oracle=len(ob), matched=matched)
    return report


def strict_check(model, weights, img, P, rpn_thr=0.7, final_thr=0.3, T=None, stages=True):
    """Run one image through the HIP path and the oracle; assert (1)-(3).  Returns a report dict."""
    import torch
    from oracle import densecap_oracle as O
    oracle_threads()
    T = T or int(weights["seq_length"])
    report = {}
    model.setTestArgs(rpn_nms_thresh=rpn_thr, final_nms_thresh=final_thr, num_proposals=P)
    hip = model.forward_raw(img)
    st = {}
    ora = O.forward_test(img, weights, rpn_thr, final_thr, P, T, stages=st)
    if not stages:
        return compare_final(O, weights, hip, ora, st, final_thr, T, report)
    H, W = img.shape[1:]
    fh, fw = st["feat"].shape[1:]
    k = O.DEFAULT_ANCHORS.shape[1]
    A = k * fh * fw
    # ---- (1) continuous: trunk, RPN ---------------------------------------------------------------------
    feat, _ = model.debug_fetch("feat_hwc", (fh, fw, 512))
    report["trunk_rel_err"] = rel_err(feat.transpose(2, 0, 1), st["feat"])
    assert report["trunk_rel_err"] < REL
    valid, _ = model.debug_fetch("rpn_valid", (A,), np.uint8)
    np.testing.assert_array_equal(valid.astype(bool), st["rpn"]["valid"])
    rows = st["rpn"]["rows"]                       # anchor rows the oracle kept (valid-mask compaction)
    p, _ = model.debug_fetch("rpn_p", (A,))
    report["rpn_p_abs_err"] = float(np.abs(p[rows] - st["rpn"]["p"]).max())
    assert report["rpn_p_abs_err"] <= REL          # probabilities live in [0,1]: absolute == relative to 1
    rb, _ = model.debug_fetch("rpn_boxes", (A, 4))
    report["rpn_boxes_rel_err"] = row_rel_err(rb[rows], st["rpn"]["boxes"])
    assert report["rpn_boxes_rel_err"] <= REL
    xyxy, _ = model.debug_fetch("rpn_x1y1x2y2", (A, 4))
    # ---- (2) RPN NMS, teacher-forced on the HIP path's own boxes / scores: bit-exact picks ----------------------
    Pcap = model._capacity(H, W)
    idx, _ = model.debug_fetch("rpn_nms_idx", (Pcap,), np.int32)
    cnt, _ = model.debug_fetch("rpn_nms_count", (1,), np.int32)
    B = int(cnt[0])
    vrows = np.nonzero(valid)[0]
    tf = O.nms(np.concatenate([xyxy[vrows], p[vrows, None]], 1), rpn_thr, None if P == -1 else P)
    np.testing.assert_array_equal(idx[:B], vrows[tf], err_msg="RPN NMS picks differ from the oracle run on the same inputs")
    opicks = rows[st["rpn_nms_idx"]]
    assert B == len(opicks)
    report["rpn_picks"] = B
    report["rpn_picks_same_rank"] = int((idx[:B] == opicks).sum())
    if not (idx[:B] == opicks).all():
        # fed the oracle's own p/boxes the pick list differs: must be an fp32 near-tie in the ORACLE's NMS
        first = int(np.argmin(idx[:B] == opicks))
        b5 = np.concatenate([st["rpn"]["x1y1x2y2"], st["rpn"]["p"][:, None]], 1)
        why = nms_near_tie(b5, st["rpn_nms_idx"], rpn_thr, first, tol=REL)
        assert why is not None, "RPN pick lists differ at rank %d with no oracle near-tie" % first
        report["rpn_flip"] = why
    roi, _ = model.debug_fetch("roi_boxes", (Pcap, 4))
    np.testing.assert_array_equal(roi[:B], rb[idx[:B]])
    # ---- (1) continuous after the RPN, rows paired through the anchor id of the pick ---------------------
    pos_o = {int(v): i for i, v in enumerate(opicks)}
    pairs = [(i, pos_o[int(v)]) for i, v in enumerate(idx[:B]) if int(v) in pos_o]
    report["rpn_picks_in_common"] = len(pairs)
    hi = np.array([a for a, _ in pairs], np.int64); oi = np.array([b for _, b in pairs], np.int64)
    D = int(weights["fc7_w"].shape[0])
    codes, _ = model.debug_fetch("codes", (Pcap, D))
    report["fc7_codes_rel_err"] = rel_err(codes[hi], st["codes"][oi])
    assert report["fc7_codes_rel_err"] <= REL
    obj, _ = model.debug_fetch("obj", (Pcap,))
    report["obj_rel_err"] = row_rel_err(obj[hi], st["obj"][oi])
    assert report["obj_rel_err"] <= REL
    fb, _ = model.debug_fetch("final_boxes", (Pcap, 4))
    report["final_boxes_pre_nms_rel_err"] = row_rel_err(fb[hi], st["final_boxes_pre_nms"][oi])
    assert report["final_boxes_pre_nms_rel_err"] <= REL
    # ---- (2) greedy decode, teacher-forced on the HIP path's fc7 codes -----------------------------------------
    seq, _ = model.debug_fetch("seq", (Pcap, T), np.int32)
    tf_seq = O.lm_sample(torch.from_numpy(np.ascontiguousarray(codes[:B])), weights, T)
    bad = np.nonzero((seq[:B] != tf_seq).any(axis=1))[0]
    for r in bad:
        ok, why = token_divergence_proven(O, codes[r], weights, T, seq[r], tf_seq[r])
        assert ok, "decode row %d differs from the oracle on identical codes without a near-tie (%s)" % (r, why)
        report.setdefault("decode_near_ties", []).append(why)
    report["decode_rows"] = B
    report["decode_rows_identical"] = B - len(bad)
    assert seq[:B].min() >= 1 and seq[:B].max() <= int(weights["vocab_size"]) + 1
    # ---- (2) final NMS, teacher-forced on the HIP path's final boxes / objectness ---------------------------------
    idx2, _ = model.debug_fetch("final_nms_idx", (Pcap,), np.int32)
    cnt2, _ = model.debug_fetch("final_nms_count", (1,), np.int32)
    K = int(cnt2[0])
    if final_thr > 0:
        tf2 = O.nms(np.concatenate([O.xcycwh_to_x1y1x2y2(fb[:B]), obj[:B, None]], 1), final_thr, None)
    else:
        tf2 = np.arange(B)
    np.testing.assert_array_equal(idx2[:K], tf2, err_msg="final NMS picks differ from the oracle run on the same inputs")
    np.testing.assert_array_equal(hip[0], fb[idx2[:K]])
    np.testing.assert_array_equal(hip[1], obj[idx2[:K]])
    np.testing.assert_array_equal(hip[2], seq[idx2[:K]])
    # ---- (3) final outputs vs the oracle -------------------------------------------------------------------------
    return compare_final(O, weights, hip, ora, st, final_thr, T, report)
