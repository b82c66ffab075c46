"""CPU ORACLE for the densecap test-time hot path -- TEST INFRASTRUCTURE ONLY.

This file restates, on the CPU, the arithmetic of the reference's
`DenseCapModel:forward_test()` path (jcjohnson/densecap).  It exists so that
the HIP path in `densecap_amd/` can be checked; it is NEVER imported by the
product path.  Only `tests/`, `__graft_entry__.smoke()` and the
`cpu_baseline` leg of `bench.py` may import it.

Pinning status
--------------
* In-tree Lua arithmetic (box algebra, NMS, anchors, box->affine, token
  decode): PINNED against the reference's own known-answer tests
  (`tests/golden/reference_vectors.json`, transcribed from
  test/nms_test.lua:9-95, test/ApplyBoxTransform_test.lua:12-35,
  test/BoxToAffine_test.lua:14-44, test/MakeBoxes_test.lua:47-234,
  test/ReshapeBoxFeatures_test.lua:33-58, test/LanguageModel_test.lua:135-160,
  test/box_conversion_test.lua:12-23).
* Arithmetic that lives in un-vendored, un-pinned third-party rocks
  (torch/nn SpatialConvolution / SpatialMaxPooling(ceil) / Linear /
  LookupTable, qassemoquab/stnbhwd AffineGridGeneratorBHWD +
  BilinearSamplerBHWD, jcjohnson/torch-rnn nn.LSTM): restated from the
  published algorithms; the reference has only shape/self-consistency tests at
  those boundaries, so these stages are "PARITY UNPINNED" (no Torch7/Lua
  runtime and no checkpoint exist in this environment).

Conventions: every function takes/returns numpy float32 (or torch CPU fp32 for
the dense stages) and keeps the reference's operation ORDER in fp32 so that
integer outputs (NMS picks, argmax tokens) can be compared bit-exactly under
teacher forcing.  Indices are 0-based here; the reference is 1-based.
"""
from __future__ import annotations

import ctypes
import os
import numpy as np

F32 = np.float32

# ----------------------------------------------------------------------------
# Box algebra  (densecap/box_utils.lua)
# ----------------------------------------------------------------------------

def xcycwh_to_x1y1x2y2(boxes):
    """box_utils.lua:270-298. x0 = ((w-1)/2)*-1 + xc ; x1 = (w-1)/2 + xc (fp32)."""
    b = np.asarray(boxes, dtype=F32)
    xc, yc, w, h = b[..., 0], b[..., 1], b[..., 2], b[..., 3]
    hw = (w - F32(1)) / F32(2)
    hh = (h - F32(1)) / F32(2)
    out = np.empty_like(b)
    out[..., 0] = -hw + xc
    out[..., 1] = -hh + yc
    out[..., 2] = hw + xc
    out[..., 3] = hh + yc
    return out


def x1y1x2y2_to_xcycwh(boxes):
    """box_utils.lua:382-410. xc=(x0+x1)/2, w = x1-x0 (NO +1)."""
    b = np.asarray(boxes, dtype=F32)
    out = np.empty_like(b)
    out[..., 0] = (b[..., 0] + b[..., 2]) / F32(2)
    out[..., 1] = (b[..., 1] + b[..., 3]) / F32(2)
    out[..., 2] = b[..., 2] - b[..., 0]
    out[..., 3] = b[..., 3] - b[..., 1]
    return out


def x1y1x2y2_to_xywh(boxes):
    """box_utils.lua:300-327: w = x1 - x0 + 1."""
    b = np.asarray(boxes, dtype=F32)
    out = np.empty_like(b)
    out[..., 0] = b[..., 0]
    out[..., 1] = b[..., 1]
    out[..., 2] = b[..., 2] - b[..., 0] + F32(1)
    out[..., 3] = b[..., 3] - b[..., 1] + F32(1)
    return out


def xywh_to_x1y1x2y2(boxes):
    """box_utils.lua:329-356: x1 = x0 + w - 1."""
    b = np.asarray(boxes, dtype=F32)
    out = np.empty_like(b)
    out[..., 0] = b[..., 0]
    out[..., 1] = b[..., 1]
    out[..., 2] = b[..., 0] + b[..., 2] - F32(1)
    out[..., 3] = b[..., 1] + b[..., 3] - F32(1)
    return out


def xcycwh_to_xywh(boxes):
    """box_utils.lua:441-445 (called run_model.lua:78)."""
    return x1y1x2y2_to_xywh(xcycwh_to_x1y1x2y2(boxes))


def clip_boxes_xcycwh(boxes, x_min, y_min, x_max, y_max):
    """box_utils.clip_boxes(boxes, bounds, 'xcycwh')  (box_utils.lua:486-523).

    Returns (clipped_xcycwh, valid[bool]).  Note the two conversions are not
    inverses: every box loses 1 px of w and h."""
    c = xcycwh_to_x1y1x2y2(boxes).reshape(-1, 4)
    c[:, 0] = np.clip(c[:, 0], F32(x_min), F32(x_max - 1))
    c[:, 1] = np.clip(c[:, 1], F32(y_min), F32(y_max - 1))
    c[:, 2] = np.clip(c[:, 2], F32(x_min + 1), F32(x_max))
    c[:, 3] = np.clip(c[:, 3], F32(y_min + 1), F32(y_max))
    valid = (c[:, 2] > c[:, 0]) & (c[:, 3] > c[:, 1])
    return x1y1x2y2_to_xcycwh(c).reshape(np.shape(boxes)), valid


def nms_py(boxes5, overlap, max_boxes=None):
    """box_utils.nms (box_utils.lua:154-256), vector-op-per-pick form.

    boxes5: (N,5) x1,y1,x2,y2,score.  Returns 0-based picks in decreasing
    score order.  Tie rule (TH sort order on exact ties is unobservable):
    among equal scores the LOWER original index is picked first."""
    b = np.asarray(boxes5, dtype=F32)
    if b.size == 0:
        return np.zeros((0,), np.int64)
    x1, y1, x2, y2, s = b[:, 0], b[:, 1], b[:, 2], b[:, 3], b[:, 4]
    area = (x2 - x1 + F32(1)) * (y2 - y1 + F32(1))
    # ascending sort, take from the tail; stable on (-s) => lower index first
    # TH's sort ranks NaN above every number (end of the ascending list = picked first)
    s64 = s.astype(np.float64)
    order = np.lexsort((np.arange(len(s64)), -np.where(np.isnan(s64), 0.0, s64), ~np.isnan(s64)))  # descending
    I = order[::-1].copy()  # ascending list, best at the tail
    thr = F32(overlap)
    pick = []
    while (max_boxes is None or len(pick) < max_boxes) and I.size > 0:
        i = I[-1]
        pick.append(i)
        if I.size == 1:
            break
        I = I[:-1]
        xx1 = np.maximum(x1, x1[i]); xx2 = np.minimum(x2, x2[i])
        yy1 = np.maximum(y1, y1[i]); yy2 = np.minimum(y2, y2[i])
        w = np.maximum(xx2 - xx1 + F32(1), F32(0))
        h = np.maximum(yy2 - yy1 + F32(1), F32(0))
        inter = w * h
        union = (area + area[i]) - inter
        with np.errstate(divide="ignore", invalid="ignore"):
            iou = inter / union
        I = I[iou[I] <= thr]
    return np.asarray(pick, dtype=np.int64)


# ----------------------------------------------------------------------------
# C restatement (fast) of NMS and the bilinear sampler: oracle/oracle_c.c
# ----------------------------------------------------------------------------
_HERE = os.path.dirname(os.path.abspath(__file__))
_CLIB = None


def _clib():
    global _CLIB
    if _CLIB is None:
        path = os.path.join(_HERE, "_build", "liboracle_c.so")
        if not os.path.exists(path):
            import subprocess
            subprocess.check_call(["make", "-s", "-C", _HERE])
        lib = ctypes.CDLL(path)
        lib.oracle_nms.restype = ctypes.c_int
        lib.oracle_nms.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_float,
                                   ctypes.c_int, ctypes.c_void_p]
        lib.oracle_bilinear_roi_pool.restype = None
        lib.oracle_bilinear_roi_pool.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int,
                                                 ctypes.c_int, ctypes.c_void_p, ctypes.c_int,
                                                 ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                                 ctypes.c_int, ctypes.c_void_p]
        _CLIB = lib
    return _CLIB


def nms(boxes5, overlap, max_boxes=None):
    """C restatement of box_utils.nms (same semantics and tie rule as nms_py)."""
    b = np.ascontiguousarray(boxes5, dtype=F32)
    n = b.shape[0]
    if n == 0:
        return np.zeros((0,), np.int64)
    out = np.empty((n,), np.int32)
    k = _clib().oracle_nms(b.ctypes.data, n, ctypes.c_float(float(F32(overlap))),
                           -1 if max_boxes is None else int(max_boxes), out.ctypes.data)
    return out[:k].astype(np.int64)


# ----------------------------------------------------------------------------
# Anchors / transforms  (densecap/modules/*.lua)
# ----------------------------------------------------------------------------
DEFAULT_ANCHORS = np.array([[45, 90], [90, 45], [64, 64], [90, 180], [180, 90], [128, 128],
                            [181, 362], [362, 181], [256, 256], [362, 724], [724, 362],
                            [512, 512]], dtype=F32).T.copy()  # (2,k)  LocalizationLayer.lua:613-619
VGG16_FIELD_CENTERS = (8.5, 8.5, 16.0, 16.0)  # net_utils.lua:106-140 for layers 1..30


def make_anchors(h, w, x0, y0, sx, sy, anchors):
    """nn.MakeAnchors (MakeAnchors.lua:40-67) -> (4k,h,w); view (k,4,h,w)."""
    k = anchors.shape[1]
    xs = np.arange(w, dtype=F32) * F32(sx) + F32(x0)
    ys = np.arange(h, dtype=F32) * F32(sy) + F32(y0)
    out = np.empty((k, 4, h, w), F32)
    out[:, 0] = xs[None, None, :]
    out[:, 1] = ys[None, :, None]
    out[:, 2] = anchors[0].astype(F32)[:, None, None]
    out[:, 3] = anchors[1].astype(F32)[:, None, None]
    return out.reshape(4 * k, h, w)


def reshape_box_features(x, k):
    """nn.ReshapeBoxFeatures (ReshapeBoxFeatures.lua:24-33): (k*D,h,w)->(k*h*w,D);
    row b = a*h*w + y*w + x."""
    kd, h, w = x.shape
    d = kd // k
    return np.ascontiguousarray(x.reshape(k, d, h, w).transpose(0, 2, 3, 1)).reshape(k * h * w, d)


def th_exp(x):
    """torch.exp on a FloatTensor as 2016-era TH computes it: the C double `exp` on every element, result cast to float
    (TH generic/THTensorMath.c LAB_IMPLEMENT_BASIC_FUNCTION(exp,exp); the TH sources are not in /root/reference --
    recorded as an assumption in docs/SEMANTICS.md).  Overflow sits where the float result overflows (x > ~88.72 -> inf)."""
    with np.errstate(over="ignore", invalid="ignore"):
        return np.exp(np.asarray(x, F32).astype(np.float64)).astype(F32)


def th_sigmoid(x):
    """TH_sigmoid (THMath.h): 1.0 / (1.0 + exp(-x)) in double, cast to float -- torch-rnn's LSTM gates (`:sigmoid()`)."""
    import torch
    return (1.0 / (1.0 + torch.exp(-x.double()))).float()


def th_tanh(x):
    """`:tanh()` on a FloatTensor: the C double tanh, cast to float."""
    import torch
    return torch.tanh(x.double()).float()


def apply_box_transform(boxes, trans):
    """nn.ApplyBoxTransform (ApplyBoxTransform.lua:63-90)."""
    b = np.asarray(boxes, F32).reshape(-1, 4)
    t = np.asarray(trans, F32).reshape(-1, 4)
    out = np.empty_like(b)
    out[:, 0] = t[:, 0] * b[:, 2] + b[:, 0]
    out[:, 1] = t[:, 1] * b[:, 3] + b[:, 1]
    out[:, 2] = th_exp(t[:, 2]) * b[:, 2]
    out[:, 3] = th_exp(t[:, 3]) * b[:, 3]
    return out.reshape(np.shape(boxes))


def make_boxes(head, x0, y0, sx, sy, anchors):
    """MakeAnchors o Reshape o ApplyBoxTransform == legacy nn.MakeBoxes
    (MakeAnchors_test.lua:17-44).  head: (4k,h,w) -> (k*h*w,4)."""
    k = anchors.shape[1]
    a = reshape_box_features(make_anchors(head.shape[1], head.shape[2], x0, y0, sx, sy, anchors), k)
    t = reshape_box_features(np.asarray(head, F32), k)
    return apply_box_transform(a, t)


def box_to_affine(boxes, H, W):
    """nn.BoxToAffine (BoxToAffine.lua:69-93) -> (B,2,3); rows are (y,x)."""
    b = np.asarray(boxes, F32)
    th = np.zeros((b.shape[0], 2, 3), F32)
    th[:, 1, 2] = (b[:, 0] * F32(2) + F32(-1 - W)) / F32(W - 1)
    th[:, 0, 2] = (b[:, 1] * F32(2) + F32(-1 - H)) / F32(H - 1)
    th[:, 1, 1] = b[:, 2] / F32(W)
    th[:, 0, 0] = b[:, 3] / F32(H)
    return th


def affine_grid(theta, HH, WW):
    """stnbhwd nn.AffineGridGeneratorBHWD(HH,WW) (un-vendored; called at
    BilinearRoiPooling.lua:52).  grid[b,i,j,:] = theta_b . (y_i, x_j, 1);
    y_i = -1 + 2 i/(HH-1).  k-ordered fp32 sum (y*t0 + x*t1) + t2."""
    ys = np.array([-1.0 + (i / (HH - 1)) * 2 for i in range(HH)], dtype=F32)
    xs = np.array([-1.0 + (j / (WW - 1)) * 2 for j in range(WW)], dtype=F32)
    B = theta.shape[0]
    g = np.empty((B, HH, WW, 2), F32)
    for r in range(2):
        t0 = theta[:, r, 0][:, None, None]; t1 = theta[:, r, 1][:, None, None]
        t2 = theta[:, r, 2][:, None, None]
        g[..., r] = (ys[None, :, None] * t0 + xs[None, None, :] * t1) + t2
    return g


def bilinear_sample_hwc(feat_hwc, grids):
    """stnbhwd BilinearSamplerBHWD_updateOutput as driven by
    BatchBilinearSamplerBHWD.lua:104-122 (one (H,W,C) image, B grids).
    Taps outside the map contribute 0.  Pure numpy (vectorised over all)."""
    f = np.asarray(feat_hwc, F32)
    H, W, C = f.shape
    yf = grids[..., 0]; xf = grids[..., 1]
    xcoord = (xf + F32(1)) * F32(W - 1) / F32(2)
    ycoord = (yf + F32(1)) * F32(H - 1) / F32(2)
    x0 = np.floor(xcoord); y0 = np.floor(ycoord)
    wx = F32(1) - (xcoord - x0); wy = F32(1) - (ycoord - y0)
    x0 = x0.astype(np.int64); y0 = y0.astype(np.int64)

    def tap(yy, xx):
        ok = (xx >= 0) & (xx <= W - 1) & (yy >= 0) & (yy <= H - 1)
        v = f[np.clip(yy, 0, H - 1), np.clip(xx, 0, W - 1)]
        return np.where(ok[..., None], v, F32(0))

    wx = wx[..., None].astype(F32); wy = wy[..., None].astype(F32)
    out = (wx * wy * tap(y0, x0)
           + (F32(1) - wx) * wy * tap(y0, x0 + 1)
           + wx * (F32(1) - wy) * tap(y0 + 1, x0)
           + (F32(1) - wx) * (F32(1) - wy) * tap(y0 + 1, x0 + 1))
    return out.astype(F32)


def bilinear_roi_pool(feat_chw, boxes, img_h, img_w, HH=7, WW=7):
    """nn.BilinearRoiPooling forward (BilinearRoiPooling.lua:42-60,84-91):
    (C,h,w)+(B,4 xcycwh in image px) -> (B,C,HH,WW).  C restatement."""
    f = np.ascontiguousarray(feat_chw, dtype=F32)
    b = np.ascontiguousarray(boxes, dtype=F32)
    C, h, w = f.shape
    out = np.empty((b.shape[0], C, HH, WW), F32)
    _clib().oracle_bilinear_roi_pool(f.ctypes.data, C, h, w, b.ctypes.data, b.shape[0],
                                     img_h, img_w, HH, WW, out.ctypes.data)
    return out


def bilinear_roi_pool_np(feat_chw, boxes, img_h, img_w, HH=7, WW=7):
    """Same as bilinear_roi_pool, numpy composition of the four modules."""
    theta = box_to_affine(boxes, img_h, img_w)
    g = affine_grid(theta, HH, WW)
    out = bilinear_sample_hwc(np.transpose(feat_chw, (1, 2, 0)), g)  # (B,HH,WW,C)
    return np.ascontiguousarray(out.transpose(0, 3, 1, 2))


def box_iou_module(b1, b2):
    """nn.BoxIoU (BoxIoU.lua:40-73), code-as-written: corners via (w-1)/2,
    area = w*h, intersection WITHOUT the +1 convention.  b1:(B1,4) b2:(B2,4)."""
    a = xcycwh_to_x1y1x2y2(b1); b = xcycwh_to_x1y1x2y2(b2)
    area1 = (b1[:, 2] * b1[:, 3]).astype(F32)[:, None]
    area2 = (b2[:, 2] * b2[:, 3]).astype(F32)[None, :]
    x0 = np.maximum(a[:, None, 0], b[None, :, 0]); y0 = np.maximum(a[:, None, 1], b[None, :, 1])
    x1 = np.minimum(a[:, None, 2], b[None, :, 2]); y1 = np.minimum(a[:, None, 3], b[None, :, 3])
    w = np.maximum(x1 - x0, F32(0)); h = np.maximum(y1 - y0, F32(0))
    inter = w * h
    return (inter / ((area1 + area2) - inter)).astype(F32)


def box_iou(b1, b2, convention="boxiou_module"):
    """Pairwise IoU of xcycwh boxes under the three conventions found in the reference:
    "boxiou_module"  nn.BoxIoU as written today (BoxIoU.lua:40-73): (w-1)/2 corners, area w*h, no +1;
    "nms_plus1"      box_utils.nms inline (box_utils.lua:178-181,219-227): (w-1)/2 corners, +1 on every extent;
    "legacy_half_w"  the module's original converter (BoxIoU.lua:15-37, commented out): corners xc -/+ w/2, area w*h,
                     no +1 -- the convention test/BoxIoU_test.lua:13-94 was written for."""
    b1 = np.asarray(b1, F32); b2 = np.asarray(b2, F32)
    if convention == "boxiou_module":
        return box_iou_module(b1, b2)
    if convention == "legacy_half_w":
        def conv(b):
            o = np.empty_like(b)
            o[:, 0] = (b[:, 2] / F32(2)) * F32(-1) + b[:, 0]; o[:, 2] = b[:, 2] / F32(2) + b[:, 0]
            o[:, 1] = (b[:, 3] / F32(2)) * F32(-1) + b[:, 1]; o[:, 3] = b[:, 3] / F32(2) + b[:, 1]
            return o
        a, b, one = conv(b1), conv(b2), F32(0)
        area1 = (b1[:, 2] * b1[:, 3])[:, None]; area2 = (b2[:, 2] * b2[:, 3])[None, :]
    elif convention == "nms_plus1":
        a, b, one = xcycwh_to_x1y1x2y2(b1), xcycwh_to_x1y1x2y2(b2), F32(1)
        area1 = ((a[:, 2] - a[:, 0] + one) * (a[:, 3] - a[:, 1] + one))[:, None]
        area2 = ((b[:, 2] - b[:, 0] + one) * (b[:, 3] - b[:, 1] + one))[None, :]
    else:
        raise ValueError(convention)
    x0 = np.maximum(a[:, None, 0], b[None, :, 0]); y0 = np.maximum(a[:, None, 1], b[None, :, 1])
    x1 = np.minimum(a[:, None, 2], b[None, :, 2]); y1 = np.minimum(a[:, None, 3], b[None, :, 3])
    w = np.maximum(x1 - x0 + one, F32(0)); h = np.maximum(y1 - y0 + one, F32(0))
    inter = w * h
    return (inter / ((area1 + area2) - inter)).astype(F32)


# ----------------------------------------------------------------------------
# Dense stages (torch CPU fp32 stands in for THNN im2col+sgemm)
# ----------------------------------------------------------------------------
VGG16_CFG = [(3, 64), (64, 64), "P", (64, 128), (128, 128), "P", (128, 256), (256, 256),
             (256, 256), "P", (256, 512), (512, 512), (512, 512), "P", (512, 512), (512, 512),
             (512, 512)]  # caffemodel layers 1..30; no pool5 (DenseCapModel.lua:61-63)


def vgg16_trunk(img, conv_w, conv_b):
    """conv_net1+conv_net2 (DenseCapModel.lua:73-76). img: torch (1,3,H,W).
    Max-pools are ceil-mode (loadcaffe / Caffe semantics)."""
    import torch
    import torch.nn.functional as Fn
    x = img
    li = 0
    for item in VGG16_CFG:
        if item == "P":
            x = Fn.max_pool2d(x, 2, 2, ceil_mode=True)
        else:
            x = Fn.relu(Fn.conv2d(x, conv_w[li], conv_b[li], padding=1))
            li += 1
    return x


def rpn_heads(feat, Wt):
    """build_rpn convs (LocalizationLayer.lua:627-673): returns box head (4k,h,w)
    and score head (2k,h,w) as numpy."""
    import torch.nn.functional as Fn
    hid = Fn.relu(Fn.conv2d(feat, Wt["rpn_conv_w"], Wt["rpn_conv_b"], padding=1))
    box = Fn.conv2d(hid, Wt["rpn_box_w"], Wt["rpn_box_b"])
    sc = Fn.conv2d(hid, Wt["rpn_score_w"], Wt["rpn_score_b"])
    return box[0].numpy(), sc[0].numpy()


def rpn_decode(box_head, score_head, img_h, img_w, anchors=DEFAULT_ANCHORS,
               field_centers=VGG16_FIELD_CENTERS, clip_boxes=True):
    """LocalizationLayer._forward_test lines 265-308 after the convs:
    anchors+transform, clip, mask-compaction, corners, p(pos).
    clip_boxes=False (self.test_clip_boxes, LocalizationLayer.lua:235,272): lines 272-300 are skipped -- the boxes
    stay as transformed and every row survives.
    Returns dict with compacted arrays (A',.) and the original row ids."""
    k = anchors.shape[1]
    x0, y0, sx, sy = field_centers
    h, w = box_head.shape[1:]
    anc = reshape_box_features(make_anchors(h, w, x0, y0, sx, sy, anchors), k)
    trans = reshape_box_features(box_head, k)
    scores2 = reshape_box_features(score_head, k)
    boxes = apply_box_transform(anc, trans)
    if clip_boxes:
        clipped, valid = clip_boxes_xcycwh(boxes, 1, 1, img_w, img_h)
    else:
        clipped, valid = boxes, np.ones(boxes.shape[0], bool)
    keep = np.nonzero(valid)[0]
    boxes_c = clipped[keep]; anc_c = anc[keep]; trans_c = trans[keep]; sc_c = scores2[keep]
    x1y1x2y2 = xcycwh_to_x1y1x2y2(boxes_c)
    e = th_exp(sc_c)
    with np.errstate(divide="ignore", invalid="ignore", over="ignore"):
        p = (F32(1) / (e[:, 0] + e[:, 1])) * e[:, 0]   # pow(-1) then cmul (LocalizationLayer.lua:308)
    return dict(boxes=boxes_c, anchors=anc_c, trans=trans_c, scores2=sc_c,
                x1y1x2y2=x1y1x2y2, p=p.astype(F32), rows=keep, valid=valid)


def lstm_step(x_gates, h, c, Wh):
    """torch-rnn nn.LSTM single step (un-vendored; LanguageModel.lua:51):
    gates = (b + x.Wx) + h.Wh ; [i f o g] ; c' = f*c + i*g ; h' = o*tanh(c').
    x_gates = b + x.Wx already (N,4H); Wh (H,4H).  torch CPU fp32."""
    import torch
    Hd = h.shape[1]
    g = x_gates + h @ Wh
    i = th_sigmoid(g[:, :Hd]); f = th_sigmoid(g[:, Hd:2 * Hd])
    o = th_sigmoid(g[:, 2 * Hd:3 * Hd]); gg = th_tanh(g[:, 3 * Hd:])
    c2 = f * c + i * gg
    h2 = o * th_tanh(c2)
    return h2, c2


def lm_sample(codes, Wt, T, return_logits=False):
    """LM:sample with sample_argmax (LanguageModel.lua:293-348).
    codes: torch (N,4096).  Returns int64 (N,T) tokens, 1-based, START=END=V+1.
    Wt: lm_enc_w (512,4096), lm_enc_b, lm_emb (V+2,512), lstm_w (1024,2048)
    [rows 0..511 = Wx, 512..1023 = Wh], lstm_b (2048), lm_out_w (V+1,512), lm_out_b."""
    import torch
    N = codes.shape[0]
    Hd = Wt["lstm_w"].shape[1] // 4
    D = Wt["lstm_w"].shape[0] - Hd
    Wx = Wt["lstm_w"][:D]; Wh = Wt["lstm_w"][D:]
    V1 = Wt["lm_out_w"].shape[0]  # V+1
    enc = torch.relu(codes @ Wt["lm_enc_w"].t() + Wt["lm_enc_b"])
    h = torch.zeros(N, Hd); c = torch.zeros(N, Hd)
    h, c = lstm_step(Wt["lstm_b"] + enc @ Wx, h, c, Wh)       # step 0, output ignored
    seq = torch.zeros(N, T, dtype=torch.int64)
    tok = torch.full((N,), V1, dtype=torch.int64)             # START = V+1 (1-based)
    all_logits = []
    for t in range(T):
        x = Wt["lm_emb"][tok - 1]
        h, c = lstm_step(Wt["lstm_b"] + x @ Wx, h, c, Wh)
        logits = h @ Wt["lm_out_w"].t() + Wt["lm_out_b"]
        tok = torch.argmax(logits, dim=1) + 1                 # first max on ties
        seq[:, t] = tok
        if return_logits:
            all_logits.append(logits)
    if return_logits:
        return seq.numpy(), all_logits
    return seq.numpy()


def _log_softmax_thnn(x):
    """nn.LogSoftMax on a FloatTensor row (THNN generic/LogSoftMax.c, CPU): exp and the running sum in double
    (accreal), logsum = max + log(sum), output = float(x - logsum)."""
    x = np.asarray(x, F32)
    mx = x.max(axis=-1, keepdims=True)
    logsum = mx.astype(np.float64) + np.log(np.exp((x - mx).astype(np.float64)).sum(axis=-1, keepdims=True))
    return (x.astype(np.float64) - logsum).astype(F32)


def _topk_sorted(v, k):
    """torch.topk(v, k, dim, true) over the last axis, sorted descending; ties (unspecified in the reference): lower index first."""
    order = np.argsort(-np.asarray(v, np.float64), axis=-1, kind="stable")[..., :k]
    return np.take_along_axis(v, order, -1), order


def _topk_margin(v, k):
    """Smallest gap between neighbours among the k+1 largest DISTINCT-position values of each row (how close the
    selection or its order is to changing); exact ties (gap 0, resolved by index on both sides) do not count."""
    sv = -np.sort(-np.asarray(v, np.float64), axis=-1)[..., :k + 1]
    gaps = np.abs(np.diff(sv, axis=-1))
    gaps = np.where(gaps == 0, np.inf, gaps)
    return float(gaps.min()) if gaps.size else np.inf


def lm_beamsearch(codes, Wt, T, beam_size, return_margins=False):
    """LM:beamsearch (LanguageModel.lua:170-290), one proposal at a time with the beams as the minibatch, exactly as
    written -- including the zeroed next-word log-probabilities of finished beams (:243-247), beams initialised with
    fill(1) (:209), the first beam expansion seeding h with the CELL state (:224), and seq[i] = beams[argmax
    beam_logprobs] (:281-282).  (Consequence of :224: beam_size = 1 is NOT LM:sample.)  Returns int64 (N,T), 1-based ids."""
    import torch
    N = codes.shape[0]
    Hd = Wt["lstm_w"].shape[1] // 4
    D = Wt["lstm_w"].shape[0] - Hd
    Wx = Wt["lstm_w"][:D]; Wh = Wt["lstm_w"][D:]
    V1 = Wt["lm_out_w"].shape[0]
    END = V1
    seq = np.zeros((N, T), np.int64)
    margins = np.full((N,), np.inf)
    for i in range(N):
        enc = torch.relu(codes[i:i + 1] @ Wt["lm_enc_w"].t() + Wt["lm_enc_b"])
        h = torch.zeros(1, Hd); c = torch.zeros(1, Hd)
        h, c = lstm_step(Wt["lstm_b"] + enc @ Wx, h, c, Wh)                              # image step
        start = torch.full((1,), V1, dtype=torch.int64)
        h, c = lstm_step(Wt["lstm_b"] + Wt["lm_emb"][start - 1] @ Wx, h, c, Wh)          # START step
        scores = (h @ Wt["lm_out_w"].t() + Wt["lm_out_b"]).numpy()
        beams = np.ones((beam_size, T), np.int64)
        lp0 = _log_softmax_thnn(scores)[0]
        beam_logprobs, idx = _topk_sorted(lp0, beam_size)
        margins[i] = min(margins[i], _topk_margin(lp0, beam_size))
        beams[:, 0] = idx + 1
        # :221-226 `layer.cell = layer.cell:expand(..):clone()` then `layer.output = layer.cell:expand(..):clone()`: the
        # hidden state of every beam is seeded with the CELL state (torch-rnn's nn.LSTM with remember_states takes h0 from
        # self.output, c0 from self.cell) -- as written, not "fixed"
        c = c.expand(beam_size, Hd).clone(); h = c.clone()
        for t in range(1, T):
            words = torch.from_numpy(beams[:, t - 1])
            h, c = lstm_step(Wt["lstm_b"] + Wt["lm_emb"][words - 1] @ Wx, h, c, Wh)
            lp = _log_softmax_thnn((h @ Wt["lm_out_w"].t() + Wt["lm_out_b"]).numpy())
            end_mask = ((beams == END).sum(1) == 0).astype(F32)
            lp = lp * end_mask[:, None]
            top_lp, word_idx = _topk_sorted(lp, beam_size)                               # (beam, beam)
            all_next = (top_lp.reshape(-1) + np.repeat(beam_logprobs, beam_size)).astype(F32)
            live = end_mask > 0
            if live.any():
                margins[i] = min(margins[i], _topk_margin(lp[live], beam_size))
            margins[i] = min(margins[i], _topk_margin(all_next, beam_size))
            beam_logprobs, flat = _topk_sorted(all_next, beam_size)
            all_next_beams = np.repeat(beams, beam_size, axis=0)
            all_next_beams[:, t] = word_idx.reshape(-1) + 1
            beams = all_next_beams[flat]
            parent = torch.from_numpy(flat // beam_size)
            h = h[parent]; c = c[parent]
        seq[i] = beams[int(np.argmax(beam_logprobs))]
    if return_margins:
        return seq, margins
    return seq


def decode_sequence(seq, idx_to_token, vocab_size):
    """LM:decodeSequence (LanguageModel.lua:86-103): stop at END(=V+1) or 0."""
    caps = []
    end = vocab_size + 1
    for row in np.asarray(seq):
        words = []
        for tok in row:
            tok = int(tok)
            if tok == end or tok == 0:
                break
            words.append(idx_to_token[tok])
        caps.append(" ".join(words))
    return caps


# ----------------------------------------------------------------------------
# run_image preprocessing (run_model.lua:67-74); image.scale lives in torch/image (un-vendored)
# ----------------------------------------------------------------------------
def _scale_linear_rowcol(src, dst_len):
    """torch/image generic/image.c scaleLinear_rowcol on one 1-D sequence of a DoubleTensor, scalar loops (small cases
    only).  run_model.lua:67-68 never sets the default tensor type, so image.load returns a DoubleTensor and `real` is
    double in the C loops -- which keep `float scale`, `float acc`, `float n`, `float si_f` locals: a product of a float
    weight and a double sample is a double, every assignment to `acc` rounds it to float, `acc / n` is a float division,
    and the interpolation `(1 - si_f) * a + si_f * b` stays double (round 6; rounds 2-5 restated an all-float32 chain)."""
    F64 = np.float64
    src = [F64(v) for v in src]
    n_src = len(src)
    if dst_len == n_src:
        return list(src)
    dst = [F64(0)] * dst_len
    if dst_len > n_src:
        if n_src == 1:
            return [src[0]] * dst_len
        scale = F32(n_src - 1) / F32(dst_len - 1)
        for di in range(dst_len - 1):
            si_f = F32(di) * scale
            si_i = int(si_f)
            si_f = F32(si_f - F32(si_i))
            dst[di] = F64(F32(F32(1) - si_f)) * src[si_i] + F64(si_f) * src[si_i + 1]
        dst[dst_len - 1] = src[n_src - 1]
        return dst
    scale = F32(n_src) / F32(dst_len)
    si0_i, si0_f = 0, F32(0)
    for di in range(dst_len):
        si1_f = F32(di + 1) * scale
        si1_i = int(si1_f)
        si1_f = F32(si1_f - F32(si1_i))
        acc = F32(F64(F32(F32(1) - si0_f)) * src[si0_i])
        n = F32(F32(1) - si0_f)
        for si in range(si0_i + 1, si1_i):
            acc = F32(F64(acc) + src[si]); n = F32(n + F32(1))
        if si1_i < n_src:
            acc = F32(F64(acc) + F64(si1_f) * src[si1_i]); n = F32(n + si1_f)
        dst[di] = F64(F32(acc / n))
        si0_i, si0_f = si1_i, si1_f
    return dst


def image_load_u8(rgb_u8_hwc):
    """image.load(path, 3) after the file decode (run_model.lua:67): a DoubleTensor (3,H,W) of byte / 255."""
    return np.asarray(rgb_u8_hwc, np.uint8).astype(np.float64).transpose(2, 0, 1) / np.float64(255.0)


def image_scale(img_chw, size):
    """image.scale(img, size) (run_model.lua:68) on the DoubleTensor image.load returns: longer side -> size, bilinear =
    rows then columns; a double (C, oh, ow) array (the `:float()` of run_model.lua:68 is the caller's)."""
    img = np.asarray(img_chw, np.float64)
    C, ih, iw = img.shape
    imax = max(ih, iw)
    oh, ow = int(ih * size / imax), int(iw * size / imax)
    tmp = np.empty((C, ih, ow), np.float64)
    for c in range(C):
        for y in range(ih):
            tmp[c, y] = _scale_linear_rowcol(img[c, y], ow)
    out = np.empty((C, oh, ow), np.float64)
    for c in range(C):
        for x in range(ow):
            out[c, :, x] = _scale_linear_rowcol(tmp[c, :, x], oh)
    return out


def preprocess(img_rgb01_chw, image_size):
    """run_model.lua:68-74: scale (in double), :float(), index {3,2,1} (BGR), mul 255, add -vgg_mean."""
    img = image_scale(img_rgb01_chw, image_size).astype(F32)
    mean = np.array([103.939, 116.779, 123.68], F32)
    return (img[::-1] * F32(255) - mean[:, None, None])[None].astype(F32)


def forward_test(img, Wt, rpn_nms_thresh=0.7, final_nms_thresh=0.3, num_proposals=1000,
                 T=15, stages=None, beam_size=None, clip_boxes=True, nms_impl="c"):
    """DenseCapModel:forward_test numerics (DenseCapModel.lua:242-275,319-327 via
    LocalizationLayer.lua:250-363).  img: numpy/torch (3,H,W) BGR mean-subtracted.
    Returns (boxes_xcycwh (K,4), scores (K,), tokens (K,T) int64 1-based).
    If `stages` is a dict, intermediate tensors are stored in it.
    nms_impl: "c" = the early-out C restatement (fast; what the parity tests use), "vector" = nms_py, the reference's own
    algorithmic form -- one full-length vector pass per pick (box_utils.lua:206-249) -- which is what BASELINE.md 3 asks the
    CPU baseline to time.  Same picks either way (tests/test_oracle_golden.py)."""
    import torch
    torch.set_grad_enabled(False)
    nms_fn = nms_py if nms_impl == "vector" else nms
    st = stages if stages is not None else {}
    img_t = torch.as_tensor(np.asarray(img, F32))[None]
    H, W = img_t.shape[2:]
    feat = vgg16_trunk(img_t, Wt["conv_w"], Wt["conv_b"])                  # (1,512,h,w)
    st["feat"] = feat[0].numpy()
    box_head, score_head = rpn_heads(feat, Wt)
    st["box_head"] = box_head; st["score_head"] = score_head
    d = rpn_decode(box_head, score_head, H, W, clip_boxes=clip_boxes)
    st["rpn"] = d
    b5 = np.concatenate([d["x1y1x2y2"], d["p"][:, None]], 1)
    idx = nms_fn(b5, rpn_nms_thresh, None if num_proposals == -1 else num_proposals)
    st["rpn_nms_idx"] = idx
    roi_boxes = d["boxes"][idx]
    st["roi_boxes"] = roi_boxes
    roi = bilinear_roi_pool(st["feat"], roi_boxes, H, W)                    # (B,512,7,7)
    st["roi_feats"] = roi
    x = torch.from_numpy(roi.reshape(roi.shape[0], -1))
    x = torch.relu(x @ Wt["fc6_w"].t() + Wt["fc6_b"])
    codes = torch.relu(x @ Wt["fc7_w"].t() + Wt["fc7_b"])
    st["codes"] = codes.numpy()
    obj = (codes @ Wt["obj_w"].t() + Wt["obj_b"])[:, 0].numpy()
    trans = (codes @ Wt["boxreg_w"].t() + Wt["boxreg_b"]).numpy()
    final_boxes = apply_box_transform(roi_boxes, trans)
    st["obj"] = obj; st["final_trans"] = trans; st["final_boxes_pre_nms"] = final_boxes
    if beam_size:                                   # LM:updateOutput dispatch (LanguageModel.lua:129-131)
        seq, st["beam_margins"] = lm_beamsearch(codes, Wt, T, beam_size, return_margins=True)
    else:
        seq = lm_sample(codes, Wt, T)
    st["seq_pre_nms"] = seq
    if final_nms_thresh > 0:
        b5 = np.concatenate([xcycwh_to_x1y1x2y2(final_boxes), obj[:, None]], 1)
        idx2 = nms_fn(b5, final_nms_thresh, None)
    else:
        idx2 = np.arange(final_boxes.shape[0])
    st["final_nms_idx"] = idx2
    return final_boxes[idx2], obj[idx2], seq[idx2]
