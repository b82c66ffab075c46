"""Per-op host wrappers over the C ABI (numpy in / numpy out, device buffers via dc_malloc).

Names follow the reference's modules (densecap/modules/*.lua, densecap/box_utils.lua) so the
parity tests read like the reference's own tests.  Every function runs the HIP kernel; there is
no CPU path here.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib
from ._lib import check


class Context:
    """Owns a dc_ctx (utils.setup_gpus equivalent, densecap/utils.lua:22-36)."""

    def __init__(self, device=0):
        self.lib = _lib.lib()
        h = C.c_void_p()
        rc = self.lib.dc_create(C.byref(h), int(device))
        if rc < 0:
            msg = self.lib.dc_last_error(None)
            raise _lib.DenseCapError("dc_create failed (%d): %s" % (rc, msg.decode() if msg else "?"))
        self.h = h
        self.device = device

    def set_math_mode(self, mode):
        """dc_set_math_mode (include/densecap.h): 0 = fp32 MFMA, 1 = split-bf16 -- applies to every contraction of this ctx."""
        check(self.h, self.lib.dc_set_math_mode(self.h, int(mode)), "dc_set_math_mode")

    def close(self):
        if getattr(self, "h", None):
            self.lib.dc_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- device buffers ----
    def to_device(self, arr):
        arr = np.ascontiguousarray(arr)
        buf = DeviceArray(self, arr.shape, arr.dtype)
        check(self.h, self.lib.dc_memcpy_h2d(self.h, buf.ptr, arr.ctypes.data, arr.nbytes), "dc_memcpy_h2d")
        return buf

    def empty(self, shape, dtype=np.float32):
        return DeviceArray(self, shape, dtype)


class DeviceArray:
    def __init__(self, ctx, shape, dtype):
        self.ctx = ctx
        self.shape = tuple(int(s) for s in (shape if isinstance(shape, (tuple, list)) else (shape,)))
        self.dtype = np.dtype(dtype)
        self.nbytes = int(np.prod(self.shape, dtype=np.int64)) * self.dtype.itemsize
        p = C.c_void_p()
        check(ctx.h, ctx.lib.dc_malloc(ctx.h, C.byref(p), max(self.nbytes, 16)), "dc_malloc")
        self.ptr = p

    def numpy(self):
        out = np.empty(self.shape, self.dtype)
        if self.nbytes:
            check(self.ctx.h, self.ctx.lib.dc_memcpy_d2h(self.ctx.h, out.ctypes.data, self.ptr, self.nbytes),
                  "dc_memcpy_d2h")
        return out

    def free(self):
        if self.ptr is not None and self.ctx.h:
            self.ctx.lib.dc_free(self.ctx.h, self.ptr)
        self.ptr = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


# ---- preprocessing (run_model.lua:67-74 on the device) ----------------------------------------
def preprocess_size(lib, H0, W0, image_size):
    H, W = C.c_int(0), C.c_int(0)
    rc = lib.dc_preprocess_size(int(H0), int(W0), int(image_size), C.byref(H), C.byref(W))
    if rc < 0:
        raise ValueError("image.scale: %dx%d -> size %d leaves no pixels" % (W0, H0, image_size))
    return H.value, W.value


def preprocess_u8(ctx, rgb_hwc_u8, image_size, want_rgb=True, out=None, rgb=None):
    """image.load's float conversion + image.scale(img, image_size) + BGR, x255, minus the VGG mean, on the device
    (dc_preprocess_u8).  rgb_hwc_u8: (H0,W0,3) uint8 as a JPEG decoder delivers it.  Returns (DeviceArray (3,H,W) float32 --
    what forward_images_device takes --, DeviceArray (H,W,3) uint8 of the scaled image for the visualiser or None).
    out / rgb: buffers of exactly those shapes to fill instead of new ones."""
    a = np.ascontiguousarray(rgb_hwc_u8, dtype=np.uint8)
    if a.ndim != 3 or a.shape[2] != 3:
        raise ValueError("preprocess_u8 wants an (H,W,3) uint8 image")
    H0, W0 = a.shape[:2]
    H, W = preprocess_size(ctx.lib, H0, W0, image_size)
    if out is None:
        out = ctx.empty((3, H, W), np.float32)
    if rgb is None and want_rgb:
        rgb = ctx.empty((H, W, 3), np.uint8)
    if out.shape != (3, H, W) or (rgb is not None and rgb.shape != (H, W, 3)):
        raise ValueError("preprocess_u8: output buffers do not have the scaled image's shape")
    check(ctx.h, ctx.lib.dc_preprocess_u8(ctx.h, a.ctypes.data, H0, W0, 0, int(image_size), out.ptr, rgb.ptr if rgb else None),
          "dc_preprocess_u8")
    return out, rgb


# ---- layout ----------------------------------------------------------------------------------
def chw_to_hwc(ctx, x):
    C_, H, W = x.shape
    a = ctx.to_device(_f32(x)); o = ctx.empty((H, W, C_))
    check(ctx.h, ctx.lib.dc_op_chw_to_hwc(ctx.h, a.ptr, o.ptr, C_, H, W), "dc_op_chw_to_hwc")
    return o.numpy()


def hwc_to_chw(ctx, x):
    H, W, C_ = x.shape
    a = ctx.to_device(_f32(x)); o = ctx.empty((C_, H, W))
    check(ctx.h, ctx.lib.dc_op_hwc_to_chw(ctx.h, a.ptr, o.ptr, C_, H, W), "dc_op_hwc_to_chw")
    return o.numpy()


# ---- dense ops ---------------------------------------------------------------------------------
def conv3x3(ctx, x_nchw, w_oihw, bias, relu=True):
    """nn.SpatialConvolution(Cin,Cout,3,3,1,1,1,1)[+ReLU] on (N,Cin,H,W) -> (N,Cout,H,W)."""
    x = _f32(x_nchw); w = _f32(w_oihw)
    N, Cin, H, W = x.shape
    Cout = w.shape[0]
    if Cin == 3:
        assert N == 1
        xi = ctx.to_device(x[0]); wd = ctx.to_device(w); bd = ctx.to_device(_f32(bias))
        o = ctx.empty((H, W, Cout))
        check(ctx.h, ctx.lib.dc_op_conv3x3_c3(ctx.h, xi.ptr, wd.ptr, bd.ptr, o.ptr, H, W, Cout, int(relu)),
              "dc_op_conv3x3_c3")
        return np.ascontiguousarray(o.numpy().transpose(2, 0, 1))[None]
    xh = ctx.to_device(np.ascontiguousarray(x.transpose(0, 2, 3, 1)))
    wd = ctx.to_device(w); wp = ctx.empty((Cout, 9 * Cin)); bd = ctx.to_device(_f32(bias))
    check(ctx.h, ctx.lib.dc_op_pack_conv3x3_weights(ctx.h, wd.ptr, wp.ptr, Cout, Cin), "pack")
    o = ctx.empty((N, H, W, Cout))
    check(ctx.h, ctx.lib.dc_op_conv3x3(ctx.h, xh.ptr, wp.ptr, bd.ptr, o.ptr, N, H, W, Cin, Cout, int(relu)),
          "dc_op_conv3x3")
    return np.ascontiguousarray(o.numpy().transpose(0, 3, 1, 2))


def conv3x3_relu_pool(ctx, x_chw, w_oihw, bias):
    """conv3x3 + ReLU + ceil-mode 2x2 max-pool in one launch: (Cin,H,W) -> (Cout, ceil(H/2), ceil(W/2))."""
    x = _f32(x_chw); w = _f32(w_oihw)
    Cin, H, W = x.shape
    Cout = w.shape[0]
    xh = ctx.to_device(np.ascontiguousarray(x.transpose(1, 2, 0)))
    wd = ctx.to_device(w); wp = ctx.empty((Cout, 9 * Cin)); bd = ctx.to_device(_f32(bias))
    check(ctx.h, ctx.lib.dc_op_pack_conv3x3_weights(ctx.h, wd.ptr, wp.ptr, Cout, Cin), "pack")
    o = ctx.empty(((H + 1) // 2, (W + 1) // 2, Cout))
    check(ctx.h, ctx.lib.dc_op_conv3x3_relu_pool(ctx.h, xh.ptr, wp.ptr, bd.ptr, o.ptr, H, W, Cin, Cout),
          "dc_op_conv3x3_relu_pool")
    return np.ascontiguousarray(o.numpy().transpose(2, 0, 1))


def maxpool2x2_ceil(ctx, x_nchw):
    x = _f32(x_nchw)
    N, C_, H, W = x.shape
    xh = ctx.to_device(np.ascontiguousarray(x.transpose(0, 2, 3, 1)))
    o = ctx.empty((N, (H + 1) // 2, (W + 1) // 2, C_))
    check(ctx.h, ctx.lib.dc_op_maxpool2x2_ceil(ctx.h, xh.ptr, o.ptr, N, H, W, C_), "dc_op_maxpool2x2_ceil")
    return np.ascontiguousarray(o.numpy().transpose(0, 3, 1, 2))


def linear(ctx, x, w, bias=None, relu=False):
    """nn.Linear: x (M,K) . w (N,K)^T + bias."""
    x = _f32(x); w = _f32(w)
    M, K = x.shape
    N = w.shape[0]
    xd = ctx.to_device(x); wd = ctx.to_device(w)
    bd = ctx.to_device(_f32(bias)) if bias is not None else None
    o = ctx.empty((M, N))
    check(ctx.h, ctx.lib.dc_op_linear(ctx.h, xd.ptr, wd.ptr, bd.ptr if bd else None, o.ptr, M, N, K, int(relu)),
          "dc_op_linear")
    return o.numpy()


# ---- box ops ------------------------------------------------------------------------------------
def make_anchors(ctx, h, w, x0, y0, sx, sy, anchors):
    anchors = _f32(anchors)
    k = anchors.shape[1]
    ad = ctx.to_device(anchors); o = ctx.empty((k * h * w, 4))
    check(ctx.h, ctx.lib.dc_op_make_anchors(ctx.h, o.ptr, h, w, x0, y0, sx, sy, ad.ptr, k), "dc_op_make_anchors")
    return o.numpy()


def apply_box_transform(ctx, boxes, trans):
    b = _f32(boxes).reshape(-1, 4); t = _f32(trans).reshape(-1, 4)
    bd = ctx.to_device(b); td = ctx.to_device(t); o = ctx.empty(b.shape)
    check(ctx.h, ctx.lib.dc_op_apply_box_transform(ctx.h, bd.ptr, td.ptr, o.ptr, b.shape[0]), "apply_box_transform")
    return o.numpy().reshape(np.shape(boxes))


def clip_boxes(ctx, boxes, bounds):
    """box_utils.clip_boxes(boxes, bounds, 'xcycwh') -> (clipped, valid)."""
    b = _f32(boxes).reshape(-1, 4)
    bd = ctx.to_device(b); o = ctx.empty(b.shape); v = ctx.empty((b.shape[0],), np.uint8)
    check(ctx.h, ctx.lib.dc_op_clip_boxes(ctx.h, bd.ptr, o.ptr, v.ptr, b.shape[0], bounds["x_min"], bounds["y_min"],
                                          bounds["x_max"], bounds["y_max"]), "dc_op_clip_boxes")
    return o.numpy().reshape(np.shape(boxes)), v.numpy().astype(bool)


def xcycwh_to_x1y1x2y2(ctx, boxes):
    b = _f32(boxes).reshape(-1, 4)
    bd = ctx.to_device(b); o = ctx.empty(b.shape)
    check(ctx.h, ctx.lib.dc_op_xcycwh_to_x1y1x2y2(ctx.h, bd.ptr, o.ptr, b.shape[0]), "xcycwh_to_x1y1x2y2")
    return o.numpy().reshape(np.shape(boxes))


def box_iou(ctx, b1, b2, convention=0):
    b1 = _f32(b1); b2 = _f32(b2)
    d1 = ctx.to_device(b1); d2 = ctx.to_device(b2); o = ctx.empty((b1.shape[0], b2.shape[0]))
    check(ctx.h, ctx.lib.dc_op_box_iou(ctx.h, d1.ptr, d2.ptr, o.ptr, b1.shape[0], b2.shape[0], convention), "box_iou")
    return o.numpy()


def rpn_decode(ctx, box_head, score_head, img_h, img_w, anchors, field_centers):
    """box_head (4k,h,w), score_head (2k,h,w) as the reference's RPN convs emit them."""
    k = anchors.shape[1]
    h, w = box_head.shape[1:]
    heads = np.concatenate([_f32(box_head), _f32(score_head)], 0).transpose(1, 2, 0)  # (h,w,6k)
    hd = ctx.to_device(np.ascontiguousarray(heads)); ad = ctx.to_device(_f32(anchors))
    A = k * h * w
    boxes = ctx.empty((A, 4)); anc = ctx.empty((A, 4)); trans = ctx.empty((A, 4)); xyxy = ctx.empty((A, 4))
    p = ctx.empty((A,)); valid = ctx.empty((A,), np.uint8)
    x0, y0, sx, sy = field_centers
    check(ctx.h, ctx.lib.dc_op_rpn_decode(ctx.h, hd.ptr, h, w, k, ad.ptr, x0, y0, sx, sy, img_h, img_w, boxes.ptr,
                                          anc.ptr, trans.ptr, xyxy.ptr, p.ptr, valid.ptr), "dc_op_rpn_decode")
    return dict(boxes=boxes.numpy(), anchors=anc.numpy(), trans=trans.numpy(), x1y1x2y2=xyxy.numpy(),
                p=p.numpy(), valid=valid.numpy().astype(bool))


def nms(ctx, boxes5, overlap, max_boxes=None, valid=None):
    """box_utils.nms(boxes(N,5), overlap, max_boxes) -> 0-based picks (decreasing score)."""
    b = _f32(boxes5)
    n = b.shape[0]
    if n == 0:
        return np.zeros((0,), np.int64)
    bd = ctx.to_device(np.ascontiguousarray(b[:, :4])); sd = ctx.to_device(np.ascontiguousarray(b[:, 4]))
    vd = ctx.to_device(np.ascontiguousarray(valid, dtype=np.uint8)) if valid is not None else None
    cap = n if max_boxes is None else min(n, int(max_boxes))
    picks = ctx.empty((max(cap, 1),), np.int32); cnt = ctx.empty((1,), np.int32)
    check(ctx.h, ctx.lib.dc_op_nms(ctx.h, bd.ptr, sd.ptr, vd.ptr if vd else None, n, C.c_float(float(np.float32(overlap))),
                                   -1 if max_boxes is None else int(max_boxes), picks.ptr, cnt.ptr), "dc_op_nms")
    kk = int(cnt.numpy()[0])
    return picks.numpy()[:kk].astype(np.int64)


def bilinear_roi_pool(ctx, feat_chw, boxes, img_h, img_w, HH=7, WW=7, out_layout=0):
    """nn.BilinearRoiPooling forward: (C,h,w)+(B,4) -> (B,C,HH,WW) [layout 0] or (B,HH,WW,C) [1]."""
    f = _f32(feat_chw); b = _f32(boxes)
    C_, h, w = f.shape
    fd = ctx.to_device(np.ascontiguousarray(f.transpose(1, 2, 0))); bd = ctx.to_device(b)
    B = b.shape[0]
    o = ctx.empty((B, C_, HH, WW) if out_layout == 0 else (B, HH, WW, C_))
    check(ctx.h, ctx.lib.dc_op_bilinear_roi_pool(ctx.h, fd.ptr, h, w, C_, bd.ptr, B, img_h, img_w, HH, WW, o.ptr,
                                                 out_layout), "dc_op_bilinear_roi_pool")
    return o.numpy()
