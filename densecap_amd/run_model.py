"""run_model.lua on the MI355X path.

Same flags, preprocessing and result JSON as the reference's `run_model.lua` (flags :26-61,
`run_image` :64-87, `result_to_json` :89-95, main loop :145-188); only the model object differs.

    python -m densecap_amd.run_model -input_image imgs/elephant.jpg -checkpoint model.t7
    python -m densecap_amd.run_model -input_dir imgs -synthetic_weights 1      # no checkpoint offline

File decode is third-party in the reference too (torch/image over libjpeg); PIL does it here.  `image.scale` is
restated from torch/image's C source (`image_scale` below: separable, interpolating when enlarging, area-averaging when
shrinking -- not PIL's resize) and checked against hand-computed values and the oracle's scalar restatement; no Torch7
binary exists here to pin it further.
"""
from __future__ import annotations

import argparse
import json
import os
import shutil
import sys

import numpy as np

VGG_MEAN_BGR = np.array([103.939, 116.779, 123.68], np.float32)   # run_model.lua:73


def build_parser():
    p = argparse.ArgumentParser(prefix_chars="-", description=__doc__,
                                formatter_class=argparse.RawDescriptionHelpFormatter)
    a = p.add_argument
    a("-checkpoint", default="data/models/densecap/densecap-pretrained-vgg16.t7")
    a("-image_size", type=int, default=720)
    a("-rpn_nms_thresh", type=float, default=0.7)
    a("-final_nms_thresh", type=float, default=0.3)
    a("-num_proposals", type=int, default=1000)
    a("-input_image", default="")
    a("-input_dir", default="")
    a("-max_images", type=int, default=100)
    a("-output_vis", type=int, default=1)
    a("-output_vis_dir", default="vis/data")
    a("-gpu", type=int, default=0)
    a("-lanes", type=int, default=2, help="images in flight when a directory is processed (not a reference flag)")
    a("-synthetic_weights", type=int, default=0,
      help="1: random weights in checkpoint shapes (no pretrained .t7 is available offline)")
    return p


def _scale_linear_axis(src, dst_len, axis):
    """torch/image generic/image.c `scaleLinear_rowcol` along one axis, float32 like the library:
    longer output  -> linear interpolation at di*(src_len-1)/(dst_len-1), last sample copied;
    shorter output -> area average: output di integrates the source over [di*s, (di+1)*s), s = src_len/dst_len, with
                      fractional end weights, divided by the accumulated weight;
    equal          -> copy.  Vectorised over the other axes; the order of the fp32 operations along the axis is the
    library's (accumulate, then divide)."""
    src = np.moveaxis(np.asarray(src, np.float32), axis, 0)
    src_len = src.shape[0]
    F = np.float32
    if dst_len == src_len:
        out = src.copy()
    elif dst_len > src_len:
        out = np.empty((dst_len,) + src.shape[1:], np.float32)
        if src_len == 1:
            out[:] = src[0]
        else:
            scale = F(src_len - 1) / F(dst_len - 1)
            for di in range(dst_len - 1):
                si_f = F(di) * scale
                si_i = int(si_f)
                si_f = F(si_f - F(si_i))
                out[di] = (F(1) - si_f) * src[si_i] + si_f * src[si_i + 1]
            out[dst_len - 1] = src[src_len - 1]
    else:
        out = np.empty((dst_len,) + src.shape[1:], np.float32)
        scale = F(src_len) / F(dst_len)
        si0_i, si0_f = 0, F(0)
        for di in range(dst_len):
            si1_f = F(di + 1) * scale
            si1_i = int(si1_f)
            si1_f = F(si1_f - F(si1_i))
            acc = (F(1) - si0_f) * src[si0_i]
            n = F(1) - si0_f
            for si in range(si0_i + 1, si1_i):
                acc = acc + src[si]
                n = F(n + F(1))
            if si1_i < src_len:
                acc = acc + si1_f * src[si1_i]
                n = F(n + si1_f)
            out[di] = acc / n
            si0_i, si0_f = si1_i, si1_f
    return np.moveaxis(out, 0, axis)


def image_scale(img_chw, size):
    """torch/image `image.scale(src, size)` with a number (run_model.lua:68): the LONGER side becomes `size`, the
    other keeps the aspect ratio (height = iheight*size/imax, truncated like a Lua number handed to Tensor:resize);
    mode 'bilinear' = `scaleBilinear`: rows first (width), then columns (height), each with `scaleLinear_rowcol`.
    img_chw: (C,H,W) float32.  torch/image is not vendored in the reference: restated from its published C source."""
    img = np.asarray(img_chw, np.float32)
    _, ih, iw = img.shape
    imax = max(ih, iw)
    oh, ow = int(ih * size / imax), int(iw * size / imax)
    if oh < 1 or ow < 1:
        raise ValueError("image.scale: %dx%d -> %dx%d" % (iw, ih, ow, oh))
    tmp = _scale_linear_axis(img, ow, 2)          # compress/expand rows first
    return _scale_linear_axis(tmp, oh, 1)         # then columns


def preprocess_rgb01(img_rgb_chw, image_size):
    """run_model.lua:68-74 after image.load: scale, RGB->BGR, x255, minus the VGG mean.  (3,H,W) in [0,1] -> (1,3,H',W')."""
    img = image_scale(img_rgb_chw, image_size)
    bgr = img[::-1] * np.float32(255.0) - VGG_MEAN_BGR[:, None, None]
    return np.ascontiguousarray(bgr[None], dtype=np.float32), img


def load_image_caffe(path, image_size):
    """run_model.lua:67-74: image.load(path, 3) (float RGB in [0,1] = byte/255), image.scale (max side = image_size),
    BGR, x255, minus VGG mean.  Returns (img_caffe (1,3,H,W) float32, scaled RGB uint8 (H,W,3) for the visualiser).
    Only the file DECODE is third-party here (PIL); the resize is image_scale above, not PIL's."""
    from PIL import Image
    im = Image.open(path).convert("RGB")
    x = np.asarray(im, dtype=np.uint8).astype(np.float32).transpose(2, 0, 1) / np.float32(255.0)
    img_caffe, scaled = preprocess_rgb01(x, image_size)
    # image.save clamps to [0,1] and writes bytes (x255, truncated)
    rgb = (np.clip(scaled, 0, 1) * 255.0).astype(np.uint8).transpose(1, 2, 0)
    return img_caffe, np.ascontiguousarray(rgb)


def xcycwh_to_xywh(boxes):
    """box_utils.xcycwh_to_xywh (box_utils.lua:441-445), host side like run_model.lua:78."""
    b = np.asarray(boxes, np.float32).reshape(-1, 4)
    hw = (b[:, 2] - np.float32(1)) / np.float32(2)
    hh = (b[:, 3] - np.float32(1)) / np.float32(2)
    x1 = -hw + b[:, 0]; y1 = -hh + b[:, 1]; x2 = hw + b[:, 0]; y2 = hh + b[:, 1]
    return np.stack([x1, y1, x2 - x1 + np.float32(1), y2 - y1 + np.float32(1)], 1)


def get_input_images(opt):
    if opt.input_image:
        return [opt.input_image]
    if opt.input_dir:
        return [os.path.join(opt.input_dir, fn) for fn in sorted(os.listdir(opt.input_dir))
                if not fn.startswith(".")]
    raise SystemExit("one of input_image or input_dir must be provided.")


def result_to_json(boxes_xywh, scores, captions):
    return dict(boxes=[[float(v) for v in r] for r in boxes_xywh],
                scores=[float(s) for s in np.asarray(scores).reshape(-1)], captions=list(captions))


def main(argv=None):
    opt = build_parser().parse_args(argv)
    from . import DenseCapModel
    if opt.synthetic_weights:
        from .weights import make_synthetic_weights
        weights = make_synthetic_weights()
    else:
        from . import t7
        if not os.path.exists(opt.checkpoint):
            raise SystemExit("checkpoint %s not found (use -synthetic_weights 1 for random weights)" % opt.checkpoint)
        try:
            ck = t7.load(opt.checkpoint)
        except t7.T7FormatError as e:
            # the two checks that rest on torch.save's habits (object numbering, nothing after the object) must not lock a
            # user out of an unusual but well-formed file: say so and read it again with only the bounds checks on
            print("warning: %s -- reading %s again without the writer-habit checks" % (e, opt.checkpoint), file=sys.stderr)
            ck = t7.load(opt.checkpoint, strict=False)
        weights = t7.weights_from_checkpoint(ck)
    model = DenseCapModel(weights, device=opt.gpu)                      # utils.setup_gpus + model:convert
    paths = get_input_images(opt)
    num = min(len(paths), opt.max_images)
    # one image: single-image mode (lowest latency, like the reference); a directory: its images -- whatever their
    # sizes -- are pipelined over the lanes in chunks (dc_forward_images), results identical to one-by-one processing
    model.setLanes(1 if num == 1 else opt.lanes)
    model.setTestArgs(rpn_nms_thresh=opt.rpn_nms_thresh, final_nms_thresh=opt.final_nms_thresh,
                      num_proposals=opt.num_proposals)
    model.evaluate()
    results = []
    CHUNK = 16
    for k0 in range(0, num, CHUNK):
        chunk = paths[k0:min(k0 + CHUNK, num)]
        pre = []
        for j, path in enumerate(chunk):
            print("%d/%d processing image %s" % (k0 + j + 1, num, path))
            pre.append(load_image_caffe(path, opt.image_size))
        outs = model.forward_images([p[0] for p in pre])
        for path, (img_caffe, rgb), (boxes, scores, tokens) in zip(chunk, pre, outs):
            rj = result_to_json(xcycwh_to_xywh(boxes), scores, model.decodeSequence(tokens))
            if opt.output_vis == 1:
                os.makedirs(opt.output_vis_dir, exist_ok=True)
                from PIL import Image
                Image.fromarray(rgb).save(os.path.join(opt.output_vis_dir, os.path.basename(path)))
                rj["img_name"] = os.path.basename(path)
                results.append(rj)
    if results:
        out = dict(results=results, opt={k: v for k, v in vars(opt).items()})
        with open(os.path.join(opt.output_vis_dir, "results.json"), "w") as f:
            json.dump(out, f)
    return 0


if __name__ == "__main__":
    sys.exit(main())
