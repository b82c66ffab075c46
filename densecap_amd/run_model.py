"""run_model.lua on the MI355X path.

Same flags, preprocessing and result JSON as the reference's `run_model.lua` (flags :26-61,
`run_image` :64-87, `result_to_json` :89-95, main loop :145-188); only the model object differs.

    python -m densecap_amd.run_model -input_image imgs/elephant.jpg -checkpoint model.t7
    python -m densecap_amd.run_model -input_dir imgs -synthetic_weights 1      # no checkpoint offline

Decode + `image.scale` are host-side third-party code in the reference (torch/image: libjpeg + its own
bilinear); here PIL decodes and resizes (bilinear, max side = -image_size).  Pixel-exact agreement with
torch/image is unpinned (no Torch7 here); everything after the resize is the checked path.
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
    a("-synthetic_weights", type=int, default=0,
      help="1: random weights in checkpoint shapes (no pretrained .t7 is available offline)")
    return p


def load_image_caffe(path, image_size):
    """run_model.lua:67-74: load RGB [0,1], scale max side to image_size, BGR, x255, minus VGG mean.
    Returns (img_caffe (1,3,H,W) float32, scaled RGB uint8 (H,W,3))."""
    from PIL import Image
    im = Image.open(path).convert("RGB")
    w, h = im.size
    s = float(image_size) / max(w, h)
    nw, nh = max(1, int(round(w * s))), max(1, int(round(h * s)))
    im = im.resize((nw, nh), Image.BILINEAR)
    rgb = np.asarray(im, dtype=np.uint8)
    x = rgb.astype(np.float32) / 255.0
    bgr = x[:, :, ::-1].transpose(2, 0, 1) * 255.0 - VGG_MEAN_BGR[:, None, None]
    return np.ascontiguousarray(bgr[None], dtype=np.float32), rgb


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
        weights = t7.weights_from_checkpoint(t7.load(opt.checkpoint))
    model = DenseCapModel(weights, device=opt.gpu)                      # utils.setup_gpus + model:convert
    model.setLanes(1)      # one image at a time, like the reference: single-image mode has the lowest latency
    model.setTestArgs(rpn_nms_thresh=opt.rpn_nms_thresh, final_nms_thresh=opt.final_nms_thresh,
                      num_proposals=opt.num_proposals)
    model.evaluate()
    paths = get_input_images(opt)
    num = min(len(paths), opt.max_images)
    results = []
    for k in range(num):
        path = paths[k]
        print("%d/%d processing image %s" % (k + 1, num, path))
        img_caffe, rgb = load_image_caffe(path, opt.image_size)
        boxes, scores, captions = model.forward_test(img_caffe)
        rj = result_to_json(xcycwh_to_xywh(boxes), scores, captions)
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
