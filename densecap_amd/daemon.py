"""webcam/daemon.lua on the MI355X path (the second caller of the boundary).

File-drop protocol of the reference (webcam/daemon.lua:55-102): poll `-input_dir` for `*.jpg`, run
forward_test, rescale boxes to the ORIGINAL image size (`box_utils.scale_boxes_xywh`, box_utils.lua:459-467),
write `<id>.json` {boxes, captions, height, width} to `-output_dir`, delete the input; sleep 50 ms.

    python -m densecap_amd.daemon -input_dir webcam/inputs -output_dir webcam/outputs -synthetic_weights 1
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

from .run_model import load_image_caffe, xcycwh_to_xywh


def build_parser():
    p = argparse.ArgumentParser(prefix_chars="-", description=__doc__,
                                formatter_class=argparse.RawDescriptionHelpFormatter)
    a = p.add_argument
    a("-checkpoint", default="data/models/densecap/densecap-pretrained-vgg16.t7")
    a("-max_image_size", type=int, default=720)
    a("-input_dir", default="webcam/inputs")
    a("-input_ext", default=".jpg")
    a("-output_dir", default="webcam/outputs")
    a("-rpn_nms_thresh", type=float, default=0.7)
    a("-final_nms_thresh", type=float, default=0.3)
    a("-num_proposals", type=int, default=1000)
    a("-gpu", type=int, default=0)
    a("-synthetic_weights", type=int, default=0)
    a("-beam_size", type=int, default=0,
      help="language_model.beam_size (LanguageModel.lua:129-131): 0 = greedy sampling, n = beam search")
    a("-timing", type=int, default=0, help="1: print the wall time of every frame (decode + preprocess + forward + JSON)")
    a("-use_cudnn", type=int, default=1, help="accepted for compatibility (webcam/daemon.lua:26); this path has no cuDNN / MIOpen to switch")
    a("-max_polls", type=int, default=-1, help="stop after this many directory polls (-1 = forever)")
    a("-math_mode", type=int, default=0, choices=[0, 1],
      help="dc_set_math_mode: 0 = fp32 MFMA (default; the reference's arithmetic), 1 = split-bf16 (opt-in: six bf16 partial products per fp32 multiply-add on the bf16 matrix cores, fp32-class accuracy, ~1.2-1.3x images/s)")
    a("-host_preprocess", type=int, default=0, help="1 = image.scale & co on the host (the Python restatement) instead of dc_preprocess_u8")
    a("-caption_order", type=int, default=1, choices=[0, 1],
      help="dc_set_caption_order: 1 (default) = final NMS first, captions only for the boxes it keeps (the reference's outputs bit "
           "for bit); 0 = the reference's order (caption every proposal, then the final NMS)")
    a("-graph_replay", type=int, default=0,
      help="dc_set_graph_replay: frames of one size are captured once and relaunched as a hipGraph (bit-identical).  Off by default: "
           "measured at the webcam settings it changes nothing (4.23 against 4.21 ms per frame, profiles/r05_daemon_latency.json)")
    return p


def scale_boxes_xywh(boxes, frac):
    """box_utils.scale_boxes_xywh (box_utils.lua:459-467): 1-based -> 0-based, scale, back to 1-based."""
    b = np.array(boxes, dtype=np.float32, copy=True).reshape(-1, 4)
    b[:, :2] -= np.float32(1)
    b *= np.float32(frac)
    b[:, :2] += np.float32(1)
    return b


def process_file(model, in_path, out_path, max_image_size, host_preprocess=False):
    """One iteration of daemon.lua:57-100.  Returns True if an output was written.

    The frame is decoded on the host and scaled / BGR-ed / mean-subtracted ON THE DEVICE (dc_preprocess_u8, bit-equal to the host
    restatement of image.scale: tests/test_gpu_preprocess.py) -- the Python resize took 30-190 ms per frame in front of a forward
    of 2 ms (round-4 verdict, item 4).  host_preprocess (or a model without the device entry points) keeps the host route."""
    device = not host_preprocess and hasattr(model, "forward_images_device")
    try:
        if device:
            from . import ops
            from .run_model import _decode_file
            rgb0 = _decode_file(in_path)              # image.load(path, 3): RGB bytes
            ori_h, ori_w = int(rgb0.shape[0]), int(rgb0.shape[1])
        else:
            from PIL import Image
            with Image.open(in_path) as im:
                ori_w, ori_h = im.size
            img_caffe, _ = load_image_caffe(in_path, max_image_size)
    except Exception:                                 # pcall(image.load) failed: leave the file, try again later
        return False
    if device:
        # the preprocessed frame's device buffer is kept per size (a webcam delivers one size: hipMalloc / hipFree would
        # synchronise the device once per frame)
        bufs = model.__dict__.setdefault("_daemon_frame_bufs", {})
        Hs, Ws = ops.preprocess_size(model.ctx.lib, ori_h, ori_w, max_image_size)
        dev = bufs.get((Hs, Ws))
        if dev is None:
            if len(bufs) >= 4:                        # sizes keep changing: do not hoard device memory
                for b in bufs.values():
                    b.free()
                bufs.clear()
            dev = bufs[(Hs, Ws)] = model.ctx.empty((3, Hs, Ws), np.float32)
        ops.preprocess_u8(model.ctx, rgb0, max_image_size, want_rgb=False, out=dev)
        H = Hs
        boxes, scores, tokens = model.forward_images_device([dev])[0]
        captions = model.decodeSequence(tokens)
    else:
        H = img_caffe.shape[2]
        boxes, _scores, captions = model.forward_test(img_caffe)
    boxes_xywh = scale_boxes_xywh(xcycwh_to_xywh(boxes), float(ori_h) / H)
    out = dict(boxes=[[float(v) for v in r] for r in boxes_xywh], captions=list(captions), height=ori_h, width=ori_w)
    os.remove(in_path)
    tmp = out_path + ".tmp"
    with open(tmp, "w") as f:
        json.dump(out, f)
    os.replace(tmp, out_path)
    return True


def serve(model, opt):
    polls = 0
    os.makedirs(opt.output_dir, exist_ok=True)
    while opt.max_polls < 0 or polls < opt.max_polls:
        for fn in sorted(os.listdir(opt.input_dir)):
            if not fn.endswith(opt.input_ext):
                continue
            in_path = os.path.join(opt.input_dir, fn)
            out_path = os.path.join(opt.output_dir, fn[:-len(opt.input_ext)] + ".json")
            print("Running model on image " + in_path)
            t0 = time.perf_counter()
            ok = process_file(model, in_path, out_path, opt.max_image_size, bool(getattr(opt, "host_preprocess", 0)))
            if ok and getattr(opt, "timing", 0):
                print("  %.2f ms" % ((time.perf_counter() - t0) * 1e3))
        polls += 1
        time.sleep(0.05)


def main(argv=None):
    opt = build_parser().parse_args(argv)
    from . import DenseCapModel
    if opt.synthetic_weights:
        from .weights import make_synthetic_weights
        weights = make_synthetic_weights()
    else:
        from . import t7
        weights = t7.weights_from_checkpoint(t7.load(opt.checkpoint))
    model = DenseCapModel(weights, device=opt.gpu)
    model.setLanes(1)      # one image at a time, like the reference: single-image mode has the lowest latency
    model.evaluate()
    model.setTestArgs(num_proposals=opt.num_proposals, rpn_nms_thresh=opt.rpn_nms_thresh,
                      final_nms_thresh=opt.final_nms_thresh)
    model.setBeamSize(opt.beam_size)
    model.setMathMode(opt.math_mode)
    model.setCaptionOrder(bool(opt.caption_order))
    model.setGraphReplay(bool(opt.graph_replay))
    serve(model, opt)
    return 0


if __name__ == "__main__":
    sys.exit(main())
