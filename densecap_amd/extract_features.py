"""extract_features.lua on the MI355X path.

Same flags and output datasets as the reference's `extract_features.lua` (flags :11-27, `run_image`
:30-44, main :47-99): for each image of `-input_txt`, `DenseCapModel:extractFeatures` (boxes + fc7 codes
after the final NMS, no LSTM decode), boxes converted to xywh, the first `-boxes_per_image` rows kept.

    python -m densecap_amd.extract_features -input_txt paths.txt -output_h5 feats.h5 -checkpoint model.t7

Output: an HDF5 file with datasets `/feats` (N, M, 4096) and `/boxes` (N, M, 4), fp32, as the reference writes
with torch-hdf5 (extract_features.lua:92-96); written by densecap_amd/hdf5_min.py (this image has no h5py) in the
classic contiguous layout, readable by libhdf5 / h5py.
"""
from __future__ import annotations

import argparse
import sys

import numpy as np

from .run_model import ImagePipeline, xcycwh_to_xywh


def build_parser():
    p = argparse.ArgumentParser(prefix_chars="-", description=__doc__,
                                formatter_class=argparse.RawDescriptionHelpFormatter)
    a = p.add_argument
    a("-checkpoint", default="data/models/densecap/densecap-pretrained-vgg16.t7")
    a("-image_size", type=int, default=720)
    a("-rpn_nms_thresh", type=float, default=0.7)
    a("-final_nms_thresh", type=float, default=0.4)
    a("-num_proposals", type=int, default=1000)
    a("-boxes_per_image", type=int, default=100)
    a("-input_txt", default="")
    a("-max_images", type=int, default=0)
    a("-output_h5", default="")
    a("-gpu", type=int, default=0)
    # ---- not reference flags ----
    a("-lanes", type=int, default=2, help="images in flight")
    a("-group", type=int, default=4, help="equal-sized images that share the dense launches (dc_set_group)")
    a("-io_threads", type=int, default=8, help="threads that decode the input files")
    a("-host_preprocess", type=int, default=0, help="1: image.scale on the host (NumPy restatement) instead of dc_preprocess_u8")
    a("-use_cudnn", type=int, default=1, help="accepted for compatibility (extract_features.lua:27); this path has no cuDNN / MIOpen to switch")
    a("-timing", type=int, default=0, help="1: print the images/s of the image loop at the end")
    a("-math_mode", type=int, default=0, choices=[0, 1],
      help="dc_set_math_mode: 0 = fp32 MFMA (default; the reference's arithmetic), 1 = split-bf16 (opt-in: six bf16 partial products per fp32 multiply-add on the bf16 matrix cores, fp32-class accuracy, ~1.2-1.3x images/s)")
    a("-synthetic_weights", type=int, default=0,
      help="1: random weights in checkpoint shapes (no pretrained .t7 is available offline)")
    return p


def write_datasets(path, feats, boxes):
    """extract_features.lua:92-96: `/feats` and `/boxes` in one HDF5 file (contiguous fp32 datasets)."""
    from .hdf5_min import write_hdf5
    return write_hdf5(path, {"feats": np.ascontiguousarray(feats, np.float32),
                             "boxes": np.ascontiguousarray(boxes, np.float32)})


def main(argv=None):
    opt = build_parser().parse_args(argv)
    if not opt.input_txt:
        raise SystemExit("Must provide -input_txt")
    if not opt.output_h5:
        raise SystemExit("Must provide -output_h5")
    with open(opt.input_txt) as f:
        paths = [ln.strip() for ln in f if ln.strip()]
    if opt.max_images > 0:
        paths = paths[:opt.max_images]
    from . import DenseCapModel
    if opt.synthetic_weights:
        from .weights import make_synthetic_weights
        weights = make_synthetic_weights()
    else:
        import os
        from . import t7
        if not os.path.exists(opt.checkpoint):
            raise SystemExit("checkpoint %s not found (use -synthetic_weights 1 for random weights)" % opt.checkpoint)
        weights = t7.weights_from_checkpoint(t7.load(opt.checkpoint))
    model = DenseCapModel(weights, device=opt.gpu)
    model.setMathMode(opt.math_mode)
    model.setLanes(1 if len(paths) == 1 else opt.lanes)   # a list of images is pipelined over the lanes (same results per image)
    model.setGroup(1 if len(paths) == 1 else opt.group)
    model.setTestArgs(rpn_nms_thresh=opt.rpn_nms_thresh, final_nms_thresh=opt.final_nms_thresh,
                      num_proposals=opt.num_proposals)
    model.evaluate()
    N, M = len(paths), opt.boxes_per_image
    all_boxes = np.zeros((N, M, 4), np.float32)
    all_feats = None
    import time
    t_loop = time.perf_counter()
    # extract_features.lua:79-91 as a pipeline: files decoded ahead on io threads, image.scale & co on the device
    pipe = ImagePipeline(paths, opt.image_size, opt.gpu, model.ctx, io_threads=opt.io_threads,
                         chunk=max(1, opt.lanes) * max(1, opt.group) * 2, host_preprocess=bool(opt.host_preprocess), want_rgb=False)
    try:
        for chunk in pipe:
            for i, _, _ in chunk:
                print("Processing image %d / %d" % (i + 1, N))
            outs = model.extractFeatures_images_device([d for _, d, _ in chunk])
            for (i, dev, _), (boxes_xcycwh, feats) in zip(chunk, outs):
                pipe.recycle(dev)
                if len(boxes_xcycwh) < M:     # the reference's boxes[{{1, M}}] raises on a short result as well
                    raise SystemExit("image %s: only %d boxes survive the final NMS, -boxes_per_image is %d"
                                     % (paths[i], len(boxes_xcycwh), M))
                if all_feats is None:
                    all_feats = np.zeros((N, M, feats.shape[1]), np.float32)
                all_boxes[i] = xcycwh_to_xywh(boxes_xcycwh)[:M]
                all_feats[i] = feats[:M]
    finally:
        pipe.close()
    if opt.timing:
        dt = time.perf_counter() - t_loop
        print("TIMING %d images in %.3f s = %.1f images/s (decode + preprocess + extractFeatures; lanes %d, group %d, %d io "
              "threads, %s preprocessing)" % (N, dt, N / max(dt, 1e-9), opt.lanes, opt.group, opt.io_threads,
                                              "host" if opt.host_preprocess else "device"))
    if all_feats is None:
        all_feats = np.zeros((0, M, 4096), np.float32)
    write_datasets(opt.output_h5, all_feats, all_boxes)
    return 0


if __name__ == "__main__":
    sys.exit(main())
