"""BASELINE.json configs[0] end to end on the GPU: `run_model.lua -input_image imgs/elephant.jpg` from a checkpoint
(run_model.lua:64-95,145-164) -- checkpoint `.t7` -> product reader -> dc_load_weights -> 720x480 JPEG -> 1000 proposals
-> results.json, against the CPU oracle.

The pretrained checkpoint does not exist offline, so the `.t7` is a checkpoint-SHAPED file (real VGG-16 / RPN / fc / LM
shapes, V = 10,497, T = 15, the reference's module tree incl. the nngraph recog_net, flat-storage parameter views) written
byte by byte by tests/golden/t7_assembler.py -- not by the product's T7Writer -- with the seeded synthetic weights in it.
The image is the reference's own imgs/elephant.jpg (720x480 -> 30x45 map, A = 16,200), kept as
tests/golden/elephant_720x480.jpg because /root/reference does not exist on the GPU box.
"""
import json
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
ELEPHANT = os.path.join(HERE, "golden", "elephant_720x480.jpg")


def _to_torch(W):
    import torch
    out = {}
    for k, v in W.items():
        if isinstance(v, np.ndarray):
            out[k] = torch.from_numpy(np.ascontiguousarray(v))
        elif isinstance(v, list) and v and isinstance(v[0], np.ndarray):
            out[k] = [torch.from_numpy(np.ascontiguousarray(a)) for a in v]
        else:
            out[k] = v
    return out


def test_config0_checkpoint_elephant_results_json(tmp_path):
    from PIL import Image
    from densecap_amd import DenseCapModel, run_model, t7
    from densecap_amd.weights import make_synthetic_weights
    from oracle import densecap_oracle as O
    from tests import parity
    from tests.golden.t7_assembler import assemble_densecap_checkpoint

    W = make_synthetic_weights(seed=1234)                         # V = 10,497, T = 15: the checkpoint's real shapes
    ckpt = tmp_path / "densecap-synthetic-vgg16.t7"
    assemble_densecap_checkpoint(str(ckpt), W)
    assert os.path.getsize(ckpt) > 550e6                          # 13 convs + RPN + fc6/fc7 + LM in fp32
    del W

    # ---- the product CLI, exactly as a user runs it (single image -> single-image mode, run_model.lua:145-164) ------
    vis = tmp_path / "vis"
    rc = run_model.main(["-checkpoint", str(ckpt), "-input_image", ELEPHANT, "-num_proposals", "1000",
                         "-rpn_nms_thresh", "0.7", "-final_nms_thresh", "0.3", "-image_size", "720",
                         "-output_vis_dir", str(vis), "-gpu", "0"])
    assert rc == 0
    out = json.load(open(vis / "results.json"))
    res = out["results"][0]
    assert res["img_name"] == "elephant_720x480.jpg" and os.path.exists(vis / "elephant_720x480.jpg")
    assert out["opt"]["num_proposals"] == 1000 and out["opt"]["checkpoint"] == str(ckpt)

    # ---- the oracle on the same file and image: its own preprocessing restatement and the product reader's weights ----
    Wc = t7.weights_from_checkpoint(t7.load(str(ckpt)))
    assert Wc["vocab_size"] == 10497 and Wc["seq_length"] == 15 and Wc["fc6_w"].shape == (4096, 25088)
    Wt = _to_torch(Wc)
    rgb01 = O.image_load_u8(np.asarray(Image.open(ELEPHANT).convert("RGB"), np.uint8))     # image.load: a DoubleTensor, byte / 255
    assert rgb01.shape == (3, 480, 720) and rgb01.dtype == np.float64
    img = O.preprocess(rgb01, 720)[0]                            # run_model.lua:68-74, scalar restatement
    x_host, _ = run_model.load_image_caffe(ELEPHANT, 720)
    np.testing.assert_array_equal(x_host[0], img)                 # host preprocessing == oracle, bit for bit

    m = DenseCapModel(Wc, device=0)
    try:
        m.setLanes(1)                                             # what run_model uses for one image
        r = parity.strict_check(m, Wt, img, 1000)                 # every stage, teacher-forced integer stages, final lists
        boxes, scores, tokens = m.forward_raw(img)
        caps = m.decodeSequence(tokens)
    finally:
        m.ctx.close()
    assert r["K"] > 0 and r["matched"] == r["K_oracle"]
    # results.json == what the model returns for this image (xywh boxes, run_model.lua:78,89-95)
    np.testing.assert_array_equal(np.asarray(res["boxes"], np.float32), run_model.xcycwh_to_xywh(boxes))
    np.testing.assert_array_equal(np.asarray(res["scores"], np.float32), scores)
    assert res["captions"] == caps and len(caps) == r["K"]

    # ---- results.json vs the ORACLE's results for the same command ----------------------------------------------------
    ob, os_, oseq = O.forward_test(img, Wt, 0.7, 0.3, 1000, 15)
    ocaps = O.decode_sequence(oseq, Wc["idx_to_token"], Wc["vocab_size"])
    oxywh = O.xcycwh_to_xywh(ob)
    flips = r.get("final_list_flips", []) + r.get("token_near_ties", [])
    if not flips:
        assert len(res["boxes"]) == len(ob)
        jb = np.asarray(res["boxes"], np.float64)
        assert (np.abs(jb - oxywh).max(axis=1) <= 1e-4 * np.maximum(1.0, np.abs(oxywh).max(axis=1))).all()
        js = np.asarray(res["scores"], np.float64)
        assert (np.abs(js - os_) <= 1e-4 * np.maximum(1.0, np.abs(os_))).all()
        assert res["captions"] == ocaps
    rep = dict(image="imgs/elephant.jpg (720x480)", checkpoint_bytes=os.path.getsize(ckpt), K=r["K"], K_oracle=r["K_oracle"],
               matched=r["matched"], trunk_rel_err=r["trunk_rel_err"], fc7_codes_rel_err=r["fc7_codes_rel_err"],
               final_boxes_rel_err=r["final_boxes_pre_nms_rel_err"], decode_rows_identical="%d/%d" % (
                   r["decode_rows_identical"], r["decode_rows"]), near_tie_departures=flips,
               captions_equal_oracle=res["captions"] == ocaps, first_captions=res["captions"][:3])
    outdir = os.path.join(os.path.dirname(HERE), "gpurun_out")
    os.makedirs(outdir, exist_ok=True)
    json.dump(rep, open(os.path.join(outdir, "config0_parity.json"), "w"), indent=1)
