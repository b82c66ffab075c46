"""Torch7 `.t7` reader / checkpoint walk and run_model host helpers, CPU only."""
import io
import os

import numpy as np


def test_t7_roundtrip_primitives(tmp_path):
    from densecap_amd import t7
    shared = np.arange(6, dtype=np.float32).reshape(2, 3)
    obj = {"a": 1, "b": 2.5, "s": "hi", "flag": True, "none_is_skipped": 7, "list": [1, 2, 3],
           "t": shared, "t_again": shared, "long": np.array([1, 2, 3], np.int64),
           "obj": t7.TorchObject("nn.Linear", {"weight": np.ones((2, 2), np.float32), "bias": np.zeros(2, np.float32)})}
    p = tmp_path / "x.t7"
    t7.save(str(p), obj)
    back = t7.load(str(p))
    assert back["a"] == 1 and back["b"] == 2.5 and back["s"] == "hi" and back["flag"] is True
    assert t7._lua_list(back["list"]) == [1, 2, 3]
    np.testing.assert_array_equal(back["t"], shared)
    assert back["t_again"] is back["t"]                       # shared references are preserved
    assert back["long"].dtype == np.int64
    assert back["obj"].torch_type == "nn.Linear" and back["obj"]["weight"].shape == (2, 2)


def test_checkpoint_walk_roundtrip(tmp_path):
    from densecap_amd import t7
    from densecap_amd.weights import make_synthetic_weights
    W = make_synthetic_weights(seed=3, vocab_size=30, seq_length=4, fc_dim=256)
    # shrink the big tensors so the file stays small (shapes only need to be self-consistent)
    W["fc6_w"] = W["fc6_w"][:, :512]; W["idx_to_token"] = {i: "tok%d" % i for i in range(1, 31)}
    p = tmp_path / "ckpt.t7"
    t7.save(str(p), t7.checkpoint_from_weights(W))
    assert os.path.getsize(p) > 1_000_000
    back = t7.weights_from_checkpoint(t7.load(str(p)))
    for i in range(13):
        np.testing.assert_array_equal(back["conv_w"][i], W["conv_w"][i].numpy())
        np.testing.assert_array_equal(back["conv_b"][i], W["conv_b"][i].numpy())
    for k in ("rpn_conv_w", "rpn_box_w", "rpn_score_b", "fc6_w", "fc7_b", "obj_w", "boxreg_b", "lm_enc_w", "lm_emb",
              "lstm_w", "lstm_b", "lm_out_w", "lm_out_b", "anchors"):
        np.testing.assert_array_equal(back[k], np.asarray(W[k]))
    assert back["field_centers"] == (8.5, 8.5, 16.0, 16.0)
    assert back["vocab_size"] == 30 and back["seq_length"] == 4 and back["idx_to_token"][30] == "tok30"


def test_reads_hand_assembled_checkpoint_bytes():
    """tests/golden/handmade_checkpoint.t7 was assembled byte by byte from the Torch7 format by
    tests/golden/make_t7_fixture.py (not by T7Writer): closures in the three encodings are skipped, back-references,
    the nn.gModule with graph.Node objects, the legacy 2-D SpatialConvolutionMM weight, the strided/offset view and the
    shared storage are all resolved, and every tensor comes back bit for bit."""
    from densecap_amd import t7
    here = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    ck = t7.load(os.path.join(here, "handmade_checkpoint.t7"))
    exp = np.load(os.path.join(here, "handmade_checkpoint_expected.npz"))
    assert ck["iter"] == 20000 and ck["loss_history"] == {1: 2.5, 2: 2.25}
    model = ck["model"]
    assert model.torch_type == "nn.DenseCapModel"
    loc = model["nets"]["localization_layer"]
    for k in ("timer_hook", "legacy_hook", "old_hook"):
        assert isinstance(loc[k], t7.LuaFunction), k                    # skipped, not fatal
    assert loc["timer_hook"].upvalues == {1: {"name": "_ENV"}} and loc["image_height"] is None
    # the gModule's nodes hold the SAME objects as nets.* (back-references, not copies)
    g = model["nets"]["recog_net"]
    mods = {m.torch_type for m in t7.iter_modules(g)}
    assert {"nn.gModule", "nn.Sequential", "nn.Linear", "nn.LanguageModel", "nn.LSTM"} <= mods
    nodes = t7._lua_list(g["forwardnodes"])
    assert nodes[0]["data"]["module"] is model["nets"]["recog_base"]
    assert nodes[2]["data"]["module"] is model["nets"]["language_model"]
    W = t7.weights_from_checkpoint(ck)
    for li in range(13):
        np.testing.assert_array_equal(W["conv_w"][li], exp["conv%d_w" % li])       # incl. the 2-D legacy one and the view
        np.testing.assert_array_equal(W["conv_b"][li], exp["conv%d_b" % li])
    assert W["conv_w"][2].shape == (3, 2, 3, 3)
    for k in ("rpn_conv", "rpn_box", "rpn_score", "fc6", "fc7", "obj", "boxreg", "lm_enc", "lm_out"):
        np.testing.assert_array_equal(W[k + "_w"], exp[k + "_w"].reshape(W[k + "_w"].shape))
        np.testing.assert_array_equal(W[k + "_b"], exp[k + "_b"])
    np.testing.assert_array_equal(W["lm_emb"], exp["lm_emb"])                      # two tensors on one storage
    np.testing.assert_array_equal(W["lstm_b"], exp["lstm_b"])
    np.testing.assert_array_equal(W["lstm_w"], exp["lstm_w"])
    np.testing.assert_array_equal(W["anchors"], exp["anchors"])
    assert W["field_centers"] == (8.5, 8.5, 16.0, 16.0) and W["vocab_size"] == 5 and W["seq_length"] == 3
    assert W["idx_to_token"] == {i: "tok%d" % i for i in range(1, 6)}


def test_run_model_preprocessing_and_json(tmp_path):
    from PIL import Image
    from densecap_amd import run_model as R
    rng = np.random.default_rng(0)
    img = (rng.uniform(0, 255, (480, 720, 3))).astype(np.uint8)     # imgs/elephant.jpg is 720x480
    p = tmp_path / "a.png"
    Image.fromarray(img).save(p)
    x, rgb = R.load_image_caffe(str(p), 720)
    assert x.shape == (1, 3, 480, 720) and rgb.shape == (480, 720, 3)
    # BGR, x255, minus mean (run_model.lua:70-74)
    np.testing.assert_allclose(x[0, 0], img[:, :, 2].astype(np.float32) - 103.939, atol=1e-3)
    np.testing.assert_allclose(x[0, 2], img[:, :, 0].astype(np.float32) - 123.68, atol=1e-3)
    x2, _ = R.load_image_caffe(str(p), 360)
    assert x2.shape == (1, 3, 240, 360)
    xywh = R.xcycwh_to_xywh(np.array([[10.0, 20.0, 5.0, 7.0]], np.float32))
    np.testing.assert_allclose(xywh, [[8.0, 17.0, 5.0, 7.0]])
    j = R.result_to_json(xywh, np.array([[0.5]]), ["a cat"])
    assert j == {"boxes": [[8.0, 17.0, 5.0, 7.0]], "scores": [0.5], "captions": ["a cat"]}
    opt = R.build_parser().parse_args(["-input_image", "x.jpg", "-num_proposals", "300"])
    assert opt.rpn_nms_thresh == 0.7 and opt.final_nms_thresh == 0.3 and opt.num_proposals == 300


def test_image_scale_hand_computed_and_oracle():
    """image.scale (run_model.lua:68; torch/image scaleBilinear = scaleLinear_rowcol over rows, then columns):
    hand-computed 1-D cases, the size rule, and host == the oracle's scalar restatement bit for bit."""
    from densecap_amd import run_model as R
    from oracle import densecap_oracle as O
    f = lambda v, n: R._scale_linear_axis(np.array(v, np.float32), n, 0).tolist()
    assert f([1, 3, 5, 7], 2) == [2.0, 6.0]                    # shrink by 2: plain pair averages
    assert f([3, 6, 9], 2) == [4.0, 8.0]                       # shrink by 1.5: (s0 + .5 s1)/1.5, (.5 s1 + s2)/1.5
    assert f([2, 4], 3) == [2.0, 3.0, 4.0]                     # enlarge: samples at 0, .5, 1 (last copied)
    assert f([5], 4) == [5.0, 5.0, 5.0, 5.0]
    assert f([1, 2, 3], 3) == [1.0, 2.0, 3.0]
    np.testing.assert_allclose(f([0, 10, 20, 30], 7), [0, 5, 10, 15, 20, 25, 30], atol=1e-5)
    # 2-D: rows first then columns; a 2x4 image shrunk to max side 2 -> 1x2
    img = np.array([[[1, 3, 5, 7], [3, 5, 7, 9]]], np.float32)
    np.testing.assert_array_equal(R.image_scale(img, 2), [[[3.0, 7.0]]])
    # size rule: the LONGER side becomes `size`, the other is truncated (600x800 -> 540x720; 480x720 stays)
    assert R.image_scale(np.zeros((3, 600, 800), np.float32), 720).shape == (3, 540, 720)
    assert R.image_scale(np.zeros((3, 480, 720), np.float32), 720).shape == (3, 480, 720)
    assert R.image_scale(np.zeros((3, 333, 500), np.float32), 720).shape == (3, 479, 720)     # 333*720/500 = 479.52
    rng = np.random.default_rng(0)
    for (h, w, size) in [(7, 13, 5), (9, 5, 20), (40, 30, 17), (3, 50, 64)]:
        # image.load's DoubleTensor: bytes / 255 in double; the scaling runs in double with the library's float locals
        u8 = rng.integers(0, 256, (h, w, 3)).astype(np.uint8)
        x = R.image_load_u8(u8)
        assert x.dtype == np.float64 and np.array_equal(x, O.image_load_u8(u8))
        a, b = R.image_scale(x, size), O.image_scale(x, size)
        assert a.dtype == b.dtype == np.float64
        np.testing.assert_array_equal(a, b)
        np.testing.assert_array_equal(R.preprocess_rgb01(x, size)[0], O.preprocess(x, size))
        # the float accumulator of the area average is visible: every shrunk sample is a float value held in a double
        if size < max(h, w):
            assert np.array_equal(a, a.astype(np.float32).astype(np.float64))
        # ... and the all-float32 chain of rounds 2-5 stays within an ulp or two of it
        f32 = R.image_scale(x.astype(np.float32), size).astype(np.float32)
        assert np.abs(f32 - a.astype(np.float32)).max() <= 4 * np.finfo(np.float32).eps
    # shrinking is an area average, not point sampling: a 1-px checkerboard collapses to its mean
    cb = (np.indices((8, 8)).sum(0) % 2).astype(np.float32)[None]
    np.testing.assert_allclose(R.image_scale(cb, 4), np.full((1, 4, 4), 0.5), atol=1e-6)


def test_daemon_protocol_with_fake_model(tmp_path):
    """webcam/daemon.lua:55-102 host protocol: input consumed, <id>.json with boxes rescaled to the original size."""
    import json
    from PIL import Image
    from densecap_amd import daemon as D

    class FakeModel:
        def forward_test(self, img):
            assert img.shape == (1, 3, 360, 720)          # 1440x720 image scaled to max side 720
            return np.array([[10.5, 20.5, 4.0, 6.0]], np.float32), np.array([[1.0]], np.float32), ["a dog"]

    ind, outd = tmp_path / "in", tmp_path / "out"
    ind.mkdir()
    Image.fromarray(np.zeros((720, 1440, 3), np.uint8)).save(ind / "frame7.jpg")
    (ind / "broken.jpg").write_bytes(b"not a jpeg")
    opt = D.build_parser().parse_args(["-input_dir", str(ind), "-output_dir", str(outd), "-max_polls", "1"])
    D.serve(FakeModel(), opt)
    out = json.load(open(outd / "frame7.json"))
    assert out["height"] == 720 and out["width"] == 1440 and out["captions"] == ["a dog"]
    # xywh = (9, 18, 4, 6) in the 360-high frame -> x2 back to the original: ((9-1)*2+1, (18-1)*2+1, 8, 12)
    np.testing.assert_allclose(out["boxes"], [[17.0, 35.0, 8.0, 12.0]])
    assert not (ind / "frame7.jpg").exists() and (ind / "broken.jpg").exists()
    np.testing.assert_allclose(D.scale_boxes_xywh([[1, 1, 10, 10]], 1.5), [[1, 1, 15, 15]])


def test_reads_checkpoint_shaped_file_assembled_from_bytes(tmp_path):
    """tests/golden/t7_assembler.py writes what train.lua:157-185 saves (model.net first with the bodies, model.nets as
    back-references, the nn.gModule recog_net with nngraph.Node objects, LocalizationLayer's other nets / opt / closures
    (tag 8 with an index, tag 6 without), trainable parameters as OFFSET VIEWS of one flat storage like getParameters
    leaves them, empty post-clearState tensors, LongStorage fields, number-keyed idx_to_token).  The product reader must
    bring every tensor back bit for bit.  Reduced fc / vocabulary sizes here; the -m gpu test
    (test_gpu_config0.py) does the same at the real VGG-16 / V = 10,497 shapes and runs the file through the HIP path."""
    from densecap_amd import t7
    from densecap_amd.weights import make_synthetic_weights
    from tests.golden.t7_assembler import assemble_densecap_checkpoint
    W = make_synthetic_weights(seed=5, vocab_size=40, seq_length=6, fc_dim=256)
    W["idx_to_token"] = {i: "tok%d" % i for i in range(1, 41)}
    p = tmp_path / "ckpt.t7"
    nobj = assemble_densecap_checkpoint(str(p), W)
    assert nobj > 400
    ck = t7.load(str(p))
    assert ck["iter"] == 620000 and ck["results_history"][620000]["ap_results"]["map"] == 0.057
    model = ck["model"]
    nets = model["nets"]
    mods = t7._lua_list(model["net"]["modules"])
    assert mods[0] is nets["conv_net1"] and mods[2] is nets["localization_layer"] and mods[3] is nets["recog_net"]
    loc = nets["localization_layer"]
    assert isinstance(loc["timer_hook"], t7.LuaFunction) and isinstance(loc["old_hook"], t7.LuaFunction)
    assert loc["rpn_out"] is None and loc["roi_boxes"].size == 0
    nodes = t7._lua_list(nets["recog_net"]["forwardnodes"])
    assert nodes[4]["data"]["module"] is nets["recog_base"] and nodes[10]["data"]["module"] is nets["language_model"]
    assert nodes[5]["data"]["module"] is nets["objectness_branch"] and nodes[8]["data"]["module"] is nets["box_reg_branch"]
    lm = nets["language_model"]
    assert t7._lua_list(t7._lua_list(lm["net"]["modules"])[0]["modules"])[0] is lm["image_encoder"]
    B = t7.weights_from_checkpoint(ck)
    for i in range(13):
        np.testing.assert_array_equal(B["conv_w"][i], W["conv_w"][i].numpy())
        np.testing.assert_array_equal(B["conv_b"][i], W["conv_b"][i].numpy())
    for k in ("rpn_conv_w", "rpn_conv_b", "rpn_box_w", "rpn_box_b", "rpn_score_w", "rpn_score_b", "fc6_w", "fc6_b", "fc7_w",
              "fc7_b", "obj_w", "obj_b", "boxreg_w", "boxreg_b", "lm_enc_w", "lm_enc_b", "lm_emb", "lstm_w", "lstm_b",
              "lm_out_w", "lm_out_b", "anchors"):
        np.testing.assert_array_equal(B[k], np.asarray(W[k]), err_msg=k)
    assert B["field_centers"] == (8.5, 8.5, 16.0, 16.0) and B["vocab_size"] == 40 and B["seq_length"] == 6
    assert B["idx_to_token"] == W["idx_to_token"]
