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
