"""Synthetic weights in the densecap checkpoint's shapes (seeded, CPU, fp32).

The pretrained `.t7` checkpoint of the reference is not available offline
(scripts/download_pretrained_model.sh), so every measurable configuration uses
random weights of the same architecture: VGG-16 conv1_1..conv5_3
(DenseCapModel.lua:61-67), RPN (LocalizationLayer.lua:609-690), fc6/fc7 +
objectness / box-regression heads (DenseCapModel.lua:93-100) and the language
model (LanguageModel.lua:10-74).  Scales are chosen so activations neither die
nor explode and the RPN / objectness logits have O(1) spread.
"""
from __future__ import annotations

import math

import numpy as np
import torch

VGG16_CONVS = [(3, 64), (64, 64), (64, 128), (128, 128), (128, 256), (256, 256), (256, 256),
               (256, 512), (512, 512), (512, 512), (512, 512), (512, 512), (512, 512)]
DEFAULT_ANCHORS = np.array([[45, 90], [90, 45], [64, 64], [90, 180], [180, 90], [128, 128],
                            [181, 362], [362, 181], [256, 256], [362, 724], [724, 362], [512, 512]],
                           dtype=np.float32).T.copy()  # (2,k) LocalizationLayer.lua:613-619
VGG_MEAN_BGR = (103.939, 116.779, 123.68)  # run_model.lua:73
# conv5_3 output RMS for U[0,255)-mean input with He-normal convs (measured once with the CPU
# oracle); dividing conv5_3 by it gives unit-RMS features for the RPN / RoI heads.
FEAT_RMS = 226.8


def make_synthetic_weights(seed=1234, vocab_size=10497, seq_length=15, rpn_hidden=256,
                           enc_size=512, rnn_size=512, fc_dim=4096, feat_rms=None):
    g = torch.Generator().manual_seed(seed)

    def randn(*shape, std=1.0):
        return (torch.randn(*shape, generator=g, dtype=torch.float32) * std).contiguous()

    feat_rms = FEAT_RMS if feat_rms is None else feat_rms
    W = {}
    conv_w, conv_b = [], []
    for li, (cin, cout) in enumerate(VGG16_CONVS):
        std = math.sqrt(2.0 / (9 * cin))
        w = randn(cout, cin, 3, 3, std=std)
        b = randn(cout, std=0.05)
        if li == len(VGG16_CONVS) - 1:
            w = w / feat_rms
            b = b / feat_rms
        conv_w.append(w.contiguous()); conv_b.append(b.contiguous())
    W["conv_w"], W["conv_b"] = conv_w, conv_b
    k = DEFAULT_ANCHORS.shape[1]
    W["rpn_conv_w"] = randn(rpn_hidden, 512, 3, 3, std=math.sqrt(2.0 / (9 * 512)))
    W["rpn_conv_b"] = randn(rpn_hidden, std=0.05)
    W["rpn_box_w"] = randn(4 * k, rpn_hidden, 1, 1, std=0.02)
    W["rpn_box_b"] = randn(4 * k, std=0.02)
    W["rpn_score_w"] = randn(2 * k, rpn_hidden, 1, 1, std=0.1)
    W["rpn_score_b"] = randn(2 * k, std=0.1)
    W["fc6_w"] = randn(fc_dim, 512 * 7 * 7, std=math.sqrt(2.0 / (512 * 49)))
    W["fc6_b"] = randn(fc_dim, std=0.05)
    W["fc7_w"] = randn(fc_dim, fc_dim, std=math.sqrt(2.0 / fc_dim))
    W["fc7_b"] = randn(fc_dim, std=0.05)
    W["obj_w"] = randn(1, fc_dim, std=0.1)
    W["obj_b"] = randn(1, std=0.1)
    W["boxreg_w"] = randn(4, fc_dim, std=0.002)
    W["boxreg_b"] = randn(4, std=0.01)
    W["lm_enc_w"] = randn(enc_size, fc_dim, std=math.sqrt(2.0 / fc_dim))
    W["lm_enc_b"] = randn(enc_size, std=0.05)
    W["lm_emb"] = randn(vocab_size + 2, enc_size, std=0.7)
    W["lstm_w"] = randn(enc_size + rnn_size, 4 * rnn_size, std=1.0 / math.sqrt(rnn_size))
    lb = randn(4 * rnn_size, std=0.05)
    lb[rnn_size:2 * rnn_size] += 1.0  # forget-gate bias
    W["lstm_b"] = lb.contiguous()
    W["lm_out_w"] = randn(vocab_size + 1, rnn_size, std=3.0 / math.sqrt(rnn_size))
    W["lm_out_b"] = randn(vocab_size + 1, std=0.1)
    W["anchors"] = torch.from_numpy(DEFAULT_ANCHORS.copy())
    W["field_centers"] = (8.5, 8.5, 16.0, 16.0)  # net_utils.compute_field_centers for VGG-16 layers 1..30
    W["vocab_size"] = vocab_size
    W["seq_length"] = seq_length
    W["idx_to_token"] = {i: "w%d" % i for i in range(1, vocab_size + 1)}
    return W


def make_synthetic_image(H=600, W=720, seed=0):
    """U[0,255) pixels, BGR, minus the VGG mean (the tensor run_model.lua:67-74 hands to
    forward_test).  Returns float32 numpy (3,H,W)."""
    g = torch.Generator().manual_seed(10_000 + seed)
    img = torch.rand(3, H, W, generator=g, dtype=torch.float32) * 255.0
    mean = torch.tensor(VGG_MEAN_BGR, dtype=torch.float32).view(3, 1, 1)
    return (img - mean).contiguous().numpy()


def check_weight_shapes(W):
    """dc_load_weights reads every tensor with the sizes the struct's dimension fields imply (include/densecap.h): the
    host must not hand it an array that is shorter.  Raises ValueError naming the first tensor that does not fit the
    architecture the others describe (DenseCapModel.lua:61-100, LocalizationLayer.lua:609-690, LanguageModel.lua:27-61).
    Called by DenseCapModel before any pointer crosses the C ABI -- a checkpoint of another architecture, or one whose
    tensor headers were damaged, ends here."""
    def shape(key, a=None):
        a = W[key] if a is None else a
        return tuple(int(v) for v in (a.shape if hasattr(a, "shape") else np.asarray(a).shape))

    def need(key, want, a=None):
        got = shape(key, a)
        if got != tuple(want):
            raise ValueError("%s has shape %s, the architecture needs %s" % (key, got, tuple(want)))

    if len(W["conv_w"]) != len(VGG16_CONVS) or len(W["conv_b"]) != len(VGG16_CONVS):
        raise ValueError("conv_w / conv_b: expected the %d VGG-16 convolutions" % len(VGG16_CONVS))
    for i, (cin, cout) in enumerate(VGG16_CONVS):
        need("conv_w[%d]" % i, (cout, cin, 3, 3), W["conv_w"][i])
        need("conv_b[%d]" % i, (cout,), W["conv_b"][i])
    a = shape("anchors")
    if len(a) != 2 or a[0] != 2 or a[1] < 1:
        raise ValueError("anchors has shape %s, need (2, k)" % (a,))
    k = a[1]
    R = shape("rpn_conv_w")[0] if len(shape("rpn_conv_w")) == 4 else -1
    need("rpn_conv_w", (R, 512, 3, 3)); need("rpn_conv_b", (R,))
    need("rpn_box_w", (4 * k, R, 1, 1)); need("rpn_box_b", (4 * k,))
    need("rpn_score_w", (2 * k, R, 1, 1)); need("rpn_score_b", (2 * k,))
    D = shape("fc7_w")[0] if len(shape("fc7_w")) == 2 else -1
    need("fc7_w", (D, D)); need("fc7_b", (D,))
    need("fc6_w", (D, 512 * 7 * 7)); need("fc6_b", (D,))
    need("obj_w", (1, D)); need("obj_b", (1,)); need("boxreg_w", (4, D)); need("boxreg_b", (4,))
    E = shape("lm_enc_w")[0] if len(shape("lm_enc_w")) == 2 else -1
    need("lm_enc_w", (E, D)); need("lm_enc_b", (E,))
    V = int(W["vocab_size"])
    need("lm_emb", (V + 2, E))
    Hd4 = shape("lstm_w")[1] if len(shape("lstm_w")) == 2 else -1
    if Hd4 <= 0 or Hd4 % 4:
        raise ValueError("lstm_w has shape %s, need (E + Hd, 4*Hd)" % (shape("lstm_w"),))
    Hd = Hd4 // 4
    need("lstm_w", (E + Hd, 4 * Hd)); need("lstm_b", (4 * Hd,))
    need("lm_out_w", (V + 1, Hd)); need("lm_out_b", (V + 1,))
    if len(W["field_centers"]) != 4 or int(W["seq_length"]) < 1 or V < 1:
        raise ValueError("field_centers / seq_length / vocab_size are not usable")
