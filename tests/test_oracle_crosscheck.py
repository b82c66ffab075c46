"""Independent second opinions on the oracle's restatements of the UN-VENDORED third-party ops
(stnbhwd sampler/grid, torch-rnn LSTM, THNN ceil-mode max-pool, LookupTable/Linear).

Torch7 cannot run here, so these stay "parity unpinned" against the Lua binaries; what these tests add is that
two separately written implementations of the published algorithms agree: the oracle's numpy/C code versus
PyTorch's own kernels (`grid_sample` implements the same spatial-transformer sampler as stnbhwd with
align_corners=True / zero padding; `nn.LSTMCell` is the same cell with gate order i,f,g,o)."""
import numpy as np
import pytest
import torch
import torch.nn.functional as Fn

from oracle import densecap_oracle as O


def test_bilinear_sampler_agrees_with_grid_sample():
    rng = np.random.default_rng(0)
    C, h, w, H, W = 24, 19, 23, 304, 368
    feat = rng.standard_normal((C, h, w)).astype(np.float32)
    B = 40
    boxes = np.concatenate([rng.uniform(-40, W + 40, (B, 1)), rng.uniform(-40, H + 40, (B, 1)),
                            rng.uniform(2, W, (B, 1)), rng.uniform(2, H, (B, 1))], 1).astype(np.float32)
    ours = O.bilinear_roi_pool(feat, boxes, H, W, 7, 7)                                # (B,C,7,7), C path
    theta = O.box_to_affine(boxes, H, W)                                               # rows (y), (x)
    grid_yx = O.affine_grid(theta, 7, 7)                                               # (B,7,7,2) last dim (y,x)
    grid_xy = torch.from_numpy(np.ascontiguousarray(grid_yx[..., ::-1]))               # grid_sample wants (x,y)
    ref = Fn.grid_sample(torch.from_numpy(feat)[None].expand(B, -1, -1, -1), grid_xy, mode="bilinear",
                         padding_mode="zeros", align_corners=True).numpy()
    np.testing.assert_allclose(ours, ref, atol=2e-5, rtol=0)
    np.testing.assert_allclose(O.bilinear_roi_pool_np(feat, boxes, H, W, 7, 7), ref, atol=2e-5, rtol=0)


def test_affine_grid_agrees_with_torch_affine_grid():
    rng = np.random.default_rng(1)
    boxes = np.concatenate([rng.uniform(0, 700, (16, 2)), rng.uniform(5, 500, (16, 2))], 1).astype(np.float32)
    theta = O.box_to_affine(boxes, 600, 720)                     # [[h/H,0,ty],[0,w/W,tx]]: output (y,x) of input (y,x,1)
    ours = O.affine_grid(theta, 7, 7)
    # torch.affine_grid maps (x,y,1) -> (x,y): permute rows and columns of theta accordingly
    t = torch.from_numpy(theta)
    t_xy = torch.stack([torch.stack([t[:, 1, 1], t[:, 1, 0], t[:, 1, 2]], 1),
                        torch.stack([t[:, 0, 1], t[:, 0, 0], t[:, 0, 2]], 1)], 1)
    ref = Fn.affine_grid(t_xy, (16, 1, 7, 7), align_corners=True).numpy()          # (B,7,7,2) (x,y)
    np.testing.assert_allclose(ours[..., ::-1], ref, atol=1e-6)


def test_lstm_step_agrees_with_lstm_cell():
    g = torch.Generator().manual_seed(2)
    D, Hd, N = 48, 32, 9
    Wx = torch.randn(D, 4 * Hd, generator=g) * 0.2        # torch-rnn layout: (D+H, 4H), gate slices i,f,o,g
    Wh = torch.randn(Hd, 4 * Hd, generator=g) * 0.2
    b = torch.randn(4 * Hd, generator=g) * 0.1
    x = torch.randn(N, D, generator=g); h = torch.randn(N, Hd, generator=g); c = torch.randn(N, Hd, generator=g)
    h2, c2 = O.lstm_step(b + x @ Wx, h, c, Wh)
    cell = torch.nn.LSTMCell(D, Hd)
    perm = torch.cat([torch.arange(0, Hd), torch.arange(Hd, 2 * Hd), torch.arange(3 * Hd, 4 * Hd),
                      torch.arange(2 * Hd, 3 * Hd)])           # torch order i,f,g,o  <-  torch-rnn i,f,o,g
    with torch.no_grad():
        cell.weight_ih.copy_(Wx.t()[perm]); cell.weight_hh.copy_(Wh.t()[perm])
        cell.bias_ih.copy_(b[perm]); cell.bias_hh.zero_()
        rh, rc = cell(x, (h, c))
    np.testing.assert_allclose(h2.numpy(), rh.numpy(), atol=2e-6)
    np.testing.assert_allclose(c2.numpy(), rc.numpy(), atol=2e-6)


def test_ceil_mode_maxpool_agrees_with_explicit_loop():
    rng = np.random.default_rng(3)
    for (h, w) in ((5, 7), (38, 45), (6, 6), (1, 3)):
        x = rng.standard_normal((1, 3, h, w)).astype(np.float32)
        got = Fn.max_pool2d(torch.from_numpy(x), 2, 2, ceil_mode=True).numpy()   # what oracle.vgg16_trunk calls
        oh, ow = (h + 1) // 2, (w + 1) // 2                                      # Caffe/loadcaffe ceil rule
        ref = np.empty((1, 3, oh, ow), np.float32)
        for i in range(oh):
            for j in range(ow):
                ref[:, :, i, j] = x[:, :, 2 * i:min(2 * i + 2, h), 2 * j:min(2 * j + 2, w)].max(axis=(2, 3))
        np.testing.assert_array_equal(got, ref)


def test_vgg_field_centers_match_pool_geometry():
    # net_utils.compute_field_centers (net_utils.lua:106-140): x0 = 8.5, stride 16 after four 2x2/2 pools
    x0, s = 1.0, 1.0
    for _ in range(4):
        x0 += s / 2.0
        s *= 2.0
    assert (x0, s) == (8.5, 16.0)
    assert tuple(O.VGG16_FIELD_CENTERS) == (8.5, 8.5, 16.0, 16.0)


def test_lm_sample_matches_stepwise_embedding_path():
    """lm_sample feeds Emb[tok] through Wx each step; the HIP path uses the precomputed table xg = b + Emb.Wx.
    Same association (b + x.Wx) + h.Wh, so the oracle's tokens must not change when it is driven that way."""
    from densecap_amd.weights import make_synthetic_weights
    W = make_synthetic_weights(seed=3, vocab_size=200, seq_length=5)
    codes = torch.relu(torch.randn(12, 4096, generator=torch.Generator().manual_seed(0)))
    seq = O.lm_sample(codes, W, 5)
    D = W["lstm_w"].shape[0] - W["lstm_w"].shape[1] // 4
    Wx, Wh = W["lstm_w"][:D], W["lstm_w"][D:]
    xg = W["lstm_b"] + W["lm_emb"] @ Wx
    enc = torch.relu(codes @ W["lm_enc_w"].t() + W["lm_enc_b"])
    h = torch.zeros(12, Wh.shape[0]); c = torch.zeros_like(h)
    h, c = O.lstm_step(W["lstm_b"] + enc @ Wx, h, c, Wh)
    tok = torch.full((12,), W["lm_out_w"].shape[0], dtype=torch.int64)
    for t in range(5):
        h, c = O.lstm_step(xg[tok - 1], h, c, Wh)
        tok = torch.argmax(h @ W["lm_out_w"].t() + W["lm_out_b"], dim=1) + 1
        assert tok.tolist() == seq[:, t].tolist()


def test_th_transcendentals_against_independent_float_implementations():
    """Round-5 advisor finding: the oracle and the device kernels both compute exp / sigmoid / tanh as (float)f((double)x)
    (the recalled TH CPU form, docs/SEMANTICS.md), so the bit-exact parity tests compare two sides that were changed
    together.  Here the oracle's three functions stand against implementations that share nothing with them: PyTorch's
    float32 kernels (vectorised SLEEF-style polynomials) and numpy's float32 libm.  A correctly rounded float result is
    within half an ulp of the truth, an independent float kernel within an ulp or two: 2 ulp apart at most -- which is also
    the bound on what the reference's own GPU path (cutorch float intrinsics, `-gpu 0`) would differ by."""
    import torch
    rng = np.random.default_rng(3)
    x = np.concatenate([rng.standard_normal(20000) * 4, rng.uniform(-80, 80, 20000), [0.0, -0.0, 1e-8, -1e-8, 30.0, -30.0, 88.0, -87.0]]).astype(np.float32)
    xt = torch.from_numpy(x)
    for mine, theirs_t, theirs_np in ((O.th_exp, torch.exp, np.exp), (lambda v: O.th_sigmoid(torch.from_numpy(v)).numpy(), torch.sigmoid, None),
                                      (lambda v: O.th_tanh(torch.from_numpy(v)).numpy(), torch.tanh, np.tanh)):
        got = np.asarray(mine(x), np.float32)
        assert got.dtype == np.float32
        for other in [theirs_t(xt).numpy()] + ([theirs_np(x).astype(np.float32)] if theirs_np else []):
            fin = np.isfinite(other) & np.isfinite(got) & (np.abs(other) > 1e-37)       # (denormal results: flush-to-zero differs)
            ulp = np.spacing(np.abs(other[fin]))
            assert (np.abs(got[fin] - other[fin]) <= 2 * ulp).all(), theirs_t.__name__
            assert np.array_equal(np.isinf(got), np.isinf(other))
    # the LSTM cell built from them against PyTorch's fused float cell is test_lstm_step_agrees_with_lstm_cell above


def test_rpn_probability_agrees_with_a_stable_softmax_away_from_overflow():
    """LocalizationLayer.lua:304-308 computes p(object) as pow(e1 + e2, -1) * e1 with no max subtraction.  Away from the
    overflow the restatement must be the two-class softmax: against torch.softmax in float64 on moderate logits."""
    import torch
    rng = np.random.default_rng(8)
    k, h, w = 12, 5, 7
    box_head = (rng.standard_normal((4 * k, h, w)) * 0.1).astype(np.float32)
    score_head = (rng.standard_normal((2 * k, h, w)) * 5).astype(np.float32)
    d = O.rpn_decode(box_head, score_head, 160, 224, clip_boxes=False)            # every row kept, in row order
    want = torch.softmax(torch.from_numpy(d["scores2"]).double(), dim=1)[:, 0].numpy()
    assert d["p"].shape == (k * h * w,) and np.abs(d["p"] - want).max() < 2e-7
    # which of the two score channels is "object": the FIRST of a pair (LocalizationLayer.lua:305 pos = scores[{{},1}])
    assert ((d["scores2"][:, 0] > d["scores2"][:, 1]) == (d["p"] > 0.5)).all()


def test_beam_search_of_width_one_is_a_greedy_walk_from_the_cell_seeded_state():
    """LM:beamsearch (LanguageModel.lua:170-290) with one beam, against a walk written out by hand: image step, START step,
    first word = arg-max, then -- LanguageModel.lua:221-226 as written -- BOTH states of the beam restart from the CELL state,
    arg-max of every further step until END (log-probabilities zeroed from then on: the first index wins)."""
    import torch
    from densecap_amd.weights import make_synthetic_weights
    Wt = make_synthetic_weights(seed=5, vocab_size=40, seq_length=6)
    Wt = {k_: (torch.from_numpy(v) if isinstance(v, np.ndarray) else v) for k_, v in Wt.items()}
    T, V1 = 6, 41
    Hd = Wt["lstm_w"].shape[1] // 4
    D = Wt["lstm_w"].shape[0] - Hd
    Wx, Wh = Wt["lstm_w"][:D], Wt["lstm_w"][D:]
    codes = torch.relu(torch.from_numpy(np.random.default_rng(1).standard_normal((9, 4096)).astype(np.float32)))
    got = O.lm_beamsearch(codes, Wt, T, 1)
    for i in range(len(codes)):
        enc = torch.relu(codes[i:i + 1] @ Wt["lm_enc_w"].t() + Wt["lm_enc_b"])
        h, c = O.lstm_step(Wt["lstm_b"] + enc @ Wx, torch.zeros(1, Hd), torch.zeros(1, Hd), Wh)
        h, c = O.lstm_step(Wt["lstm_b"] + Wt["lm_emb"][V1 - 1:V1] @ Wx, h, c, Wh)
        want = [int(torch.argmax(h @ Wt["lm_out_w"].t() + Wt["lm_out_b"])) + 1]
        h = c.clone()                                                      # the slip of :224, replicated
        for t in range(1, T):
            h, c = O.lstm_step(Wt["lstm_b"] + Wt["lm_emb"][want[-1] - 1:want[-1]] @ Wx, h, c, Wh)
            want.append(1 if V1 in want else int(torch.argmax(h @ Wt["lm_out_w"].t() + Wt["lm_out_b"])) + 1)
        assert got[i].tolist() == want, (i, got[i].tolist(), want)
