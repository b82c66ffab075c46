"""Stage-wise (teacher-forced) parity of the HIP kernels against the CPU oracle.

Every test calls the kernels through the C ABI (densecap_amd.ops -> libdensecap_hip.so).
Integer outputs (NMS picks, valid flags) must be bit-exact; fp32 outputs within the
north-star tolerance 1e-4 relative (tighter where the arithmetic is order-identical)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

REL = 1e-4  # BASELINE.json north_star: boxes/scores within 1e-4 relative fp32


@pytest.fixture(scope="module")
def ctx():
    from densecap_amd.ops import Context
    c = Context(0)
    yield c
    c.close()


def _close(a, b, rel=REL):
    scale = max(float(np.abs(b).max()), 1e-30)
    err = float(np.abs(a - b).max())
    assert err <= rel * scale, "max abs err %g vs scale %g" % (err, scale)


# ---- dense contractions --------------------------------------------------------------------------
@pytest.mark.parametrize("shape", [(1, 32, 9, 11, 64), (1, 64, 38, 45, 72), (2, 64, 20, 17, 128),
                                   (1, 128, 75, 90, 256), (1, 512, 38, 45, 256), (1, 64, 150, 180, 64)])
def test_conv3x3_matches_torch(ctx, shape):
    import torch
    from densecap_amd import ops
    N, Cin, H, W, Cout = shape
    g = torch.Generator().manual_seed(sum(shape))
    x = torch.randn(N, Cin, H, W, generator=g)
    w = torch.randn(Cout, Cin, 3, 3, generator=g) * (2.0 / (9 * Cin)) ** 0.5
    b = torch.randn(Cout, generator=g)
    ref = torch.relu(torch.nn.functional.conv2d(x.double(), w.double(), b.double(), padding=1)).float().numpy()
    out = ops.conv3x3(ctx, x.numpy(), w.numpy(), b.numpy(), relu=True)
    _close(out, ref)
    out2 = ops.conv3x3(ctx, x.numpy(), w.numpy(), b.numpy(), relu=False)
    ref2 = torch.nn.functional.conv2d(x.double(), w.double(), b.double(), padding=1).float().numpy()
    _close(out2, ref2)


@pytest.mark.parametrize("hw", [(67, 83), (3, 33), (600, 720), (1203, 1601)])
def test_conv1_1_c3_matches_torch(ctx, hw):
    """conv1_1 walks several 4x32 tiles per workgroup once the image has more tiles than one round of workgroups
    (720x600: 5 per workgroup, 1601x1203: 20): ragged right / bottom edges, one-tile images and both walks."""
    import torch
    from densecap_amd import ops
    g = torch.Generator().manual_seed(3)
    x = torch.rand(1, 3, *hw, generator=g) * 255 - 110
    w = torch.randn(64, 3, 3, 3, generator=g) * 0.27
    b = torch.randn(64, generator=g)
    ref = torch.relu(torch.nn.functional.conv2d(x.double(), w.double(), b.double(), padding=1)).float().numpy()
    _close(ops.conv3x3(ctx, x.numpy(), w.numpy(), b.numpy(), relu=True), ref)


@pytest.mark.parametrize("hw", [(8, 8), (75, 90), (7, 13), (1, 5)])
def test_maxpool_ceil(ctx, hw):
    import torch
    from densecap_amd import ops
    x = torch.randn(2, 64, *hw, generator=torch.Generator().manual_seed(1))
    ref = torch.nn.functional.max_pool2d(x, 2, 2, ceil_mode=True).numpy()
    np.testing.assert_array_equal(ops.maxpool2x2_ceil(ctx, x.numpy()), ref)


@pytest.mark.parametrize("mnk", [(1000, 4096, 512), (1000, 72, 256), (37, 5, 4096), (300, 10498, 512),
                                 (300, 4096, 25088),       # fc6 at 300 proposals: split-K 8 over three rounds of workgroups
                                 (500, 4096, 12544),       # 128 tiles: split-K 2 fills the chip in one round
                                 (640, 4096, 6272),        # 160 tiles: split-K over several rounds instead of one round on 160 CUs
                                 (1710, 64, 64), (129, 257, 96), (1, 1, 32)])
def test_linear_matches_fp64(ctx, mnk):
    from densecap_amd import ops
    M, N, K = mnk
    rng = np.random.default_rng(M + N + K)
    x = rng.standard_normal((M, K)).astype(np.float32)
    w = (rng.standard_normal((N, K)) / np.sqrt(K)).astype(np.float32)
    b = rng.standard_normal(N).astype(np.float32)
    ref = x.astype(np.float64) @ w.astype(np.float64).T + b
    _close(ops.linear(ctx, x, w, b), ref.astype(np.float32), rel=2e-5)
    _close(ops.linear(ctx, x, w, b, relu=True), np.maximum(ref, 0).astype(np.float32), rel=2e-5)
    _close(ops.linear(ctx, x, w, None), (ref - b).astype(np.float32), rel=2e-5)


def test_fc6_shape_k25088(ctx):
    """recog_base fc6 (DenseCapModel.lua:133) at its real size: M=1000 RoIs, N=4096, K=25088 -- the K-split 128x128
    kernel with 784 K-tiles, 8 M-tiles sharing each 411 MB weight panel -- against an fp64 reference."""
    import torch
    from densecap_amd import ops
    M, N, K = 1000, 4096, 25088
    g = torch.Generator().manual_seed(6)
    x = torch.relu(torch.randn(M, K, generator=g))
    w = torch.randn(N, K, generator=g) * (2.0 / K) ** 0.5
    b = torch.randn(N, generator=g)
    torch.set_num_threads(max(1, min(32, torch.get_num_threads())))
    ref = torch.relu(torch.addmm(b.double(), x.double(), w.double().t())).float().numpy()
    _close(ops.linear(ctx, x.numpy(), w.numpy(), b.numpy(), relu=True), ref, rel=2e-5)


@pytest.mark.parametrize("case", [("conv5_x split-K", 1, 512, 38, 45, 512), ("rpn conv split-K", 1, 512, 38, 45, 256),
                                  ("conv2_2 tail plan", 1, 128, 300, 360, 128), ("conv3_2 tail plan", 1, 256, 150, 180, 256),
                                  ("conv4_2 tail plan", 1, 512, 75, 90, 512)])
def test_conv_splitk_and_tail_plans_single_lane(case):
    """dc_set_lanes(1) (single-image mode) routes layers whose 128x128 tile count is small or not a multiple of 256
    through split-K / the tail plan (another fixed fp32 summation order): each such shape against fp64 torch."""
    import torch
    from densecap_amd import ops
    from densecap_amd._lib import check
    name, N, Cin, H, W, Cout = case
    c = ops.Context(0)
    try:
        check(c.h, c.lib.dc_set_lanes(c.h, 1), "dc_set_lanes")
        g = torch.Generator().manual_seed(Cin + H + Cout)
        x = torch.relu(torch.randn(N, Cin, H, W, generator=g))
        w = torch.randn(Cout, Cin, 3, 3, generator=g) * (2.0 / (9 * Cin)) ** 0.5
        b = torch.randn(Cout, generator=g)
        torch.set_num_threads(max(1, min(32, torch.get_num_threads())))
        ref = torch.relu(torch.nn.functional.conv2d(x.double(), w.double(), b.double(), padding=1)).float().numpy()
        _close(ops.conv3x3(c, x.numpy(), w.numpy(), b.numpy(), relu=True), ref, rel=2e-5)
    finally:
        c.close()


@pytest.mark.parametrize("case", [("conv2_2", 128, 300, 360, 128), ("conv3_1", 128, 150, 180, 256), ("conv3_2", 256, 150, 180, 256),
                                  ("conv4_1", 256, 75, 90, 512), ("conv4_2", 512, 75, 90, 512),
                                  ("webcam conv3_2", 256, 80, 120, 256), ("webcam conv2_2", 128, 160, 240, 128)])
def test_streamk_last_round_all_routes(case):
    """Single-image mode, layers whose 128x128 tile count is not a multiple of the CU count: the three routes of the
    last partial round -- stream-K with in-kernel fix-up (default), the K-split tail plan, whole tiles (dc_debug_set
    "tail_mode") -- each against fp64; stream-K three times over with fresh data in the same buffers (a stale partial tile
    read across XCDs would show as a wrong block) and bit-identical to itself on a repeat."""
    import torch
    from densecap_amd import ops
    from densecap_amd._lib import check
    name, Cin, H, W, Cout = case
    c = ops.Context(0)
    try:
        check(c.h, c.lib.dc_set_lanes(c.h, 1), "dc_set_lanes")
        torch.set_num_threads(max(1, min(32, torch.get_num_threads())))
        g = torch.Generator().manual_seed(Cin + H + Cout)
        w = torch.randn(Cout, Cin, 3, 3, generator=g) * (2.0 / (9 * Cin)) ** 0.5
        b = torch.randn(Cout, generator=g)
        first = None
        for rep in range(3):
            x = torch.relu(torch.randn(1, Cin, H, W, generator=g)) * (1.0 + rep)
            ref = torch.relu(torch.nn.functional.conv2d(x.double(), w.double(), b.double(), padding=1)).float().numpy()
            for mode in ((0, 1, 2) if rep == 0 else (0,)):
                check(c.h, c.lib.dc_debug_set(c.h, b"tail_mode", mode), "dc_debug_set")
                out = ops.conv3x3(c, x.numpy(), w.numpy(), b.numpy(), relu=True)
                _close(out, ref, rel=2e-5)
                if mode == 0 and rep == 0:
                    first = (x, out)
            check(c.h, c.lib.dc_debug_set(c.h, b"tail_mode", 0), "dc_debug_set")
        again = ops.conv3x3(c, first[0].numpy(), w.numpy(), b.numpy(), relu=True)
        np.testing.assert_array_equal(again, first[1])
    finally:
        c.close()


def test_short_row_tiles_skip_dead_blocks(ctx):
    """K-split kernel, row tiles with <= 64 live rows (a 50-proposal batch; the last tile of 300 rows): the two dead 32-row
    blocks are neither fetched nor multiplied.  Against fp64, and ROW-INVARIANT: a row gives the same bits whether its
    tile ran the short variant (M = 300: rows 256..299) or the full one (the same rows inside M = 384)."""
    from densecap_amd import ops
    rng = np.random.default_rng(9)
    N, K = 512, 4096
    w = (rng.standard_normal((N, K)) / np.sqrt(K)).astype(np.float32)
    b = rng.standard_normal(N).astype(np.float32)
    x = rng.standard_normal((384, K)).astype(np.float32)
    full = ops.linear(ctx, x, w, b, relu=True)
    ref = np.maximum(x.astype(np.float64) @ w.astype(np.float64).T + b, 0).astype(np.float32)
    _close(full, ref, rel=2e-5)
    for M in (1, 33, 50, 64, 65, 172, 300):
        out = ops.linear(ctx, x[:M], w, b, relu=True)
        np.testing.assert_array_equal(out, full[:M], err_msg="M=%d" % M)
    # the 128x64 kernel (short K, many tiles): a last row tile with 44 live rows runs its 64x64 variant -- same bits as inside a full tile
    Mb, Kb = 128 * 96 + 128, 512
    wb = (rng.standard_normal((N, Kb)) / np.sqrt(Kb)).astype(np.float32)
    xb = rng.standard_normal((Mb, Kb)).astype(np.float32)
    fullb = ops.linear(ctx, xb, wb, b)
    _close(fullb, (xb.astype(np.float64) @ wb.astype(np.float64).T + b).astype(np.float32), rel=2e-5)
    for M in (128 * 96 + 44, 128 * 96 + 64, 128 * 96 + 65):
        np.testing.assert_array_equal(ops.linear(ctx, xb[:M], wb, b), fullb[:M], err_msg="M=%d" % M)
    # split-K on top (few tiles, long K): fc6-like at 50 rows
    K2 = 25088
    w2 = (rng.standard_normal((256, K2)) / np.sqrt(K2)).astype(np.float32)
    x2 = rng.standard_normal((128, K2)).astype(np.float32)
    full2 = ops.linear(ctx, x2, w2, None)
    _close(full2, (x2.astype(np.float64) @ w2.astype(np.float64).T).astype(np.float32), rel=2e-5)
    _close(ops.linear(ctx, x2[:50], w2, None), full2[:50], rel=2e-5)      # (the split-K factor may differ with M: tolerance)


@pytest.mark.parametrize("case", [(64, 300, 360, 128, False),      # conv2_1 at 720x600: 1688 tiles of 128x64 -> two-stage ring by default
                                  (64, 301, 203, 64, True),        # conv1_2-like with the pool epilogue, odd sizes, ragged last round
                                  (64, 75, 90, 64, False)])        # few tiles: three stages by default
def test_ring_depth_bit_identical(case):
    """The 128x64-tile kernel with a two- or a three-stage LDS ring (dc_debug_set "v2_stages") adds the same products in
    the same order: bit-identical outputs, whichever the tile count selects by default."""
    from densecap_amd import ops
    from densecap_amd._lib import check
    Cin, H, W, Cout, pool = case
    c = ops.Context(0)
    try:
        rng = np.random.default_rng(H + Cout)
        x = np.maximum(rng.standard_normal((1, Cin, H, W)), 0).astype(np.float32)
        w = (rng.standard_normal((Cout, Cin, 3, 3)) * (2.0 / (9 * Cin)) ** 0.5).astype(np.float32)
        b = rng.standard_normal(Cout).astype(np.float32)
        outs = []
        for st in (0, 2, 3):
            check(c.h, c.lib.dc_debug_set(c.h, b"v2_stages", st), "dc_debug_set")
            outs.append(ops.conv3x3_relu_pool(c, x[0], w, b) if pool else ops.conv3x3(c, x, w, b, relu=True))
        np.testing.assert_array_equal(outs[1], outs[0]); np.testing.assert_array_equal(outs[2], outs[0])
        import torch
        ref = torch.relu(torch.nn.functional.conv2d(torch.from_numpy(x).double(), torch.from_numpy(w).double(),
                                                    torch.from_numpy(b).double(), padding=1))
        if pool:
            ref = torch.nn.functional.max_pool2d(ref, 2, 2, ceil_mode=True)
        _close(outs[0].reshape(ref.shape[1:]) if pool else outs[0], ref.float().numpy()[0] if pool else ref.float().numpy(), rel=2e-5)
    finally:
        c.close()


def test_streamk_dense_gemm_single_lane():
    """The same machinery on a dense contraction whose tile count leaves a partial round: (6750, 512, 4608) = 212 tiles."""
    from densecap_amd import ops
    from densecap_amd._lib import check
    c = ops.Context(0)
    try:
        check(c.h, c.lib.dc_set_lanes(c.h, 1), "dc_set_lanes")
        M, N, K = 6750, 512, 4608
        rng = np.random.default_rng(4)
        x = rng.standard_normal((M, K)).astype(np.float32)
        w = (rng.standard_normal((N, K)) / np.sqrt(K)).astype(np.float32)
        b = rng.standard_normal(N).astype(np.float32)
        ref = (x.astype(np.float64) @ w.astype(np.float64).T + b).astype(np.float32)
        for mode in (0, 1, 2):
            check(c.h, c.lib.dc_debug_set(c.h, b"tail_mode", mode), "dc_debug_set")
            _close(ops.linear(c, x, w, b), ref, rel=2e-5)
    finally:
        c.close()


@pytest.mark.parametrize("case", [(64, 150, 180, 64, 3), (64, 37, 53, 64, 3), (128, 75, 90, 128, 3), (32, 9, 11, 32, 3),
                                  (64, 301, 203, 64, 3),            # conv1_2-like, odd sizes: ceil-mode windows at both borders
                                  (128, 300, 360, 128, 1),          # conv2_2 at 720x600: tail plan (single-lane mode)
                                  (256, 150, 180, 256, 1),          # conv3_3: tail plan
                                  (512, 75, 90, 512, 1),            # conv4_3: 212 tiles, odd height
                                  (512, 90, 136, 512, 3),           # conv4_3 at 1080x720: 384 tiles of 1.5 rounds -> costed onto 128x64 tiles
                                  (512, 19, 23, 512, 1)])           # few tiles: split-K + pooled reduce
def test_conv_relu_pool_fused_equals_conv_then_pool(case):
    """The pool taken in the conv epilogue (pool-window-ordered implicit GEMM rows) must be bit-identical to the conv
    followed by the stand-alone ceil-mode pool (same K order per pixel; max and ReLU commute exactly) whenever every row
    takes the same kernel route, i.e. in the multi-lane mode; in single-image mode it agrees to fp32 rounding."""
    import torch
    from densecap_amd import ops
    from densecap_amd._lib import check
    Cin, H, W, Cout, lanes = case
    c = ops.Context(0)
    try:
        check(c.h, c.lib.dc_set_lanes(c.h, lanes), "dc_set_lanes")
        g = torch.Generator().manual_seed(Cin + H + W)
        x = torch.randn(Cin, H, W, generator=g).numpy()
        w = (torch.randn(Cout, Cin, 3, 3, generator=g) * (2.0 / (9 * Cin)) ** 0.5).numpy()
        b = torch.randn(Cout, generator=g).numpy()
        fused = ops.conv3x3_relu_pool(c, x, w, b)
        full = ops.conv3x3(c, x[None], w, b, relu=True)
        unfused = ops.maxpool2x2_ceil(c, full)[0]
        assert fused.shape == (Cout, (H + 1) // 2, (W + 1) // 2)
        if lanes != 1:
            np.testing.assert_array_equal(fused, unfused)
        else:
            # single-image mode K-splits the LAST partial round of tiles (another fixed fp32 summation order for those
            # rows); the fused conv walks pool windows, the plain one raster rows, so different pixels fall in that round
            _close(fused, unfused, rel=1e-5)
        # and the full-resolution conv itself against fp64 (the pooled values inherit its accuracy)
        ref = torch.relu(torch.nn.functional.conv2d(torch.from_numpy(x)[None].double(), torch.from_numpy(w).double(),
                                                    torch.from_numpy(b).double(), padding=1))
        refp = torch.nn.functional.max_pool2d(ref, 2, 2, ceil_mode=True)[0].float().numpy()
        _close(fused, refp, rel=2e-5)
    finally:
        c.close()


@pytest.mark.parametrize("case", [(3, 40, 32), (2, 9, 33), (1, 130, 47), (2, 64, 63), (1, 200, 17), (4, 3, 97)])
def test_conv_row_pass_walk_wraps_image_rows(case):
    """Interior tiles derive their other row passes from ONE division pair per lane (a step of 32 pixels that may wrap the
    image row -- every step at width 32 -- and the image); widths below 32 keep the division per pass."""
    import torch
    from densecap_amd import ops
    N, H, W = case
    c = ops.Context(0)
    try:
        g = torch.Generator().manual_seed(N * 1000 + H * 10 + W)
        x = torch.randn(N, 64, H, W, generator=g)
        w = torch.randn(64, 64, 3, 3, generator=g) * (2.0 / (9 * 64)) ** 0.5
        b = torch.randn(64, generator=g)
        ref = torch.relu(torch.nn.functional.conv2d(x.double(), w.double(), b.double(), padding=1)).float().numpy()
        out = ops.conv3x3(c, x.numpy(), w.numpy(), b.numpy(), relu=True)
        _close(out, ref)
        # the images of a launch are independent: each one alone gives the same bits
        for i in range(N):
            one = ops.conv3x3(c, x[i:i + 1].numpy(), w.numpy(), b.numpy(), relu=True)
            np.testing.assert_array_equal(one[0], out[i])
    finally:
        c.close()


@pytest.mark.parametrize("hw", [(40, 16), (33, 17), (21, 15), (64, 12), (300, 3), (300, 2), (301, 4), (19, 64)])
def test_pooled_conv_window_walk_narrow_images(hw):
    """The pooled epilogue walks a lane's windows in steps of two from one division pair (a step wraps the window row at
    most once: every step at two windows per row); the prologue steps eight windows per row pass from widths of 16 on."""
    import torch
    from densecap_amd import ops
    from densecap_amd._lib import check
    H, W = hw
    c = ops.Context(0)
    try:
        check(c.h, c.lib.dc_set_lanes(c.h, 3), "dc_set_lanes")
        g = torch.Generator().manual_seed(H * 1000 + W)
        x = torch.randn(64, H, W, generator=g).numpy()
        w = (torch.randn(64, 64, 3, 3, generator=g) * (2.0 / (9 * 64)) ** 0.5).numpy()
        b = torch.randn(64, generator=g).numpy()
        fused = ops.conv3x3_relu_pool(c, x, w, b)
        unfused = ops.maxpool2x2_ceil(c, ops.conv3x3(c, x[None], w, b, relu=True))[0]
        np.testing.assert_array_equal(fused, unfused)
        ref = torch.relu(torch.nn.functional.conv2d(torch.from_numpy(x)[None].double(), torch.from_numpy(w).double(),
                                                    torch.from_numpy(b).double(), padding=1))
        _close(fused, torch.nn.functional.max_pool2d(ref, 2, 2, ceil_mode=True)[0].float().numpy(), rel=2e-5)
    finally:
        c.close()


def test_lm_encoder_splitk_single_lane():
    """image_encoder Linear(4096,512) at M=1000 (32 tiles, K=4096 -> split-K in single-image mode)."""
    from densecap_amd import ops
    from densecap_amd._lib import check
    c = ops.Context(0)
    try:
        check(c.h, c.lib.dc_set_lanes(c.h, 1), "dc_set_lanes")
        rng = np.random.default_rng(11)
        x = np.maximum(rng.standard_normal((1000, 4096)), 0).astype(np.float32)
        w = (rng.standard_normal((512, 4096)) / 64).astype(np.float32)
        b = rng.standard_normal(512).astype(np.float32)
        ref = np.maximum(x.astype(np.float64) @ w.astype(np.float64).T + b, 0).astype(np.float32)
        _close(ops.linear(c, x, w, b, relu=True), ref, rel=2e-5)
    finally:
        c.close()


def test_linear_transpose_detecting(ctx):
    # A = I with an asymmetric W catches a swapped C/D register map
    from densecap_amd import ops
    K = 64
    x = np.eye(K, dtype=np.float32)
    w = np.arange(96 * K, dtype=np.float32).reshape(96, K)
    np.testing.assert_array_equal(ops.linear(ctx, x, w, None), w.T)


def test_layout_roundtrip(ctx):
    from densecap_amd import ops
    x = np.random.default_rng(0).standard_normal((19, 13, 37)).astype(np.float32)
    h = ops.chw_to_hwc(ctx, x)
    np.testing.assert_array_equal(h, x.transpose(1, 2, 0))
    np.testing.assert_array_equal(ops.hwc_to_chw(ctx, h), x)


# ---- box algebra: the reference's known-answer tests, on the GPU ---------------------------------
def test_apply_box_transform_golden(ctx, golden):
    from densecap_amd import ops
    g = golden["apply_box_transform"]
    out = ops.apply_box_transform(ctx, np.array(g["boxes"]), np.array(g["trans"]))
    np.testing.assert_allclose(out, np.array(g["expected"]), atol=1e-4, rtol=1e-6)


def test_make_boxes_golden(ctx, golden):
    from densecap_amd import ops
    for case in golden["make_boxes"]:
        N, k, H, W = case["N"], case["k"], case["H"], case["W"]
        anchors = np.array(case["anchors"], np.float32)
        head = np.zeros((N, 4 * k, H, W), np.float32)
        for (n, a, y, x), v in case["inputs"]:
            head[n, 4 * a:4 * a + 4, y, x] = v
        anc = ops.make_anchors(ctx, H, W, case["x0"], case["y0"], case["sx"], case["sy"], anchors)
        for n in range(N):
            trans = head[n].reshape(k, 4, H, W).transpose(0, 2, 3, 1).reshape(-1, 4)
            boxes = ops.apply_box_transform(ctx, anc, trans)
            for (nn, y, x, a), v in case["expected"]:
                if nn == n:
                    np.testing.assert_allclose(boxes[a * H * W + y * W + x], v, atol=1e-4)


def test_box_conversions_and_clip_exact(ctx):
    from densecap_amd import ops
    from oracle import densecap_oracle as O
    rng = np.random.default_rng(5)
    b = np.concatenate([rng.uniform(-100, 900, (5000, 2)), rng.uniform(0.1, 800, (5000, 2))], 1).astype(np.float32)
    b[:50, 2:] = rng.uniform(0, 1.2, (50, 2))  # degenerate boxes exercise valid=false
    np.testing.assert_array_equal(ops.xcycwh_to_x1y1x2y2(ctx, b), O.xcycwh_to_x1y1x2y2(b))
    c, v = ops.clip_boxes(ctx, b, dict(x_min=1, y_min=1, x_max=720, y_max=600))
    co, vo = O.clip_boxes_xcycwh(b, 1, 1, 720, 600)
    np.testing.assert_array_equal(c, co)
    np.testing.assert_array_equal(v, vo)
    assert 0 < (~vo).sum() < 50


def test_device_corners_and_host_xywh_close_the_reference_round_trip(ctx):
    """test/box_conversion_test.lua:12-23 with the product's two conversions in it: the device's xcycwh -> corners
    (box_utils.lua:288-291, what both NMS runs read) and the host's xcycwh -> xywh (run_model.lua:78).  Corners from the
    device, re-expressed as xywh by the oracle's x1y1x2y2_to_xywh, are the host function's output bit for bit -- that IS
    box_utils.xcycwh_to_xywh (box_utils.lua:441-445) -- and the reference's round trip closes on them."""
    from densecap_amd import ops
    from densecap_amd.run_model import xcycwh_to_xywh
    from oracle import densecap_oracle as O
    rng = np.random.default_rng(12)
    b = np.concatenate([rng.uniform(-100, 900, (3000, 2)), rng.uniform(0, 800, (3000, 2))], 1).astype(np.float32)
    corners = ops.xcycwh_to_x1y1x2y2(ctx, b)
    xywh = xcycwh_to_xywh(b)
    np.testing.assert_array_equal(O.x1y1x2y2_to_xywh(corners), xywh)
    np.testing.assert_array_equal(xywh, O.xcycwh_to_xywh(b))
    big = np.abs(b).max(axis=1) + 1.0
    again = O.xywh_to_x1y1x2y2(xywh)
    assert (np.abs(again - corners).max(axis=1) <= 1e-6 * big).all()
    assert (np.abs(O.x1y1x2y2_to_xywh(again) - xywh).max(axis=1) <= 1e-6 * big).all()


def test_box_iou_module(ctx):
    from densecap_amd import ops
    from oracle import densecap_oracle as O
    rng = np.random.default_rng(6)
    b1 = np.concatenate([rng.uniform(0, 300, (70, 2)), rng.uniform(5, 200, (70, 2))], 1).astype(np.float32)
    b2 = np.concatenate([rng.uniform(0, 300, (33, 2)), rng.uniform(5, 200, (33, 2))], 1).astype(np.float32)
    np.testing.assert_array_equal(ops.box_iou(ctx, b1, b2, 0), O.box_iou_module(b1, b2))


@pytest.mark.parametrize("conv", [(0, "boxiou_module"), (1, "nms_plus1"), (2, "legacy_half_w")])
def test_box_iou_conventions_exact(ctx, conv):
    """dc_op_box_iou under the three conventions of SURVEY.md 8 a21, bit-exact against the oracle; the +1 convention
    must also reproduce the IoU that box_utils.nms computes inline (checked through a pick/suppress decision)."""
    from densecap_amd import ops
    from oracle import densecap_oracle as O
    code, name = conv
    rng = np.random.default_rng(code)
    b1 = np.concatenate([rng.uniform(0, 300, (70, 2)), rng.uniform(1, 120, (70, 2))], 1).astype(np.float32)
    b2 = np.concatenate([rng.uniform(0, 300, (133, 2)), rng.uniform(1, 120, (133, 2))], 1).astype(np.float32)
    b2[:20] = b1[:20]                                            # identical boxes
    got = ops.box_iou(ctx, b1, b2, code)
    np.testing.assert_array_equal(got, O.box_iou(b1, b2, name))
    assert got.min() >= 0 and got.max() <= 1 + 2e-6          # legacy_half_w can round a self-IoU to 1 + 1 ulp
    if name == "nms_plus1":
        assert (np.diag(got[:20, :20]) == 1).all()             # intersection and both areas are the same fp32 expression
    elif name == "legacy_half_w":
        np.testing.assert_allclose(np.diag(got[:20, :20]), 1.0, rtol=1e-5)   # (xc + w/2) - (xc - w/2) rounds, w*h does not
    else:
        # the live module mixes (w-1) extents in the intersection with w*h areas: IoU(b, b) = (w-1)(h-1) / (2wh - (w-1)(h-1)) < 1
        assert (np.diag(got[:20, :20]) < 1).all()
    if name == "nms_plus1":
        # a pair is suppressed by box_utils.nms at threshold t  <=>  its +1 IoU > t
        i, j = np.unravel_index(np.argmax(np.where(got < 0.999, got, 0)), got.shape)
        pair = np.stack([b1[i], b2[j]])
        b5 = np.concatenate([O.xcycwh_to_x1y1x2y2(pair), np.array([[2.0], [1.0]], np.float32)], 1)
        t = float(got[i, j])
        assert len(O.nms(b5, np.nextafter(np.float32(t), np.float32(0)), None)) == 1     # thr just below: suppressed
        assert len(O.nms(b5, t, None)) == 2                                               # iou <= thr: kept


def test_box_iou_legacy_half_w_golden(ctx, golden):
    """test/BoxIoU_test.lua:13-94 vectors (written for the module's original xc -/+ w/2 converter) on the GPU."""
    from densecap_amd import ops
    for case in golden["box_iou_legacy_half_w"]:
        got = ops.box_iou(ctx, np.array(case["boxes1"], np.float32), np.array(case["boxes2"], np.float32), 2)
        np.testing.assert_allclose(got, np.array(case["expected"]), atol=1e-6, rtol=1e-6)


def _bits_equal(a, b):
    """float32 arrays equal bit for bit (NaN == NaN of the same payload class: both NaN counts as equal)."""
    a = np.ascontiguousarray(a, np.float32); b = np.ascontiguousarray(b, np.float32)
    return ((a.view(np.uint32) == b.view(np.uint32)) | (np.isnan(a) & np.isnan(b)))


def test_rpn_decode_matches_oracle(ctx):
    """Round 5: exp on a FloatTensor is the C double exp cast to float on BOTH sides (TH's form, docs/SEMANTICS.md; the
    device used its own expf before, a few ulp away from glibc's) -- every other operation of the decode is a single
    correctly rounded fp32 operation in the reference's order, so the whole stage is bit-exact."""
    from densecap_amd import ops
    from oracle import densecap_oracle as O
    rng = np.random.default_rng(7)
    k, h, w = 12, 38, 45
    box_head = (rng.standard_normal((4 * k, h, w)) * 0.3).astype(np.float32)
    score_head = (rng.standard_normal((2 * k, h, w)) * 1.5).astype(np.float32)
    d = ops.rpn_decode(ctx, box_head, score_head, 600, 720, O.DEFAULT_ANCHORS, O.VGG16_FIELD_CENTERS)
    o = O.rpn_decode(box_head, score_head, 600, 720)
    np.testing.assert_array_equal(d["valid"], o["valid"])
    rows = o["rows"]
    np.testing.assert_array_equal(d["anchors"][rows], o["anchors"])
    np.testing.assert_array_equal(d["trans"][rows], o["trans"])
    # (float)exp((double)x) is the same float in any libm unless the double lands within 2^-29 of a rounding boundary:
    # allow one such element in a million, and then only by one ulp
    for name in ("boxes", "x1y1x2y2", "p"):
        eq = _bits_equal(d[name][rows], o[name])
        assert eq.mean() >= 1 - 1e-6, (name, float(eq.mean()))
        np.testing.assert_allclose(d[name][rows], o[name], rtol=2.4e-7, atol=0)


def test_rpn_decode_overflowing_logits_replicate_the_reference(ctx):
    """`(e1+e2)^-1 * e1` (LocalizationLayer.lua:304-308) is NOT a stable softmax -- |s| > ~88.7 makes
    exp overflow and p becomes 0, inf*0 = NaN or 1/inf*finite = 0.  Replicated, not fixed: with score logits of +-60..+-120
    the device's p equals the oracle's bit for bit, NaNs included, and the NMS that follows ranks NaN first on both sides."""
    from densecap_amd import ops
    from oracle import densecap_oracle as O
    rng = np.random.default_rng(11)
    k, h, w = 12, 20, 24
    box_head = (rng.standard_normal((4 * k, h, w)) * 0.3).astype(np.float32)
    mag = rng.uniform(60, 120, (2 * k, h, w)) * rng.choice([-1.0, 1.0], (2 * k, h, w))
    score_head = np.where(rng.uniform(size=mag.shape) < 0.5, mag, rng.standard_normal(mag.shape) * 1.5).astype(np.float32)
    d = ops.rpn_decode(ctx, box_head, score_head, 320, 384, O.DEFAULT_ANCHORS, O.VGG16_FIELD_CENTERS)
    o = O.rpn_decode(box_head, score_head, 320, 384)
    np.testing.assert_array_equal(d["valid"], o["valid"])
    rows = o["rows"]
    p = d["p"][rows]
    assert np.isnan(p).sum() > 50 and (p == 0).sum() > 50 and np.isfinite(p).sum() > 50       # the case is what it claims to be
    assert np.isnan(o["p"]).sum() == np.isnan(p).sum()
    assert _bits_equal(p, o["p"]).mean() >= 1 - 1e-5
    b5d = np.concatenate([d["x1y1x2y2"][rows], p[:, None]], 1)
    b5o = np.concatenate([o["x1y1x2y2"], o["p"][:, None]], 1)
    for maxb in (300, None):
        got = ops.nms(ctx, b5d, 0.7, maxb)
        assert got.tolist() == O.nms(b5d, 0.7, maxb).tolist()                # teacher-forced on the device's own rows
        assert np.isnan(p[got[0]])                                            # NaN ranks above every number (SEMANTICS.md)
        if _bits_equal(b5d, b5o).all():
            assert got.tolist() == O.nms(b5o, 0.7, maxb).tolist()


# ---- NMS ----------------------------------------------------------------------------------------
def test_nms_golden(ctx, golden):
    from densecap_amd import ops
    for case in golden["nms"]:
        pick = ops.nms(ctx, np.array(case["boxes"], np.float32), case["thresh"])
        assert pick.tolist() == case["expected"], case["cite"]


def _random_boxes5(rng, n, ties=False):
    xy = rng.uniform(0, 700, (n, 2)); wh = rng.uniform(8, 400, (n, 2))
    s = rng.uniform(0, 1, (n, 1))
    if ties:
        s = np.round(s, 3)
    return np.concatenate([xy, xy + wh, s], 1).astype(np.float32)


@pytest.mark.parametrize("n,thr,maxb,ties", [(1, 0.7, 10, False), (63, 0.5, None, False), (64, 0.3, None, True),
                                              (65, 0.7, 7, False), (1000, 0.3, None, True),
                                              (5000, 0.7, 300, True), (20520, 0.7, 1000, False),
                                              (20520, 0.7, 1000, True), (36720, 0.7, 2000, False)])
def test_nms_exact_vs_oracle(ctx, n, thr, maxb, ties):
    from densecap_amd import ops
    from oracle import densecap_oracle as O
    b = _random_boxes5(np.random.default_rng(n + int(thr * 10)), n, ties)
    assert ops.nms(ctx, b, thr, maxb).tolist() == O.nms(b, thr, maxb).tolist()


@pytest.mark.parametrize("n,maxb", [(90000, 1000), (150000, None)])
def test_nms_beyond_65536_boxes(ctx, n, maxb):
    """No box limit in the reference (box_utils.lua:154-256): 90,000 = a 1600x1200 image's anchors; 150,000 near-duplicate
    clusters, uncapped, walk the 4096-row window and five 32768-row windows with suppression carried across all of them."""
    from densecap_amd import ops
    from oracle import densecap_oracle as O
    rng = np.random.default_rng(n)
    if maxb is None:
        ncl, per = n // 50, 50
        cxy = rng.uniform(0, 6000, (ncl, 1, 2)); wh = rng.uniform(30, 60, (ncl, 1, 2))
        xy = cxy + rng.uniform(-2, 2, (ncl, per, 2))
        b = np.concatenate([xy, xy + wh + rng.uniform(-2, 2, (ncl, per, 2))], 2).reshape(-1, 4)
        b5 = np.concatenate([b, np.round(rng.uniform(0, 1, (n, 1)), 4)], 1).astype(np.float32)
    else:
        b5 = _random_boxes5(rng, n, True)
    got, ref = ops.nms(ctx, b5, 0.6, maxb), O.nms(b5, 0.6, maxb)
    assert got.tolist() == ref.tolist() and len(ref) >= (1000 if maxb else 2000)


def _clustered_boxes5(rng, n, per, ties=True):
    """near-duplicate clusters whose members carry independent scores: a pick suppresses boxes anywhere later in the sorted list"""
    ncl = (n + per - 1) // per
    cxy = rng.uniform(0, 3000, (ncl, 1, 2)); wh = rng.uniform(30, 80, (ncl, 1, 2))
    xy = cxy + rng.uniform(-6, 6, (ncl, per, 2))
    b = np.concatenate([xy, xy + wh + rng.uniform(-6, 6, (ncl, per, 2))], 2).reshape(-1, 4)[:n]
    sc = rng.uniform(0, 1, (n, 1))
    return np.concatenate([b, np.round(sc, 3) if ties else sc], 1).astype(np.float32)


@pytest.mark.parametrize("n,per,thr,maxb", [(4096, 40, 0.5, None), (4096, 8, 0.3, None), (4000, 100, 0.7, 300), (3000, 3, 0.6, None),
                                            (20520, 30, 0.7, 1000), (20520, 200, 0.5, 1000), (1000, 5, 0.3, None), (130, 10, 0.5, None),
                                            (4097, 64, 0.4, None), (36720, 20, 0.7, 2000), (4096, 1024, 0.3, None), (3024, 3024, 0.2, 100)])
def test_nms_band_scan_equals_the_chunk_scan(ctx, n, per, thr, maxb):
    """Round 6: windows of <= 4096 rows are scanned by nms_scan_band_kernel (near words of the mask in LDS, far words fetched two
    chunks behind the resolving wave); dc_debug_set("nms_band", 0) puts every window back on nms_scan_kernel.  Clustered
    boxes make picks suppress candidates many chunks later -- the far path -- and both kernels must give the oracle's list."""
    from densecap_amd import ops
    from densecap_amd._lib import check
    from oracle import densecap_oracle as O
    b = _clustered_boxes5(np.random.default_rng(n * 7 + per), n, per)
    ref = O.nms(b, thr, maxb).tolist()
    try:
        check(ctx.h, ctx.lib.dc_debug_set(ctx.h, b"nms_band", 0), "dc_debug_set")
        chunk = ops.nms(ctx, b, thr, maxb).tolist()
    finally:
        check(ctx.h, ctx.lib.dc_debug_set(ctx.h, b"nms_band", 1), "dc_debug_set")
    band = ops.nms(ctx, b, thr, maxb).tolist()
    assert chunk == ref and band == ref
    assert len(ref) < n                                      # (something was suppressed)


def test_nms_valid_mask_equals_compaction(ctx):
    # LocalizationLayer.lua:285-298 compacts by `valid` before NMS; masking is equivalent
    from densecap_amd import ops
    from oracle import densecap_oracle as O
    rng = np.random.default_rng(11)
    b = _random_boxes5(rng, 3000, True)
    valid = rng.uniform(size=3000) > 0.2
    rows = np.nonzero(valid)[0]
    assert ops.nms(ctx, b, 0.6, 200, valid=valid).tolist() == rows[O.nms(b[rows], 0.6, 200)].tolist()


def test_nms_empty(ctx):
    from densecap_amd import ops
    assert ops.nms(ctx, np.zeros((0, 5), np.float32), 0.5).size == 0


# ---- bilinear RoI pooling ----------------------------------------------------------------------
@pytest.mark.parametrize("cfg", [dict(B=10, C=128, h=32, w=33, HH=7, WW=8, H=512, W=528),   # BatchBilinearSamplerBHWD_test.lua:15-46 shapes
                                 dict(B=1000, C=512, h=38, w=45, HH=7, WW=7, H=600, W=720),
                                 dict(B=128, C=512, h=32, w=32, HH=7, WW=7, H=512, W=512)])  # BilinearRoiPooling_test.lua:58-97
def test_bilinear_roi_pool_vs_oracle(ctx, cfg):
    from densecap_amd import ops
    from oracle import densecap_oracle as O
    rng = np.random.default_rng(cfg["B"])
    feat = rng.standard_normal((cfg["C"], cfg["h"], cfg["w"])).astype(np.float32)
    B = cfg["B"]
    boxes = np.concatenate([rng.uniform(-50, cfg["W"] + 50, (B, 1)), rng.uniform(-50, cfg["H"] + 50, (B, 1)),
                            rng.uniform(2, cfg["W"], (B, 1)), rng.uniform(2, cfg["H"], (B, 1))], 1).astype(np.float32)
    ref = O.bilinear_roi_pool(feat, boxes, cfg["H"], cfg["W"], cfg["HH"], cfg["WW"])
    out = ops.bilinear_roi_pool(ctx, feat, boxes, cfg["H"], cfg["W"], cfg["HH"], cfg["WW"], out_layout=0)
    np.testing.assert_allclose(out, ref, atol=1e-6, rtol=0)   # the reference's own fast-vs-naive tolerance
    out1 = ops.bilinear_roi_pool(ctx, feat, boxes, cfg["H"], cfg["W"], cfg["HH"], cfg["WW"], out_layout=1)
    np.testing.assert_array_equal(out1.transpose(0, 3, 1, 2), out)
    assert (np.abs(ref).sum(axis=(1, 2, 3)) > 0).mean() > 0.9


def test_bilinear_roi_pool_identity_property(ctx):
    # whole-image box sampled at the map's own resolution returns the map (BoxToAffine_visual_test.ipynb)
    from densecap_amd import ops
    feat = np.random.default_rng(2).standard_normal((16, 9, 11)).astype(np.float32)
    H, W = 144, 176
    box = np.array([[(W + 1) / 2, (H + 1) / 2, W, H]], np.float32)
    out = ops.bilinear_roi_pool(ctx, feat, box, H, W, 9, 11)
    np.testing.assert_allclose(out[0], feat, atol=1e-5)


def test_nms_nan_and_inf_scores(ctx):
    # docs/SEMANTICS.md: NaN scores sort last (never beat a number); +-inf order like numbers
    from densecap_amd import ops
    from oracle import densecap_oracle as O
    rng = np.random.default_rng(5)
    b = _random_boxes5(rng, 2000, True)
    b[rng.choice(2000, 100, replace=False), 4] = np.nan
    b[rng.choice(2000, 20, replace=False), 4] = np.inf
    b[rng.choice(2000, 20, replace=False), 4] = -np.inf
    for maxb in (None, 150):
        assert ops.nms(ctx, b, 0.5, maxb).tolist() == O.nms(b, 0.5, maxb).tolist()


@pytest.mark.parametrize("maxb", [1000, 2500, None])
def test_nms_clustered_boxes_cross_every_window(ctx, maxb):
    """Near-duplicate clusters: the pick budget is met only deep into the sorted list, so the suppression state has to
    be carried across the 4096-row window, the intermediate 8192-row window of large budgets, and the 32768-row ones."""
    from densecap_amd import ops
    from oracle import densecap_oracle as O
    rng = np.random.default_rng(17)
    ncl, per = 3000, 10
    cxy = rng.uniform(0, 4000, (ncl, 1, 2)); wh = rng.uniform(30, 60, (ncl, 1, 2))
    jit = rng.uniform(-2, 2, (ncl, per, 2))
    xy = cxy + jit
    b = np.concatenate([xy, xy + wh + rng.uniform(-2, 2, (ncl, per, 2))], 2).reshape(-1, 4)
    s = rng.uniform(0, 1, (ncl * per, 1))
    b5 = np.concatenate([b, s], 1).astype(np.float32)
    got = ops.nms(ctx, b5, 0.5, maxb)
    ref = O.nms(b5, 0.5, maxb)
    assert got.tolist() == ref.tolist()
    assert len(ref) > 900
