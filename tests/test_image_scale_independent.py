"""An independent second opinion on `image.scale` (run_model.lua:68; torch/image is not vendored in the reference).

`densecap_amd.run_model.image_scale` and `oracle.image_scale` restate torch/image's `scaleLinear_rowcol` loop by loop --
one author's reading, twice.  This file states WHAT that routine computes, not how, and builds it differently:
an explicit (dst_len x src_len) weight matrix in exact rational arithmetic (fractions.Fraction),

  shrinking  (dst < src):  W[d][s] = | [d*r, (d+1)*r)  intersected with  [s, s+1) | / r,   r = src/dst
                           -- the box-filter area average of the piecewise-constant source;
  enlarging  (dst > src):  output d samples the piecewise-LINEAR source at x = d*(src-1)/(dst-1);
  equal:                   identity,

applied to rows then columns.  The restatement walks the same positions in fp32 (scale, running fractions, accumulate then
divide), so the two agree to fp32 round-off (a few 1e-6 of the value range), never exactly -- which is the point: a wrong
weight, an off-by-one window, a swapped axis or a wrong output size shows up at 1e-2, not 1e-6.
"""
from fractions import Fraction

import numpy as np
import pytest


def weight_matrix(src_len, dst_len):
    """Exact resampling weights as a float64 matrix (rows sum to 1)."""
    W = np.zeros((dst_len, src_len))
    if dst_len == src_len:
        return np.eye(src_len)
    if dst_len > src_len:
        if src_len == 1:
            W[:, 0] = 1.0
            return W
        for d in range(dst_len):
            x = Fraction(d * (src_len - 1), dst_len - 1)
            i = int(x)                                   # floor (x >= 0)
            f = x - i
            if i >= src_len - 1:
                W[d, src_len - 1] = 1.0
            else:
                W[d, i] = float(1 - f)
                W[d, i + 1] = float(f)
        return W
    r = Fraction(src_len, dst_len)
    for d in range(dst_len):
        lo, hi = d * r, (d + 1) * r
        for s in range(int(lo), min(src_len, int(hi) + 1)):
            ov = min(hi, Fraction(s + 1)) - max(lo, Fraction(s))
            if ov > 0:
                W[d, s] = float(ov / r)
    return W


def scale_reference(img, size):
    """image.scale(img, size) with a number: the longer side becomes `size`, the other floor(side*size/longer) -- in
    integers here, where the product divides through Lua doubles."""
    C, ih, iw = img.shape
    imax = max(ih, iw)
    oh, ow = (ih * size) // imax, (iw * size) // imax
    Wx, Wy = weight_matrix(iw, ow), weight_matrix(ih, oh)
    rows = np.einsum("chw,xw->chx", img.astype(np.float64), Wx)          # rows first (width) ...
    return np.einsum("chx,yh->cyx", rows, Wy)                            # ... then columns (height)


def test_weight_matrix_is_a_partition_of_unity_and_matches_hand_values():
    for s, d in [(7, 3), (10, 4), (720, 480), (5, 9), (1, 4), (6, 6), (1000, 720), (3, 2)]:
        W = weight_matrix(s, d)
        np.testing.assert_allclose(W.sum(axis=1), 1.0, atol=1e-12)
        assert (W >= 0).all()
    # 4 -> 2: plain pair averages; 3 -> 2: windows [0,1.5) and [1.5,3)
    np.testing.assert_allclose(weight_matrix(4, 2), [[.5, .5, 0, 0], [0, 0, .5, .5]])
    np.testing.assert_allclose(weight_matrix(3, 2), [[2 / 3, 1 / 3, 0], [0, 1 / 3, 2 / 3]])
    # 2 -> 3: samples at 0, 0.5, 1
    np.testing.assert_allclose(weight_matrix(2, 3), [[1, 0], [.5, .5], [0, 1]])


@pytest.mark.parametrize("seed", range(6))
def test_product_image_scale_against_the_exact_resampler(seed):
    from densecap_amd.run_model import image_scale
    rng = np.random.default_rng(seed)
    for _ in range(6):
        ih, iw = int(rng.integers(1, 90)), int(rng.integers(1, 90))
        size = int(rng.integers(1, 120))
        imax = max(ih, iw)
        if (ih * size) // imax < 1 or (iw * size) // imax < 1:
            with pytest.raises(ValueError):
                image_scale(np.zeros((3, ih, iw), np.float32), size)
            continue
        img = rng.uniform(0, 1, (3, ih, iw)).astype(np.float32)
        got = image_scale(img, size)
        want = scale_reference(img, size)
        assert got.shape == want.shape, (ih, iw, size, got.shape, want.shape)       # incl. the floor of the short side
        assert got.dtype == np.float64                       # image.scale on the DoubleTensor image.load returns; :float() comes after
        err = float(np.abs(got - want).max())
        assert err < 2e-5, (ih, iw, size, err)


def test_the_two_restatements_and_the_exact_resampler_agree_on_real_photo_sizes():
    """The sizes run_model meets: 720x480 (imgs/elephant.jpg) kept, a 1000x667 photo shrunk to 720 (short side 480.24 ->
    480), a 500x375 thumbnail enlarged."""
    from densecap_amd.run_model import image_scale
    from oracle import densecap_oracle as O
    rng = np.random.default_rng(42)
    for (ih, iw, size) in [(480, 720, 720), (667, 1000, 720), (375, 500, 720), (33, 20, 47)]:
        img = rng.uniform(0, 1, (1, ih, iw)).astype(np.float32)
        got = image_scale(img, size)
        want = scale_reference(img, size)
        assert got.shape == want.shape == (1, (ih * size) // max(ih, iw), (iw * size) // max(ih, iw))
        # the restatement's fp32 positions drift from the exact ones by ~6e-8 x the coordinate: 1e-5 of a value at x ~ 1000
        assert float(np.abs(got - want).max()) < 2e-5 + 5e-8 * max(ih, iw, size)
        if ih * iw < 5000:                       # the scalar oracle loop is slow: small case only
            np.testing.assert_array_equal(got, O.image_scale(img, size))


def test_resampler_catches_the_slips_it_is_there_for():
    """Sanity of the checker itself: a transposed application, an off-by-one window and nearest-neighbour all sit far
    outside the tolerance used above."""
    rng = np.random.default_rng(7)
    img = rng.uniform(0, 1, (1, 40, 64)).astype(np.float32)
    want = scale_reference(img, 32)
    nearest = img[:, ::2, ::2][:, :want.shape[1], :want.shape[2]]
    assert np.abs(nearest - want).max() > 1e-2
    Wx = weight_matrix(64, 32)
    shifted = np.einsum("chw,xw->chx", img.astype(np.float64), np.roll(Wx, 1, axis=1))
    assert np.abs(np.einsum("chx,yh->cyx", shifted, weight_matrix(40, 20)) - want).max() > 1e-2
