"""The parity checker's own tools (tests/parity.py), on CPU: the flip replay must reproduce a perturbed NMS run exactly
and name the decisions that flipped; with no perturbation it must be the oracle's NMS with no flips."""
import numpy as np


def _boxes5(rng, n):
    xy = rng.uniform(0, 600, (n, 2)); wh = rng.uniform(10, 300, (n, 2))
    return np.concatenate([xy, xy + wh, rng.uniform(0, 1, (n, 1))], 1).astype(np.float32)


def test_hybrid_nms_is_the_oracle_nms_without_discrepancy():
    from oracle import densecap_oracle as O
    from tests import parity
    b = _boxes5(np.random.default_rng(0), 800)
    picks, flips = parity.hybrid_nms(b, b.copy(), 0.5, 120)
    assert picks.tolist() == O.nms(b, 0.5, 120).tolist() and flips == []


def test_hybrid_nms_replays_a_perturbed_run_and_names_the_flips():
    from oracle import densecap_oracle as O
    from tests import parity
    rng = np.random.default_rng(1)
    differing = 0
    for trial in range(30):
        b = _boxes5(rng, 600)
        b[:, 4] = np.round(b[:, 4], 2)                           # many near-equal scores
        h = b.copy()
        h[:, :4] += rng.normal(0, 2e-3, (600, 4)).astype(np.float32)     # ~1e-5 relative coordinate noise
        h[:, 4] += rng.normal(0, 3e-6, 600).astype(np.float32)
        ref_o, ref_h = O.nms(b, 0.5, None), O.nms(h, 0.5, None)
        picks, flips = parity.hybrid_nms(b, h, 0.5, None)
        assert picks.tolist() == ref_h.tolist()                  # the replay IS the perturbed run
        if ref_o.tolist() != ref_h.tolist():
            differing += 1
            assert flips, "lists differ but no flipped decision was reported"
    assert differing > 0                                         # the scenario the replay exists for did occur


def test_hybrid_nms_admits_a_box_that_moves_past_several_neighbours():
    """Round 6 (tests/fuzz_e2e.py seeds 65 / 67, uncapped lists of 20,000+ boxes at threshold 1.0): the admissible rankings are
    those whose every INVERSION against the oracle lies within k x the discrepancy observed for its two scores -- a box whose
    own score moved a lot may pass neighbours that are themselves rock solid, which runs of fragile ADJACENT pairs cannot
    express.  And an inversion beyond that margin must still be refused."""
    from tests import parity
    n = 40
    boxes = np.stack([np.arange(n) * 50.0, np.zeros(n), np.arange(n) * 50.0 + 10, np.full(n, 10.0)], 1)     # disjoint boxes
    so = (1.0 - np.arange(n) * 1e-5).astype(np.float32)                       # oracle scores, 1e-5 apart
    sh = so.copy()
    sh[20] += np.float32(3.5e-5)                                                # box 20 moves up past 19, 18, 17 (3 neighbours)
    bo = np.concatenate([boxes, so[:, None]], 1).astype(np.float32)
    bh = np.concatenate([boxes, sh[:, None]], 1).astype(np.float32)
    want = np.lexsort((np.arange(n), -sh.astype(np.float64)))
    assert want.tolist().index(20) == 17
    picks, flips = parity.hybrid_nms(bo, bh, 1.0, None, k=1.0)                 # gap 3e-5 <= 1.0 x (3.5e-5 + 0) for every inversion
    assert picks.tolist() == want.tolist() and len(flips) == 1
    assert parity.k_needed(bo, bh, 1.0, None, want) == 1.0
    picks, _ = parity.hybrid_nms(bo, bh, 1.0, None, k=0.5)                     # 3e-5 > 0.5 x 3.5e-5: not admissible at k = 0.5
    assert picks.tolist() != want.tolist()
    # a swap with NO observed discrepancy to justify it is refused whatever k
    sh2 = so.copy(); sh2[[5, 9]] = sh2[[9, 5]]
    bh2 = np.concatenate([boxes, sh2[:, None]], 1).astype(np.float32)
    want2 = np.lexsort((np.arange(n), -sh2.astype(np.float64)))
    err = np.abs(sh2 - so)
    assert err[5] > 0                                                          # (the swap itself is the discrepancy: 4e-5 each)
    picks, _ = parity.hybrid_nms(bo, bh2, 1.0, None, k=0.25)                   # gap 4e-5 > 0.25 x 8e-5
    assert picks.tolist() != want2.tolist()
