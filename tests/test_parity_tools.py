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
