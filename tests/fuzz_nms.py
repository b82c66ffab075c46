"""Stress of the two NMS scan kernels against each other (round 6): random list lengths, cluster sizes, thresholds, caps and
valid masks; every list through nms_scan_band_kernel (the default for windows of <= 4096 rows) and through nms_scan_kernel
(dc_debug_set "nms_band" 0) must give the same picks; every 25th case is also checked against the oracle.
usage (GPU box): python tests/fuzz_nms.py [cases] [seed]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np


def main():
    from densecap_amd import ops
    from densecap_amd._lib import check
    from oracle import densecap_oracle as O
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 500
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
    ctx = ops.Context(0)
    t0 = time.time()
    stats = {"cases": 0, "oracle_checked": 0, "max_picks": 0, "zero_pick_lists": 0}
    try:
        for c in range(cases):
            n = int(rng.choice([rng.integers(1, 200), rng.integers(200, 4097), rng.integers(4097, 12000), rng.integers(12000, 40000)],
                               p=[0.15, 0.5, 0.25, 0.1]))
            per = int(rng.choice([1, 2, 5, 20, 100, 700, max(1, n)]))
            ncl = (n + per - 1) // per
            cxy = rng.uniform(0, rng.choice([300, 3000]), (ncl, 1, 2)); wh = rng.uniform(5, 200, (ncl, 1, 2))
            jit = rng.choice([0.5, 6.0, 40.0])
            xy = cxy + rng.uniform(-jit, jit, (ncl, per, 2))
            b = np.concatenate([xy, xy + wh + rng.uniform(-jit, jit, (ncl, per, 2))], 2).reshape(-1, 4)[:n]
            sc = rng.uniform(0, 1, (n, 1))
            if rng.uniform() < 0.5:
                sc = np.round(sc, int(rng.integers(1, 4)))
            b5 = np.concatenate([b, sc], 1).astype(np.float32)
            thr = float(rng.choice([0.05, 0.3, 0.5, 0.7, 0.95]))
            maxb = None if rng.uniform() < 0.4 else int(rng.choice([1, 50, 300, 1000, 2000]))
            valid = None if rng.uniform() < 0.7 else rng.uniform(size=n) > rng.uniform(0, 0.9)
            check(ctx.h, ctx.lib.dc_debug_set(ctx.h, b"nms_band", 0), "dc_debug_set")
            chunk = ops.nms(ctx, b5, thr, maxb, valid=valid)
            check(ctx.h, ctx.lib.dc_debug_set(ctx.h, b"nms_band", 1), "dc_debug_set")
            band = ops.nms(ctx, b5, thr, maxb, valid=valid)
            assert np.array_equal(chunk, band), ("band != chunk scan", c, n, per, thr, maxb)
            if c % 25 == 0 and n <= 12000:
                rows = np.arange(n) if valid is None else np.nonzero(valid)[0]
                ref = rows[O.nms(b5[rows], thr, maxb)] if len(rows) else np.zeros(0, np.int64)
                assert band.tolist() == ref.tolist(), ("band != oracle", c, n, per, thr, maxb)
                stats["oracle_checked"] += 1
            stats["cases"] += 1
            stats["max_picks"] = max(stats["max_picks"], len(band))
            stats["zero_pick_lists"] += int(len(band) == 0)
    finally:
        check(ctx.h, ctx.lib.dc_debug_set(ctx.h, b"nms_band", 1), "dc_debug_set")
        ctx.close()
    stats["seconds"] = round(time.time() - t0, 1)
    print("fuzz_nms ok:", stats)


if __name__ == "__main__":
    main()
