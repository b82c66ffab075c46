"""Prints (and writes to gpurun_out/parity_report.json) the end-to-end parity statistics of the HIP path vs the CPU
oracle over the BASELINE.json configurations; the same strict comparison the -m gpu tests assert (tests/parity.py).
usage (GPU box): python tests/parity_report.py   [PARITY_EXTRA=n more 720x600 images] [PARITY_MATH_MODE=1: the opt-in
split-bf16 arithmetic, written to gpurun_out/parity_report_split_bf16.json]"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from densecap_amd import DenseCapModel  # noqa: E402
from densecap_amd.weights import make_synthetic_image, make_synthetic_weights  # noqa: E402
from tests import parity  # noqa: E402

W = make_synthetic_weights(seed=1234)
m = DenseCapModel(W, device=0)
MATH_MODE = int(os.environ.get("PARITY_MATH_MODE", "0"))
if MATH_MODE:
    m.setMathMode(MATH_MODE)
rows = []
SETTINGS = [(600, 720, 1000, 0), (600, 720, 1000, 1), (600, 720, 300, 2), (720, 1080, 2000, 5), (480, 720, 1000, 7),
            (320, 480, 50, 8), (1200, 1600, 1000, 9)]
SETTINGS += [(600, 720, 1000, sd) for sd in range(10, 10 + int(os.environ.get("PARITY_EXTRA", "0")))]
for (H, Wd, P, seed) in SETTINGS:
    r = dict(H=H, W=Wd, P=P, seed=seed)
    r.update(parity.strict_check(m, W, make_synthetic_image(H, Wd, seed), P))
    for key in ("rpn_flips", "final_list_flips"):          # keep the count, and the first few decisions as examples
        if key in r:
            r[key + "_n"] = len(r[key])
            r[key] = r[key][:4]
    rows.append(r)
    print(json.dumps(r, default=str), flush=True)
ks = [r[k] for r in rows for k in ("rpn_k_needed", "final_k_needed") if r.get(k) is not None]
summary = dict(images=len(rows), math_mode=MATH_MODE, replayed_lists=len(ks), max_k_needed=max(ks) if ks else None, FLIP_K=parity.FLIP_K,
               k_ladder=list(parity.K_LADDER),
               note="k_needed = smallest k at which the flip replay reproduces the HIP list: a flipped decision's oracle margin "
                    "over the discrepancy observed for its operands; FLIP_K is set to twice the largest value any report has needed")
print(json.dumps(summary), flush=True)
rows.append(dict(_summary=summary))
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
json.dump(rows, open(os.path.join(ROOT, "gpurun_out", "parity_report_split_bf16.json" if MATH_MODE else "parity_report.json"), "w"),
          indent=1, default=str)
