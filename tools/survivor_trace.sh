#!/bin/bash
# Per-kernel durations of ONE image's packed decode (captions after the final NMS) inside the real forward, single-image mode:
# rocprofv3 kernel trace of a few forwards, kernels between survivor_compact_kernel and final_pack_kernel of the last image.
# usage (GPU box, repo root): bash tools/survivor_trace.sh <out.txt> [H W P [group]]
set -u
REPO=$(pwd); OUT=$1; H=${2:-600}; W=${3:-720}; P=${4:-1000}; G=${5:-1}
export TMPDIR=/tmp
cd /tmp
rm -rf /tmp/rp_surv
cat > /tmp/surv_target.py <<PY
import sys
sys.path.insert(0, "$REPO")
import numpy as np
from densecap_amd import DenseCapModel
from densecap_amd.weights import make_synthetic_weights, make_synthetic_image
m = DenseCapModel(make_synthetic_weights(seed=1234), device=0)
m.setTestArgs(rpn_nms_thresh=0.7, final_nms_thresh=0.3, num_proposals=$P)
G = $G
m.setLanes(1 if G == 1 else 2); m.setGroup(G); m.setCaptionOrder(True)
imgs = np.stack([make_synthetic_image($H, $W, s) for s in range(G)])
dev = m.ctx.to_device(imgs)
for _ in range(5):
    r = m.forward_batch_device(dev.ptr, G, $H, $W)
print("K =", [len(x[0]) for x in r], "stages", m.stage_times())
PY
rocprofv3 --kernel-trace --output-format csv -d /tmp/rp_surv -o s -- python /tmp/surv_target.py > /tmp/rp_surv.log 2>&1
f=$(find /tmp/rp_surv -name "s_kernel_trace.csv" | head -1)
{ echo "==== ${W}x${H} P=$P group=$G: $(grep 'K =' /tmp/rp_surv.log)"
python - "$f" <<'PY'
import csv, sys, collections
rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Dispatch_Id"]))
names = [r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0] for r in rows]
lo = max(i for i, n in enumerate(names) if n.startswith("survivor_compact"))
hi = max(i for i, n in enumerate(names) if n.startswith("final_pack"))
agg = collections.OrderedDict()
for i in range(lo, hi + 1):
    d = (int(rows[i]["End_Timestamp"]) - int(rows[i]["Start_Timestamp"])) / 1e3
    k = names[i][:64] + " grid=" + rows[i].get("Grid_Size_X", "?")
    a = agg.setdefault(k, [0, 0.0]); a[0] += 1; a[1] += d
span = (int(rows[hi]["End_Timestamp"]) - int(rows[lo]["Start_Timestamp"])) / 1e3
busy = sum(v[1] for v in agg.values())
for k, (c, u) in agg.items():
    print("  %-92s x%-3d %8.1f us total %7.2f us each" % (k, c, u, u / c))
print("  span %.1f us, kernels busy %.1f us, gaps %.1f us (profiler serialisation included)" % (span, busy, span - busy))
PY
} >> "$OUT"
cd "$REPO"
tail -20 "$OUT"
