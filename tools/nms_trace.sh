#!/bin/bash
# Per-kernel durations of the two NMS launch chains of ONE image inside the real forward, single-image mode (round 6, verdict
# item 7): rocprofv3 kernel trace of a few forwards; the kernels between rpn_decode_kernel and bilinear_roi_pool_kernel (RPN NMS)
# and between recog_heads_kernel and survivor_compact_kernel (final NMS) of the last image.
# usage (GPU box, repo root): bash tools/nms_trace.sh <out.txt> [H W P [nms_band]]
set -u
REPO=$(pwd); OUT=$1; H=${2:-600}; W=${3:-720}; P=${4:-1000}; BAND=${5:-1}
export TMPDIR=/tmp
cd /tmp
rm -rf /tmp/rp_nms
cat > /tmp/nms_target.py <<PY
import sys
sys.path.insert(0, "$REPO")
import numpy as np
from densecap_amd import DenseCapModel
from densecap_amd._lib import check
from densecap_amd.weights import make_synthetic_weights, make_synthetic_image
m = DenseCapModel(make_synthetic_weights(seed=1234), device=0)
m.setTestArgs(rpn_nms_thresh=0.7, final_nms_thresh=0.3, num_proposals=$P)
check(m.ctx.h, m.ctx.lib.dc_debug_set(m.ctx.h, b"nms_band", $BAND), "dc_debug_set")
m.setLanes(1); m.setGroup(1); m.setCaptionOrder(True)
dev = m.ctx.to_device(make_synthetic_image($H, $W, 0)[None])
for _ in range(5):
    r = m.forward_batch_device(dev.ptr, 1, $H, $W)
print("K =", [len(x[0]) for x in r], "stages", {k: round(v, 4) for k, v in m.stage_times().items()})
PY
rocprofv3 --kernel-trace --output-format csv -d /tmp/rp_nms -o s -- python /tmp/nms_target.py > /tmp/rp_nms.log 2>&1
f=$(find /tmp/rp_nms -name "s_kernel_trace.csv" | head -1)
{ echo "==== ${W}x${H} P=$P nms_band=$BAND: $(grep 'K =' /tmp/rp_nms.log)"
python - "$f" <<'PY'
import csv, sys, collections
rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Dispatch_Id"]))
names = [r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0] for r in rows]
def chain(title, first, last):
    hi = max(i for i, n in enumerate(names) if n.startswith(last))
    lo = max(i for i, n in enumerate(names[:hi]) if n.startswith(first))
    print("  " + title)
    tot = 0.0
    for i in range(lo + 1, hi):
        d = (int(rows[i]["End_Timestamp"]) - int(rows[i]["Start_Timestamp"])) / 1e3
        tot += d
        print("    %-40s grid=%-8s %8.2f us" % (names[i][:40], rows[i].get("Grid_Size_X", "?"), d))
    span = (int(rows[hi]["Start_Timestamp"]) - int(rows[lo]["End_Timestamp"])) / 1e3
    print("    %d launches, kernels busy %.1f us, span %.1f us (profiler serialisation included)" % (hi - lo - 1, tot, span))
chain("RPN NMS (rpn_decode_kernel .. bilinear_roi_pool_kernel)", "rpn_decode", "bilinear_roi_pool")
chain("final NMS (recog_heads_kernel .. survivor_compact_kernel)", "recog_heads", "survivor_compact")
PY
} >> "$OUT"
cd "$REPO"
tail -45 "$OUT"
