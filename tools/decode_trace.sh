#!/bin/bash
# Per-kernel durations of the greedy decode alone at small row counts (round 6, item 3): rocprofv3 kernel trace of
# tools/decode_bench.py, summarised per kernel.   usage (GPU box, repo root): bash tools/decode_trace.sh <out.txt> rows...
set -u
REPO=$(pwd); OUT=$1; shift
export TMPDIR=/tmp
cd /tmp
: > "$OUT"
for n in "$@"; do
  rm -rf /tmp/rp_dec_$n
  rocprofv3 --kernel-trace --output-format csv -d /tmp/rp_dec_$n -o dec -- python $REPO/tools/decode_bench.py 5 $n > /tmp/rp_dec_$n.log 2>&1
  f=$(find /tmp/rp_dec_$n -name "dec_kernel_trace.csv" | head -1)
  echo "==== rows=$n ($(grep rows= /tmp/rp_dec_$n.log | tail -1))" >> "$OUT"
  python - "$f" >> "$OUT" <<'PY'
import csv, sys, collections
rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Dispatch_Id"]))
# the LAST decode call: find the last lstm_step_tail run of 17 launches
names = [r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0] for r in rows]
tails = [i for i, n in enumerate(names) if n.startswith("lstm_step_tail")]
last = tails[-17:]
lo = last[0] - 2
hi = last[-1]
t0 = int(rows[lo]["Start_Timestamp"])
agg = collections.OrderedDict()
for i in range(lo, hi + 1):
    d = (int(rows[i]["End_Timestamp"]) - int(rows[i]["Start_Timestamp"])) / 1e3
    k = names[i][:60] + " grid=" + rows[i]["Grid_Size_X"] if "Grid_Size_X" in rows[i] else names[i][:60]
    a = agg.setdefault(k, [0, 0.0])
    a[0] += 1; a[1] += d
span = (int(rows[hi]["End_Timestamp"]) - t0) / 1e3
busy = sum(v[1] for v in agg.values())
for k, (c, u) in agg.items():
    print("  %-90s x%-3d %8.1f us total %7.2f us each" % (k, c, u, u / c))
print("  span of the decode %.1f us, kernels busy %.1f us, gaps %.1f us" % (span, busy, span - busy))
PY
done
cd "$REPO"
cat "$OUT"
