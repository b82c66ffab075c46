#!/bin/bash
# usage (GPU box): tools/probes/conv_ab.sh H W Cin Cout "cfg cfg ..."  -> kernel time per call by route
export TMPDIR=/tmp
H=$1; W=$2; CI=$3; CO=$4
for cfg in $5; do
  OUT=/tmp/rp_cab_$cfg; rm -rf $OUT
  rocprofv3 --kernel-trace --stats -d $OUT -o t --output-format csv -- python tools/probes/conv_ab.py $H $W $CI $CO $cfg 200 > /dev/null 2>&1
  python - $OUT $cfg <<'P'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True)[0]
tot = 0.0
parts = []
for r in csv.DictReader(open(f)):
    n = r["Name"]
    if "mfma_gemm" in n or "splitk_reduce" in n:
        per = float(r["TotalDurationNs"]) / 200.0 / 1e3
        tot += per
        parts.append("%s x%.1f %.1f us" % (n.split("(")[0].replace("(anonymous namespace)::", "").replace("void ", "")[:44], int(r["Calls"]) / 200.0, float(r["AverageNs"]) / 1e3))
print("force_cfg %s: %.1f us per call  [%s]" % (sys.argv[2], tot, "; ".join(parts)))
P
done
