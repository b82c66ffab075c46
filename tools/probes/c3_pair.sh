#!/bin/bash
# usage (GPU box): tools/probes/c3_pair.sh <variant suffix>  -> conv1_1 kernel time (rocprofv3 kernel trace), product library
# against libdensecap_hip<suffix>.so, three alternations; then the conv1_1 parity test on the variant
V=$1
export TMPDIR=/tmp
for rep in 1 2 3; do
  for L in "" "$V"; do
    OUT=gpurun_out/c3pair/r$rep$L; rm -rf $OUT; mkdir -p $OUT
    DENSECAP_HIP_LIB=$PWD/densecap_amd/lib/libdensecap_hip$L.so rocprofv3 --kernel-trace --stats -d $OUT -o t --output-format csv -- python tools/c3_bench.py 200 > $OUT/log.txt 2>&1
    python - "$OUT" "lib$L" <<'P'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True)
for r in csv.DictReader(open(f[0])):
    if "conv3x3_c3" in r["Name"]:
        print("%-24s calls %s avg %.2f us min %.2f us" % (sys.argv[2], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3))
P
  done
done
DENSECAP_HIP_LIB=$PWD/densecap_amd/lib/libdensecap_hip$V.so python -m pytest tests/test_gpu_ops.py -q -k conv1_1 2>&1 | tail -2
