#!/bin/bash
# usage (GPU box): tools/probes/pmc_quick.sh <variant suffix or ""> <gemm_bench args...>  -> MFMA busy fraction and clock per kernel
V=$1; shift
export TMPDIR=/tmp
OUT=gpurun_out/pmcq$V
rm -rf $OUT; mkdir -p $OUT
DENSECAP_HIP_LIB=$PWD/densecap_amd/lib/libdensecap_hip$V.so rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CU_CYCLES SQ_INSTS_VALU -d $OUT -o t --output-format csv -- python tools/gemm_bench.py 3 "$@" > $OUT/log.txt 2>&1
python - "$OUT" <<'P'
import csv, glob, sys, collections
d = sys.argv[1]
f = glob.glob(d + "/**/*counter_collection.csv", recursive=True)
kt = glob.glob(d + "/**/*kernel_trace.csv", recursive=True)
dur = {}
for r in csv.DictReader(open(kt[0])):
    dur[r["Dispatch_Id"]] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"]), r["Kernel_Name"])
acc = collections.defaultdict(lambda: collections.defaultdict(float))
for r in csv.DictReader(open(f[0])):
    acc[r["Dispatch_Id"]][r["Counter_Name"]] += float(r["Counter_Value"])
for did, c in acc.items():
    ns, name = dur.get(did, (0, "?"))
    if "mfma" not in name or ns < 50000: continue
    gui = c.get("GRBM_GUI_ACTIVE", 0)
    if gui <= 0: continue
    print("%-60s %8.1f us  mfma_busy %.3f  clock %.2f GHz  valu_insts/wave-ish %.3g" % (name[:60], ns / 1e3, c["SQ_VALU_MFMA_BUSY_CYCLES"] / (gui * 1024.0), gui / ns, c.get("SQ_INSTS_VALU", 0)))
P
