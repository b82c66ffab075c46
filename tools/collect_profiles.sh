#!/bin/bash
# Collects the rocprofv3 evidence for profiles/ on the GPU box (run through gpurun from the repo root):
#   gpurun --timeout 1500 -- 'bash tools/collect_profiles.sh gpurun_out/prof_rNN'
# then, back in the container:  python tools/pmc_summary.py gpurun_out/prof_rNN profiles/rNN
# Counter passes are separate runs with --kernel-trace + --pmc only (no sys/runtime/hip/hsa traces).
set -u
OUT=$(realpath -m "${1:-gpurun_out/prof}")
REPO=$(pwd)
mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
# one stream throughout (the un-timed setup/warm-up images would otherwise use the two-stream decode of single-image mode)
export DENSECAP_NO_DECODE_SPLIT=1
BENCH="python $REPO/bench.py --lanes 1 --no-cpu-baseline --no-alt-pass"
run() {  # name, bench args, rocprof args...
  local name=$1 args=$2; shift 2
  rm -rf /tmp/rp_$name
  timeout 600 rocprofv3 "$@" --output-format csv -d /tmp/rp_$name -o $name -- $BENCH $args > /tmp/rp_$name.json 2> /tmp/rp_$name.err
  find /tmp/rp_$name -name "${name}_*.csv" -exec cp {} "$OUT"/ \;
}
run lanes1 "--steps 10 --warmup 3" --kernel-trace --stats
grep '^{' /tmp/rp_lanes1.json > "$OUT/bench_lanes1.json"
run fetch "--steps 3 --warmup 1" --kernel-trace --pmc FETCH_SIZE
run write "--steps 3 --warmup 1" --kernel-trace --pmc WRITE_SIZE
run mfma "--steps 3 --warmup 1" --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_LDS_BANK_CONFLICT
cd "$REPO"
unset DENSECAP_NO_DECODE_SPLIT
python bench.py > "$OUT/bench_default.json" 2> "$OUT/bench_default.err"
ls -la "$OUT"
