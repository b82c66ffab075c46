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
run lanes1 "--steps 10 --warmup 3 --repeats 1" --kernel-trace --stats
grep '^{' /tmp/rp_lanes1.json > "$OUT/bench_lanes1.json"
run fetch "--steps 3 --warmup 1 --repeats 1" --kernel-trace --pmc FETCH_SIZE
run write "--steps 3 --warmup 1 --repeats 1" --kernel-trace --pmc WRITE_SIZE
run mfma "--steps 3 --warmup 1 --repeats 1" --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_LDS_BANK_CONFLICT
# the DEFAULT (multi-lane) schedule -- the one the headline number comes from -- under the kernel trace as well
# (rocprofv3 serialises dispatches, so the overlap itself is not visible; per-kernel durations and counts are)
unset DENSECAP_NO_DECODE_SPLIT
BENCH="python $REPO/bench.py --no-cpu-baseline --no-alt-pass"
run default "--steps 10 --warmup 3 --repeats 2" --kernel-trace --stats
grep '^{' /tmp/rp_default.json > "$OUT/bench_default_under_rocprof.json"
cd "$REPO"
python bench.py > "$OUT/bench_default.json" 2> "$OUT/bench_default.err"
python bench.py --height 320 --width 480 --proposals 50 --lanes 1 --steps 30 --no-cpu-baseline --no-alt-pass > "$OUT/bench_webcam_480_p50.json" 2>/dev/null
python bench.py --proposals 300 --steps 32 --no-cpu-baseline > "$OUT/bench_config3_p300.json" 2>/dev/null
python bench.py --height 720 --width 1080 --proposals 2000 --steps 12 --warmup 2 --no-cpu-baseline > "$OUT/bench_config5.json" 2>/dev/null
ls -la "$OUT"
