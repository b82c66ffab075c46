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
# one lane = one stream for the profiled images (per-launch events switch the two-stream decode off); the short legs only
LEAN="--no-cpu-baseline --no-alt-pass --no-host-input-leg --sustain-seconds 0 --no-settle --no-split-leg --no-traffic-leg --no-config-legs --gather torch"
BENCH="python $REPO/bench.py --lanes 1 --group 1 $LEAN"
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
# PROFILE WHAT IS TIMED: one stream (per-kernel counters need non-overlapping kernels) but the MULTI-LANE planning of the
# headline schedule (--plan-mode 0): conv4_2 / conv4_3 on whole K-split tiles, no stream-K, no tail plans
BENCH="python $REPO/bench.py --lanes 1 --plan-mode 0 --group 1 $LEAN"
run mlplan "--steps 10 --warmup 3 --repeats 1" --kernel-trace --stats
grep '^{' /tmp/rp_mlplan.json > "$OUT/bench_mlplan.json"
run mlplan_fetch "--steps 3 --warmup 1 --repeats 1" --kernel-trace --pmc FETCH_SIZE
run mlplan_write "--steps 3 --warmup 1 --repeats 1" --kernel-trace --pmc WRITE_SIZE
run mlplan_mfma "--steps 3 --warmup 1 --repeats 1" --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_LDS_BANK_CONFLICT
# the DEFAULT (multi-lane) schedule -- the one the headline number comes from -- under the kernel trace as well
# (rocprofv3 serialises dispatches, so the overlap itself is not visible; per-kernel durations and counts are)
BENCH="python $REPO/bench.py $LEAN"
run default "--steps 10 --warmup 3 --repeats 2" --kernel-trace --stats
grep '^{' /tmp/rp_default.json > "$OUT/bench_default_under_rocprof.json"
# round 6: the captions-after-the-final-NMS schedule (what the CLIs run; `value_captions_after_final_nms`): one stream, the
# multi-lane planning and the group of eight the timed leg picks -- the packed decode launches of a group are in this trace
BENCH="python $REPO/bench.py --lanes 1 --plan-mode 0 --group 8 --caption-order 1 $LEAN"
run capnms "--steps 16 --warmup 8 --repeats 1" --kernel-trace --stats
grep '^{' /tmp/rp_capnms.json > "$OUT/bench_capnms_under_rocprof.json"
run capnms_mfma "--steps 8 --warmup 8 --repeats 1" --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_LDS_BANK_CONFLICT
# the opt-in split-bf16 mode (round 5): kernel names + durations, MFMA busy and clock of ITS kernels (one stream, multi-lane planning)
BENCH="python $REPO/bench.py --lanes 1 --plan-mode 0 --group 1 --math-mode 1 $LEAN"
run split "--steps 10 --warmup 3 --repeats 1" --kernel-trace --stats
grep '^{' /tmp/rp_split.json > "$OUT/bench_split_lanes1.json"
run split_mfma "--steps 3 --warmup 1 --repeats 1" --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_LDS_BANK_CONFLICT
run split_fetch "--steps 3 --warmup 1 --repeats 1" --kernel-trace --pmc FETCH_SIZE
run split_write "--steps 3 --warmup 1 --repeats 1" --kernel-trace --pmc WRITE_SIZE
cd "$REPO"
python bench.py > "$OUT/bench_default.json" 2> "$OUT/bench_default.err"
Q="--no-cpu-baseline --sustain-seconds 2 --repeats 3 --no-traffic-leg"
python bench.py --height 320 --width 480 --proposals 50 --lanes 1 --steps 30 --no-alt-pass $Q > "$OUT/bench_webcam_480_p50.json" 2>/dev/null
python bench.py --height 480 --width 720 --proposals 1000 --steps 32 $Q > "$OUT/bench_config0_720x480.json" 2>/dev/null
python bench.py --proposals 300 --steps 32 $Q > "$OUT/bench_config3_p300.json" 2>/dev/null     # (images per group picked by the untimed trial)
python bench.py --height 720 --width 1080 --proposals 2000 --steps 12 --warmup 2 $Q > "$OUT/bench_config5.json" 2>/dev/null
python bench.py --math-mode 1 --steps 32 --no-split-leg $Q > "$OUT/bench_split_bf16_mode.json" 2>/dev/null
python tools/latency_check.py 2>/dev/null | grep -v amdgpu.ids > "$OUT/latency_check.txt"
bash tools/decode_trace.sh "$OUT/decode_trace_small_rows.txt" 13 50 128 221 256 300 900 > /dev/null 2>&1
for cfg in "600 720 1000 1" "600 720 1000 4" "600 720 1000 8" "600 720 300 1" "600 720 300 8" "320 480 50 1"; do
  bash tools/survivor_trace.sh "$OUT/survivor_decode_trace.txt" $cfg > /dev/null 2>&1
done
for cfg in "600 720 1000" "600 720 300" "320 480 50" "720 1080 2000"; do
  for band in 0 1; do bash tools/nms_trace.sh "$OUT/nms_chain_trace.txt" $cfg $band > /dev/null 2>&1; done
done
for seed in 21 22; do timeout 300 python tests/fuzz_nms.py 8000 $seed 2>&1 | tail -1 >> "$OUT/fuzz_nms.txt"; done
rm -rf /tmp/rp_webcam; ( cd /tmp && rocprofv3 --kernel-trace --output-format csv -d /tmp/rp_webcam -o w -- python $REPO/bench.py --height 320 --width 480 --proposals 50 --lanes 1 --group 1 --steps 10 --warmup 3 --repeats 1 --caption-order 1 $LEAN > /dev/null 2>&1 )
python tools/layer_times.py $(find /tmp/rp_webcam -name "w_kernel_trace.csv" | head -1) --height 320 --width 480 --proposals 50 > "$OUT/webcam_layers.txt" 2>&1
python tools/gemm_bench.py 5 --serial > "$OUT/gemm_bench_serial.txt" 2>/dev/null
python tools/gemm_bench.py 5 > "$OUT/gemm_bench_multilane.txt" 2>/dev/null
python tools/gemm_bench.py 5 --math-mode=1 > "$OUT/gemm_bench_split_bf16.txt" 2>/dev/null
python tools/cli_throughput.py 64 "$OUT/cli_throughput.json" > "$OUT/cli_throughput.log" 2>&1
./tools/probes/bin/mfma_probe > "$OUT/mfma_bf16_numerics.txt" 2>&1
python -m pytest tests/test_gpu_bf3.py -m gpu -q -s -k "linear_error or conv3x3_error or zz_print" 2>&1 | grep "split-bf16 error" > "$OUT/split_bf16_error_ratios.txt"
python tools/decode_bench.py 20 1000 300 50 > "$OUT/decode_bench.txt" 2>/dev/null
python tools/decode_bench.py 5 1000 50 --beam=5 >> "$OUT/decode_bench.txt" 2>/dev/null
PARITY_EXTRA=8 python tests/parity_report.py > "$OUT/parity_report.log" 2>&1
cp gpurun_out/parity_report.json "$OUT/parity_report.json" 2>/dev/null
ls -la "$OUT"
