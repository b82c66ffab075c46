#!/bin/bash
# One GPU-box session of the round-4 measurement campaign (run through gpurun from the repo root):
#   gpurun --timeout 1500 -- 'bash tools/lab_session.sh [tests] [kernels] [e2e] [pmc]'
set -u
REPO=$(pwd)
OUT=$REPO/gpurun_out
mkdir -p "$OUT"
export TMPDIR=/tmp
WHAT="${*:-tests kernels e2e pmc}"
for w in $WHAT; do
  case $w in
    tests)   timeout 900 python -m pytest tests -m gpu -x -q > "$OUT/lab_tests.log" 2>&1; tail -5 "$OUT/lab_tests.log" ;;
    kernels) timeout 600 python tools/kernel_lab.py kernels > "$OUT/lab_kernels.txt" 2>&1; tail -3 "$OUT/lab_kernels.txt" ;;
    e2e)     timeout 600 python tools/kernel_lab.py e2e > "$OUT/lab_e2e.txt" 2>&1; tail -40 "$OUT/lab_e2e.txt" ;;
    ab_*)    # end-to-end A/B of one debug knob: ab_<knob>_<a>_<b>
      k=${w#ab_}; b=${k##*_}; k=${k%_*}; a=${k##*_}; k=${k%_*}
      timeout 600 python tools/kernel_lab.py ab $k $a $b > "$OUT/lab_ab_$k.txt" 2>&1; grep -v amdgpu.ids "$OUT/lab_ab_$k.txt" | tail -12 ;;
    knob_*)  # steady-state kernels under one debug knob (bit-identity checked first): knob_<knob>_<a>_<b>
      k=${w#knob_}; b=${k##*_}; k=${k%_*}; a=${k##*_}; k=${k%_*}
      timeout 600 python tools/kernel_lab.py knob $k $a $b > "$OUT/lab_knob_$k.txt" 2>&1; grep -v amdgpu.ids "$OUT/lab_knob_$k.txt" | tail -60 ;;
    bench)   timeout 600 python bench.py > "$OUT/lab_bench_default.json" 2> "$OUT/lab_bench_default.err"; tail -c 3000 "$OUT/lab_bench_default.json"
             timeout 600 python bench.py --proposals 300 --steps 32 --no-cpu-baseline --sustain-seconds 3 --repeats 5 > "$OUT/lab_bench_p300.json" 2> "$OUT/lab_bench_p300.err"; tail -c 1500 "$OUT/lab_bench_p300.json" ;;
    steady)  # every experiment build of the library (make -C densecap_amd/csrc variants) on the same steady-state shapes
      timeout 300 python tools/kernel_lab.py steady base > "$OUT/lab_steady.txt" 2>&1
      for so in densecap_amd/lib/libdensecap_hip_*.so; do
        v=$(basename $so .so); v=${v#libdensecap_hip_}
        DENSECAP_HIP_LIB=$REPO/$so timeout 300 python tools/kernel_lab.py steady $v >> "$OUT/lab_steady.txt" 2>&1
      done
      grep -v "amdgpu.ids" "$OUT/lab_steady.txt" | tail -120 ;;
    pmcvar)  # cycles vs wall time of the steady-state shapes for every library build (DVFS give-back of the ablations)
      cd /tmp
      export LAB_PMC_REPS=6
      for so in base $REPO/densecap_amd/lib/libdensecap_hip_*.so; do
        v=base; [ "$so" != base ] && { v=$(basename $so .so); v=${v#libdensecap_hip_}; export DENSECAP_HIP_LIB=$so; }
        rm -rf /tmp/rp_var_$v
        timeout 300 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES --output-format csv -d /tmp/rp_var_$v -o var_$v -- python $REPO/tools/kernel_lab.py pmc-target b > /tmp/rp_var_$v.log 2>&1
        find /tmp/rp_var_$v -name "var_${v}_counter_collection.csv" -exec cp {} "$OUT"/ \;
        find /tmp/rp_var_$v -name "var_${v}_kernel_trace.csv" -exec cp {} "$OUT"/ \;
        tail -1 /tmp/rp_var_$v.log
      done
      unset DENSECAP_HIP_LIB
      cd "$REPO" ;;
    pmc)
      cd /tmp
      rocprofv3 -L > "$OUT/rocprof_counters.txt" 2>&1
      i=0
      for set in "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES" \
                 "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY" \
                 "SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_MISC" \
                 "SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT"; do
        i=$((i+1))
        rm -rf /tmp/rp_lab$i
        timeout 300 rocprofv3 --kernel-trace --pmc $set --output-format csv -d /tmp/rp_lab$i -o lab$i -- python $REPO/tools/kernel_lab.py pmc-target a > /tmp/rp_lab$i.log 2>&1
        find /tmp/rp_lab$i -name "lab${i}_*.csv" -exec cp {} "$OUT"/ \;
        tail -2 /tmp/rp_lab$i.log
      done
      cd "$REPO" ;;
  esac
done
ls -la "$OUT" | tail -30
