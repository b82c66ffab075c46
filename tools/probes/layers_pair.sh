#!/bin/bash
# usage (GPU box): tools/probes/layers_pair.sh <variant suffix> [bench size flags]  -> per-layer times (tools/layer_times.py) of
# the single-image schedule under the kernel trace, product library against libdensecap_hip<suffix>.so
V=$1; shift
export TMPDIR=/tmp
R=$PWD
for L in "" "$V"; do
  rm -rf /tmp/rp_lp$L
  (cd /tmp && DENSECAP_HIP_LIB=$R/densecap_amd/lib/libdensecap_hip$L.so rocprofv3 --kernel-trace --output-format csv -d /tmp/rp_lp$L -o lp -- python $R/bench.py --lanes 1 --group 1 --steps 10 --warmup 3 --repeats 1 --no-cpu-baseline --no-alt-pass --no-host-input-leg --sustain-seconds 0 --no-settle --no-split-leg --gather torch "$@" > /tmp/lp$L.json 2>/tmp/lp$L.err)
  find /tmp/rp_lp$L -name "lp_kernel_trace.csv" -exec cp {} gpurun_out/lp${L}_kernel_trace.csv \;
  echo "== lib$L"
  python tools/layer_times.py gpurun_out/lp${L}_kernel_trace.csv $LT_FLAGS | grep -v "decode_step_[2-9]\|decode_step_1[0-4]"
done
