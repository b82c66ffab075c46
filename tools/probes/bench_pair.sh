#!/bin/bash
# split-bf16 vs fp32 end to end, short: prints value / lanes / group / stage times of each
for mm in 1 0; do
  python bench.py --steps 32 --warmup 3 --repeats 3 --sustain-seconds 0 --no-cpu-baseline --no-host-input-leg --no-alt-pass --math-mode $mm > gpurun_out/bp_$mm.log 2>&1
  grep '^{' gpurun_out/bp_$mm.log | tail -1 > gpurun_out/bp_$mm.json
  python - $mm <<'P'
import json, sys
mm = sys.argv[1]
try:
    d = json.load(open("gpurun_out/bp_%s.json" % mm))
except Exception as e:
    print("math mode", mm, "no JSON line:", e); print(open("gpurun_out/bp_%s.log" % mm).read()[-1500:]); sys.exit(0)
print("math mode", mm, "value %.1f" % d["value"], "lanes", d["lanes"], "group", d["group"], ["%.1f" % v for v in d["repeats"]["images_per_s"]], d["config"]["gather"])
print("   serial stage ms", {k: round(v, 3) for k, v in d["stage_ms_serial_image"].items()})
print("   single-image latency ms", d["roofline"].get("single_image_mode", {}).get("latency_ms"))
P
done
