#!/bin/bash
# latency regime after the launch folding: webcam 480x320 / P=50 on one lane (with and without graph replay), single-image 720x600
Q="--no-cpu-baseline --sustain-seconds 0 --repeats 3 --no-alt-pass --no-host-input-leg"
python bench.py --height 320 --width 480 --proposals 50 --lanes 1 --steps 30 $Q > gpurun_out/lat_webcam.json 2>gpurun_out/lat_webcam.err
python bench.py --proposals 300 --steps 32 $Q > gpurun_out/lat_p300.json 2>/dev/null
python bench.py --steps 32 $Q > gpurun_out/lat_default.json 2>/dev/null
python - <<'P'
import json
for n in ("webcam", "p300", "default"):
    try:
        d = json.loads([l for l in open("gpurun_out/lat_%s.json" % n) if l.startswith("{")][-1])
    except Exception as e:
        print(n, "failed", e); continue
    r = d["roofline"]
    print(n, "value %.1f" % d["value"], "ms/step %.3f" % d["ms_per_step"], "lanes", d["lanes"], "group", d["group"],
          "single-image latency", (r.get("single_image_mode") or {}).get("latency_ms"), "launches/img", r.get("launches_per_image"))
    print("   stages", {k: round(v, 4) for k, v in d["stage_ms_serial_image"].items()}, d.get("hbm_stages", {}).get("bilinear_roi_pool"))
P
python tools/graph_replay_bench.py 2>&1 | tail -6
