"""End-to-end throughput of the executed host path (round-4 verdict items 4 / 12 / 13): `python -m densecap_amd.run_model
-input_dir <JPEGs>` and `python -m densecap_amd.extract_features` from image FILES to results, synthetic weights in
checkpoint shapes.  usage (GPU box): python tools/cli_throughput.py [n_images=64] [out.json]"""
import json
import os
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from PIL import Image  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 64
out_path = sys.argv[2] if len(sys.argv) > 2 else os.path.join(ROOT, "gpurun_out", "cli_throughput.json")
src = Image.open(os.path.join(ROOT, "tests", "golden", "elephant_720x480.jpg")).convert("RGB")
rec = {"images": n, "host_cores": os.cpu_count(), "runs": []}
with tempfile.TemporaryDirectory() as td:
    sets = {"720x480 files (already at the network's size)": [(720, 480)] * n,
            "1600x1200 photographs (scaled to 720x540 on the device)": [(1600, 1200)] * n,
            "mixed sizes": [[(720, 480), (1600, 1200), (1024, 768), (480, 640)][i % 4] for i in range(n)]}
    for name, sizes in sets.items():
        d = os.path.join(td, str(len(rec["runs"])))
        os.makedirs(d)
        for i, (w, h) in enumerate(sizes):
            src.resize((w, h)).rotate(i % 7).save(os.path.join(d, "im%03d.jpg" % i), quality=90)
        for extra, label in (([], "run_model"), (["-host_preprocess", "1", "-max_images", "8"], "run_model, host preprocessing (8 images)")):
            if label.startswith("run_model, host") and "1600" not in name:
                continue
            t0 = time.perf_counter()
            p = subprocess.run([sys.executable, "-m", "densecap_amd.run_model", "-input_dir", d, "-synthetic_weights", "1",
                                "-max_images", str(n), "-output_vis_dir", os.path.join(d, "vis"), "-timing", "1"] + extra,
                               cwd=ROOT, capture_output=True, text=True)
            wall = time.perf_counter() - t0
            line = [l for l in p.stdout.splitlines() if l.startswith("TIMING")]
            rec["runs"].append({"what": label, "files": name, "rc": p.returncode, "timing": line[-1] if line else p.stderr[-400:],
                                "process_wall_s": wall})
            print(rec["runs"][-1], flush=True)
    # extract_features.lua on the first set
    d = os.path.join(td, "0")
    t0 = time.perf_counter()
    txt = os.path.join(td, "paths.txt")
    open(txt, "w").write("\n".join(os.path.join(d, f) for f in sorted(os.listdir(d)) if f.endswith(".jpg")) + "\n")
    p = subprocess.run([sys.executable, "-m", "densecap_amd.extract_features", "-input_txt", txt, "-synthetic_weights", "1",
                        "-max_images", str(n), "-output_h5", os.path.join(td, "f.h5"), "-boxes_per_image", "100", "-timing", "1"],
                       cwd=ROOT, capture_output=True, text=True)
    line = [l for l in p.stdout.splitlines() if l.startswith("TIMING")]
    rec["runs"].append({"what": "extract_features", "files": "720x480 files", "rc": p.returncode,
                        "timing": line[-1] if line else (p.stdout + p.stderr)[-400:], "process_wall_s": time.perf_counter() - t0})
    print(rec["runs"][-1], flush=True)
os.makedirs(os.path.dirname(out_path), exist_ok=True)
json.dump(rec, open(out_path, "w"), indent=1)
