"""Per-frame wall time of the webcam daemon's loop body (densecap_amd/daemon.py::process_file: JPEG decode, preprocessing,
forward_test, box rescale, JSON) at the webcam settings (640x480 frames, -max_image_size 480, 50 proposals:
webcam/single_machine_demo.lua:25-26), device preprocessing against the host restatement.
usage (GPU box): python tools/daemon_latency.py [frames] [out.json]"""
import json, os, sys, tempfile, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from PIL import Image
from densecap_amd import DenseCapModel, daemon as D
from densecap_amd.weights import make_synthetic_weights

n = int(sys.argv[1]) if len(sys.argv) > 1 else 40
m = DenseCapModel(make_synthetic_weights(seed=1234), device=0)
m.setLanes(1); m.evaluate()
m.setTestArgs(rpn_nms_thresh=0.7, final_nms_thresh=0.3, num_proposals=50)
rng = np.random.default_rng(0)
out = {"frames": n, "settings": "640x480 JPEG frames, max_image_size 480, 50 proposals, synthetic weights (V = 10497, T = 15), lanes 1"}
with tempfile.TemporaryDirectory() as td:
    for name, host, graph in (("device_preprocess_graph_replay", False, True), ("device_preprocess", False, False), ("host_preprocess", True, False)):
        m.setGraphReplay(graph)
        ts = []
        for i in range(n + 3):
            p = os.path.join(td, "f.jpg")
            Image.fromarray(rng.uniform(0, 255, (480, 640, 3)).astype(np.uint8)).save(p, quality=90)
            t0 = time.perf_counter()
            ok = D.process_file(m, p, os.path.join(td, "f.json"), 480, host)
            ts.append(time.perf_counter() - t0)
            assert ok
        ts = np.array(ts[3:]) * 1e3
        out[name] = {"ms_per_frame_median": float(np.median(ts)), "ms_min": float(ts.min()), "frames_per_s": float(1e3 / np.median(ts))}
        print(name, out[name])
if len(sys.argv) > 2:
    json.dump(out, open(sys.argv[2], "w"), indent=1)
