"""Single-image latency and one-lane frame rate with and without graph replay (dc_set_graph_replay), interleaved.
usage: python tools/graph_replay_bench.py [reps]"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402

from densecap_amd import DenseCapModel  # noqa: E402
from densecap_amd.weights import make_synthetic_image, make_synthetic_weights  # noqa: E402


def main(reps):
    m = DenseCapModel(make_synthetic_weights(seed=1234), device=0)
    rows = []
    print("%-22s %-6s %12s %12s %8s" % ("workload", "lanes", "eager ms", "graph ms", "ratio"))
    for name, H, W, P in (("webcam 480x320 P=50", 320, 480, 50), ("720x480 P=300", 480, 720, 300), ("720x600 P=1000", 600, 720, 1000)):
        m.setTestArgs(rpn_nms_thresh=0.7, final_nms_thresh=0.3, num_proposals=P)
        imgs = np.stack([make_synthetic_image(H, W, i) for i in range(8)])
        dev = m.ctx.to_device(imgs)
        for lanes in (1, 2):
            m.setLanes(lanes)
            t = {0: [], 1: []}
            for g in (0, 1):
                m.setGraphReplay(g)
                for _ in range(3):
                    m.forward_batch_device(dev.ptr, 8, H, W)
            for _ in range(reps):
                for g in (0, 1):
                    m.setGraphReplay(g)
                    t0 = time.perf_counter()
                    m.forward_batch_device(dev.ptr, 8, H, W)
                    t[g].append((time.perf_counter() - t0) / 8 * 1e3)
            e, gr = float(np.median(t[0])), float(np.median(t[1]))
            rows.append(dict(workload=name, lanes=lanes, eager_ms_per_image=e, graph_ms_per_image=gr))
            print("%-22s %-6d %12.3f %12.3f %8.3f" % (name, lanes, e, gr, e / gr), flush=True)
        # one image at a time (run_model on a single file, the daemon): call-to-result latency
        m.setLanes(1)
        one = imgs[0]
        t = {0: [], 1: []}
        for g in (0, 1):
            m.setGraphReplay(g)
            for _ in range(3):
                m.forward_raw(one)
        for _ in range(reps * 4):
            for g in (0, 1):
                m.setGraphReplay(g)
                t0 = time.perf_counter()
                m.forward_raw(one)
                t[g].append((time.perf_counter() - t0) * 1e3)
        e, gr = float(np.median(t[0])), float(np.median(t[1]))
        rows.append(dict(workload=name, lanes=0, eager_ms_per_image=e, graph_ms_per_image=gr, note="one forward_test call, host image"))
        print("%-22s %-6s %12.3f %12.3f %8.3f" % (name, "call", e, gr, e / gr), flush=True)
        dev.free()
    m.setGraphReplay(0)
    out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
    os.makedirs(out, exist_ok=True)
    json.dump(rows, open(os.path.join(out, "graph_replay_bench.json"), "w"), indent=0)
    m.ctx.close()


if __name__ == "__main__":
    main(int(sys.argv[1]) if len(sys.argv) > 1 else 6)
