"""The MFMA contraction launches of ONE image, in enqueue order, with what each should cost (no GPU needed).

For every launch: the layer it computes, the kernel family the planner routes it to (dc_debug_plan_gemm, the same pure
function run_gemm acts on), its algorithmic FLOPs and its algorithmic HBM bytes (operands read once + result written once,
fp32).  tools/pmc_summary.py lays this list over the dispatch order of a `bench.py --lanes 1` rocprofv3 trace to get a
per-LAYER table (duration, MFMA-busy, clock, fabric bytes vs algorithmic bytes).

usage: python tools/launch_list.py [--serial-plan 0|1] [--height 600 --width 720 --proposals 1000]
"""
import argparse
import ctypes as C
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

VGG = [(3, 64, 0), (64, 64, 1), (64, 128, 0), (128, 128, 1), (128, 256, 0), (256, 256, 0), (256, 256, 1),
       (256, 512, 0), (512, 512, 0), (512, 512, 1), (512, 512, 0), (512, 512, 0), (512, 512, 0)]
VGG_NAMES = ["conv1_1", "conv1_2", "conv2_1", "conv2_2", "conv3_1", "conv3_2", "conv3_3", "conv4_1", "conv4_2", "conv4_3",
             "conv5_1", "conv5_2", "conv5_3"]
KIND = ["plain", "splitk", "streamk", "tail"]
ROUTE = ["ks", "v2_128x64", "v2_128x128", "v2_64x64"]


def plan(M, N, K, cin=0, amax=0, serial=0):
    from densecap_amd._lib import lib
    o = (C.c_int32 * 8)()
    rc = lib().dc_debug_plan_gemm(M, N, K, 0, cin, amax, serial, o)
    assert rc == 0, (rc, M, N, K)
    v = list(o)
    return dict(kind=KIND[v[0]], route=ROUTE[v[1]], stages=v[2], splitk=v[3], m_split=v[4])


def family(p, conv, amax):
    """Kernel name(s) as rocprofv3 prints them (namespace and argument list stripped) for a plan."""
    c, a = ("true" if conv else "false"), ("true" if amax else "false")
    if p["kind"] == "streamk":
        return ["mfma_gemm_sk_kernel<%s>" % c] + (["mfma_gemm_ks_kernel<%s>" % c] if p["m_split"] > 0 else [])
    if p["kind"] == "tail":
        return ["mfma_gemm_ks_kernel<%s>" % c] * (2 if p["m_split"] > 0 else 1)
    if p["route"] == "ks":
        return ["mfma_gemm_ks_kernel<%s>" % c]
    if p["route"] == "v2_128x64":
        return ["mfma_gemm_v2_mixed_kernel<%s, %d, %s, 0>" % (c, p["stages"], a)]      # (last argument: BF3 = 0, the fp32 arithmetic)
    if p["route"] == "v2_128x128":
        return ["mfma_gemm_v2_kernel<2, 2, %s, 3, false, 0>" % c]
    return ["mfma_gemm_v2_kernel<1, 1, %s, 3, %s, 0>" % (c, a)]


def launches(H=600, W=720, P=1000, T=15, V=10497, serial=0, k=12, R=256, D=4096, E=512, Hd=512):
    out = []

    def add(layer, M, N, K, cin=0, amax=0, in_bytes=None, out_bytes=None, flops=None):
        p = plan(M, N, K, cin, amax, serial)
        w_bytes = 4.0 * N * K
        a_bytes = in_bytes if in_bytes is not None else 4.0 * M * (cin if cin else K)
        c_bytes = out_bytes if out_bytes is not None else 4.0 * M * N
        out.append(dict(layer=layer, M=M, N=N, K=K, conv=bool(cin), kind=p["kind"], route=p["route"], stages=p["stages"],
                        splitk=p["splitk"], kernels=family(p, bool(cin), bool(amax)),
                        gflop=(flops if flops is not None else 2.0 * M * N * K) / 1e9,
                        algorithmic_bytes=a_bytes + w_bytes + c_bytes))
    h, w = H, W
    for i, (cin, cout, pool) in enumerate(VGG):
        oh, ow = ((h + 1) // 2, (w + 1) // 2) if pool else (h, w)
        if i > 0:
            M = 4 * oh * ow if pool else h * w                   # pooled convs walk window slots (4 per pooled pixel)
            add(VGG_NAMES[i] + ("+pool" if pool else ""), M, cout, 9 * cin, cin=cin, in_bytes=4.0 * h * w * cin,
                out_bytes=4.0 * oh * ow * cout, flops=2.0 * h * w * cout * 9 * cin)
        h, w = oh, ow
    add("rpn_conv", h * w, R, 9 * 512, cin=512)
    add("rpn_heads", h * w, 6 * k, R)
    add("fc6", P, D, 49 * 512)
    add("fc7", P, D, D)
    add("lm_encoder", P, E, D)
    add("image_step_gates", P, 4 * Hd, E)
    add("h0.Wh", P, 4 * Hd, Hd)
    v1pad = (V + 1 + 63) // 64 * 64
    for t in range(1, T):
        add("decode_step_%d" % t, P, v1pad + 4 * Hd, Hd, amax=1, out_bytes=4.0 * (P * 4 * Hd + 2 * P * (v1pad // 32)),
            flops=2.0 * P * (V + 1 + 4 * Hd) * Hd)
    add("decode_last_argmax", P, V + 1, Hd, amax=1, out_bytes=4.0 * 2 * P * (v1pad // 32))
    return out


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--serial-plan", type=int, default=0)
    ap.add_argument("--height", type=int, default=600)
    ap.add_argument("--width", type=int, default=720)
    ap.add_argument("--proposals", type=int, default=1000)
    ap.add_argument("--json", action="store_true")
    a = ap.parse_args()
    L = launches(a.height, a.width, a.proposals, serial=a.serial_plan)
    if a.json:
        print(json.dumps(L))
    else:
        print("%-20s %9s %6s %6s %-8s %-11s %8s %10s  %s" % ("layer", "M", "N", "K", "kind", "route", "GFLOP", "alg MB", "kernel"))
        for l in L:
            print("%-20s %9d %6d %6d %-8s %-11s %8.2f %10.1f  %s" % (l["layer"], l["M"], l["N"], l["K"], l["kind"], l["route"],
                                                                    l["gflop"], l["algorithmic_bytes"] / 1e6, " + ".join(l["kernels"])))
        print("total: %d launches, %.1f GFLOP, %.1f MB algorithmic" % (len(L), sum(l["gflop"] for l in L),
                                                                        sum(l["algorithmic_bytes"] for l in L) / 1e6))
