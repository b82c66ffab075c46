"""Per-layer times of a one-stream pass at ANY workload size: lays the MFMA dispatches of a rocprofv3 kernel trace over
tools/launch_list.py's launch order (as tools/pmc_summary.py does for the BASELINE size) and lists the other kernels by name.
(Name mismatches are reported, not fatal: the planner query always offers a split-K workspace, the image-step gate products
of the decode run without one -- at 50 proposals the query says K-split + split-K where the launch is a plain 64x64 one.  The
launch COUNT per layer is the same, so the times still land on the right rows.)

usage: python tools/layer_times.py <kernel_trace.csv> [--height 600 --width 720 --proposals 1000 --serial-plan 1]"""
import argparse, collections, csv, os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import launch_list
from pmc_summary import fam

ap = argparse.ArgumentParser()
ap.add_argument("trace")
ap.add_argument("--height", type=int, default=600)
ap.add_argument("--width", type=int, default=720)
ap.add_argument("--proposals", type=int, default=1000)
ap.add_argument("--serial-plan", type=int, default=1)
a = ap.parse_args()
L = launch_list.launches(H=a.height, W=a.width, P=a.proposals, serial=a.serial_plan)
flat = [(li, k) for li, l in enumerate(L) for k in l["kernels"]]
rows = sorted(csv.DictReader(open(a.trace)), key=lambda r: int(r["Dispatch_Id"]))
mf = [r for r in rows if fam(r["Kernel_Name"]).startswith("mfma_gemm")][1:]
n = len(flat)
nimg = len(mf) // n
mf = mf[len(mf) - nimg * n:]                       # the LAST whole images (warm)
us = [0.0] * len(L)
bad = 0
for i, r in enumerate(mf):
    if fam(r["Kernel_Name"]) != flat[i % n][1]:
        bad += 1
    us[flat[i % n][0]] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
print("%d images, %d MFMA launches each, %d name mismatches against the plan" % (nimg, n, bad))
tot = 0.0
for l, u in zip(L, us):
    u /= max(nimg, 1); tot += u
    print("%-18s %-34s %-8s %8.2f GF %8.1f us %7.1f TF" % (l["layer"], " + ".join(l["kernels"])[:34], l["kind"], l["gflop"], u, l["gflop"] / u * 1e3 if u else 0))
print("MFMA launches: %.1f us per image" % tot)
other = collections.defaultdict(lambda: [0, 0.0])
for r in rows:
    f = fam(r["Kernel_Name"])
    if not f.startswith("mfma_gemm"):
        other[f][0] += 1; other[f][1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
allimg = max(len([r for r in rows if fam(r["Kernel_Name"]).startswith("mfma_gemm")]) // n, 1)
for f, (c, u) in sorted(other.items(), key=lambda kv: -kv[1][1]):
    print("  %-44s %6.1f launches/image %8.1f us/image" % (f[:44], c / allimg, u / allimg))
