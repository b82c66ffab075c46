"""Summarise rocprofv3 CSV output (kernel stats + separate --pmc passes) into profiles/.

usage: python tools/pmc_summary.py <rocprof_dir> <out_prefix>
Expects in <rocprof_dir>: lanes1_kernel_stats.csv, lanes1_kernel_trace.csv, bench_lanes1.json and the
counter passes fetch_*, write_*, mfma_* produced by (each its own run, --kernel-trace + --pmc only):
  rocprofv3 --kernel-trace --stats ...                      -- python bench.py --lanes 1 ...
  rocprofv3 --kernel-trace --pmc FETCH_SIZE ...             -- python bench.py --lanes 1 ...
  rocprofv3 --kernel-trace --pmc WRITE_SIZE ...             -- python bench.py --lanes 1 ...
  rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_LDS_BANK_CONFLICT ...
HBM bytes follow MI355X_MICROARCH.md: FETCH_SIZE/WRITE_SIZE are KiB; on gfx950 FETCH_SIZE counts 128-B
requests as 64 B for wide coalesced reads, so the read side is doubled ("fetch_kb_x2"); WRITE_SIZE matched a
known byte count (RoI pool output, 98,000 KiB) and is used as is.
"""
import collections
import csv
import json
import os
import shutil
import sys


def fam(name):
    n = name.replace("(anonymous namespace)::", "").replace("void ", "")
    n = n.split("(")[0]
    return n


def layer_table(d, tag, serial_plan):
    """Per-LAYER figures of a one-stream pass: the MFMA launches of an image leave the library in a fixed order
    (tools/launch_list.py); the MFMA-family dispatches of the trace are cut into images and laid over that list.  Every
    position's kernel name is checked against the planner's prediction.  tag: file prefix of the passes (<tag>_kernel_trace,
    <tag>fetch_, <tag>write_, <tag>mfma_ counter passes)."""
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import launch_list
    L = launch_list.launches(serial=serial_plan)
    flat = [(li, k) for li, l in enumerate(L) for k in l["kernels"]]

    def mfma_rows(name):
        rows = [r for r in csv.DictReader(open(os.path.join(d, name + "_kernel_trace.csv"))) if fam(r["Kernel_Name"]).startswith("mfma_gemm")]
        rows.sort(key=lambda r: int(r["Dispatch_Id"]))
        return rows[1:]                                      # the first contraction of a process is dc_load_weights' xg table
    def lay(rows):
        n = len(flat)
        if len(rows) < n or len(rows) % n:
            return None, "%d MFMA dispatches are not a multiple of the %d per image" % (len(rows), n)
        for i, r in enumerate(rows):
            if fam(r["Kernel_Name"]) != flat[i % n][1]:
                return None, "dispatch %d is %s, the plan says %s (%s)" % (i, fam(r["Kernel_Name"]), flat[i % n][1], L[flat[i % n][0]]["layer"])
        return [flat[i % n][0] for i in range(len(rows))], None
    base = tag if tag else "lanes1"
    rows = mfma_rows(base)
    idx, err = lay(rows)
    if err:
        return dict(_error=err)
    tab = [collections.OrderedDict(layer=l["layer"], kernel=" + ".join(l["kernels"]), kind=l["kind"], gflop=l["gflop"],
                                   algorithmic_bytes=l["algorithmic_bytes"], us=0.0, n=0) for l in L]
    for r, li in zip(rows, idx):
        tab[li]["us"] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    nimg = len(rows) // len(flat)
    for t in tab:
        t["us"] /= nimg; t["n"] = nimg
        t["tflops"] = t["gflop"] / t["us"] * 1e3 if t["us"] else 0.0
    pre = (tag + "_") if tag else ""
    for cname, key in (("fetch", "FETCH_SIZE"), ("write", "WRITE_SIZE")):
        if not os.path.exists(os.path.join(d, pre + cname + "_counter_collection.csv")):
            continue
        crow = mfma_rows(pre + cname)
        cidx, err = lay(crow)
        if err:
            continue
        vals = {}
        for r in csv.DictReader(open(os.path.join(d, pre + cname + "_counter_collection.csv"))):
            if r["Counter_Name"] == key:
                vals[r["Dispatch_Id"]] = vals.get(r["Dispatch_Id"], 0.0) + float(r["Counter_Value"])
        acc = [0.0] * len(L)
        for r, li in zip(crow, cidx):
            acc[li] += vals.get(r["Dispatch_Id"], 0.0)
        for t, a in zip(tab, acc):
            t[cname + "_kb"] = a / (len(crow) // len(flat))
    if os.path.exists(os.path.join(d, pre + "mfma_counter_collection.csv")):
        crow = mfma_rows(pre + "mfma")
        cidx, err = lay(crow)
        if not err:
            per = collections.defaultdict(lambda: collections.defaultdict(float))
            for r in csv.DictReader(open(os.path.join(d, pre + "mfma_counter_collection.csv"))):
                per[r["Dispatch_Id"]][r["Counter_Name"]] += float(r["Counter_Value"])
            acc = [collections.defaultdict(float) for _ in L]
            for r, li in zip(crow, cidx):
                for k, v in per[r["Dispatch_Id"]].items():
                    acc[li][k] += v
                acc[li]["_dur_us"] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
            for t, c in zip(tab, acc):
                if c.get("GRBM_GUI_ACTIVE", 0) > 0:
                    gui = c["GRBM_GUI_ACTIVE"] / 8.0
                    t["mfma_util"] = c["SQ_VALU_MFMA_BUSY_CYCLES"] / (gui * 1024.0)
                    t["clock_ghz"] = gui / c["_dur_us"] / 1e3
    for t in tab:
        if "fetch_kb" in t:
            t["fabric_bytes"] = (2.0 * t["fetch_kb"] + t.get("write_kb", 0.0)) * 1024      # FETCH_SIZE x2: MI355X_MICROARCH.md (wide coalesced reads)
            t["fabric_over_algorithmic"] = t["fabric_bytes"] / t["algorithmic_bytes"]
    return tab


def main():
    d, prefix = sys.argv[1], sys.argv[2]
    os.makedirs(os.path.dirname(prefix) or ".", exist_ok=True)
    shutil.copy(os.path.join(d, "lanes1_kernel_stats.csv"), prefix + "_kernel_stats_lanes1.csv")
    if os.path.exists(os.path.join(d, "bench_lanes1.json")):
        shutil.copy(os.path.join(d, "bench_lanes1.json"), prefix + "_bench_lanes1_under_rocprof.json")
    summ = collections.OrderedDict()

    def trace(name):
        return {r["Dispatch_Id"]: r for r in csv.DictReader(open(os.path.join(d, name + "_kernel_trace.csv")))}

    # durations per family from the stats trace
    for r in csv.DictReader(open(os.path.join(d, "lanes1_kernel_trace.csv"))):
        f = fam(r["Kernel_Name"])
        e = summ.setdefault(f, collections.OrderedDict(calls=0, total_us=0.0))
        e["calls"] += 1
        e["total_us"] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    for f, e in summ.items():
        e["avg_us"] = e["total_us"] / e["calls"]
    for cname, key in (("fetch", "FETCH_SIZE"), ("write", "WRITE_SIZE")):
        acc = collections.defaultdict(lambda: [0.0, 0])
        for r in csv.DictReader(open(os.path.join(d, cname + "_counter_collection.csv"))):
            if r["Counter_Name"] != key:
                continue
            a = acc[fam(r["Kernel_Name"])]
            a[0] += float(r["Counter_Value"]); a[1] += 1
        for f, (tot, n) in acc.items():
            if f in summ:
                summ[f]["avg_%s_kb" % cname] = tot / n
    for f, e in summ.items():
        if "avg_fetch_kb" in e:
            e["avg_hbm_bytes_per_launch"] = (2.0 * e["avg_fetch_kb"] + e.get("avg_write_kb", 0.0)) * 1024
    # MFMA utilisation / clock
    kt = trace("mfma")
    acc = collections.defaultdict(lambda: collections.defaultdict(float))
    for r in csv.DictReader(open(os.path.join(d, "mfma_counter_collection.csv"))):
        f = fam(r["Kernel_Name"])
        acc[f][r["Counter_Name"]] += float(r["Counter_Value"])
        if r["Counter_Name"] == "GRBM_GUI_ACTIVE":
            k = kt[r["Dispatch_Id"]]
            acc[f]["_dur_us"] += (int(k["End_Timestamp"]) - int(k["Start_Timestamp"])) / 1e3
    for f, c in acc.items():
        if f in summ and c.get("GRBM_GUI_ACTIVE", 0) > 0 and c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) > 0:
            gui = c["GRBM_GUI_ACTIVE"] / 8.0          # summed over the 8 XCDs
            summ[f]["mfma_util"] = c["SQ_VALU_MFMA_BUSY_CYCLES"] / (gui * 1024.0)   # 1024 SIMDs
            summ[f]["clock_ghz"] = gui / c["_dur_us"] / 1e3
            summ[f]["mfma_tflops_counted"] = c["SQ_INSTS_VALU_MFMA_MOPS_F32"] * 512 / (c["_dur_us"] * 1e-6) / 1e12
            summ[f]["lds_bank_conflict_cycles"] = c.get("SQ_LDS_BANK_CONFLICT", 0.0)
    try:
        import subprocess
        summ["_commit"] = subprocess.check_output(["git", "rev-parse", "--short", "HEAD"]).decode().strip()
    except Exception:
        summ["_commit"] = None
    # per-layer tables: the serial (single-image) planning of the lanes1 passes, and -- when the passes exist -- the
    # multi-lane planning on one stream (bench.py --lanes 1 --plan-mode 0: the kernels of the TIMED schedule)
    summ["_layers_single_image_plan"] = layer_table(d, "", 1)
    if os.path.exists(os.path.join(d, "mlplan_kernel_trace.csv")):
        summ["_layers_multi_lane_plan"] = layer_table(d, "mlplan", 0)
    # algorithmic bytes per kernel family (operands once + result once), next to the fabric bytes counted above
    lt = summ["_layers_single_image_plan"]
    if isinstance(lt, list):
        famb = collections.defaultdict(lambda: [0.0, 0])
        for t in lt:
            ks = t["kernel"].split(" + ")
            for k in ks:
                famb[k][0] += t["algorithmic_bytes"] / len(ks); famb[k][1] += 1
        for f, (tot, n) in famb.items():
            if f in summ:
                summ[f]["algorithmic_bytes_per_launch"] = tot / n
                if "avg_hbm_bytes_per_launch" in summ[f]:
                    summ[f]["fabric_over_algorithmic"] = summ[f]["avg_hbm_bytes_per_launch"] / (tot / n)
    # the opt-in split-bf16 mode (round 5): its own passes (split_*), per kernel family -- names, durations, MFMA busy, clock
    if os.path.exists(os.path.join(d, "split_kernel_trace.csv")):
        sp = collections.OrderedDict()
        for r in csv.DictReader(open(os.path.join(d, "split_kernel_trace.csv"))):
            e = sp.setdefault(fam(r["Kernel_Name"]), collections.OrderedDict(calls=0, total_us=0.0))
            e["calls"] += 1
            e["total_us"] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
        for e in sp.values():
            e["avg_us"] = e["total_us"] / e["calls"]
        if os.path.exists(os.path.join(d, "split_mfma_counter_collection.csv")):
            kt2 = trace("split_mfma")
            acc2 = collections.defaultdict(lambda: collections.defaultdict(float))
            for r in csv.DictReader(open(os.path.join(d, "split_mfma_counter_collection.csv"))):
                f = fam(r["Kernel_Name"])
                acc2[f][r["Counter_Name"]] += float(r["Counter_Value"])
                if r["Counter_Name"] == "GRBM_GUI_ACTIVE":
                    k = kt2[r["Dispatch_Id"]]
                    acc2[f]["_dur_us"] += (int(k["End_Timestamp"]) - int(k["Start_Timestamp"])) / 1e3
            for f, c in acc2.items():
                if f in sp and c.get("GRBM_GUI_ACTIVE", 0) > 0 and c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) > 0:
                    gui = c["GRBM_GUI_ACTIVE"] / 8.0
                    sp[f]["mfma_util"] = c["SQ_VALU_MFMA_BUSY_CYCLES"] / (gui * 1024.0)
                    sp[f]["clock_ghz"] = gui / c["_dur_us"] / 1e3
                    sp[f]["valu_insts"] = c.get("SQ_INSTS_VALU", 0.0)
                    sp[f]["lds_bank_conflict_cycles"] = c.get("SQ_LDS_BANK_CONFLICT", 0.0)
        for cname, key in (("fetch", "FETCH_SIZE"), ("write", "WRITE_SIZE")):
            fn = os.path.join(d, "split_%s_counter_collection.csv" % cname)
            if not os.path.exists(fn):
                continue
            acc3 = collections.defaultdict(lambda: [0.0, 0])
            for r in csv.DictReader(open(fn)):
                if r["Counter_Name"] == key:
                    a3 = acc3[fam(r["Kernel_Name"])]
                    a3[0] += float(r["Counter_Value"]); a3[1] += 1
            for f, (tot, n) in acc3.items():
                if f in sp:
                    sp[f]["avg_%s_kb" % cname] = tot / n
        summ["_split_bf16_mode"] = sp
        print("_split_bf16_mode (bench.py --lanes 1 --plan-mode 0 --math-mode 1)")
        for f, e in sp.items():
            if e["total_us"] > 50:
                print("  %-52s calls=%4d avg_us=%9.1f %s" % (f[:52], e["calls"], e["avg_us"],
                      " ".join("%s=%.3g" % (k, v) for k, v in e.items() if k not in ("calls", "total_us", "avg_us"))))
    # round 6: the captions-after-the-final-NMS schedule (bench.py --lanes 1 --plan-mode 0 --group 8 --caption-order 1): the
    # packed decode launches of a group of four are in this trace; durations per family + MFMA busy of its kernels
    if os.path.exists(os.path.join(d, "capnms_kernel_trace.csv")):
        cp = collections.OrderedDict()
        for r in csv.DictReader(open(os.path.join(d, "capnms_kernel_trace.csv"))):
            f = fam(r["Kernel_Name"])
            e = cp.setdefault(f, collections.OrderedDict(calls=0, total_us=0.0))
            e["calls"] += 1
            e["total_us"] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
        for f, e in cp.items():
            e["avg_us"] = e["total_us"] / e["calls"]
        fn = os.path.join(d, "capnms_mfma_counter_collection.csv")
        if os.path.exists(fn):
            tr = trace("capnms_mfma")
            acc2 = collections.defaultdict(lambda: collections.defaultdict(float))
            seen = collections.defaultdict(set)
            for r in csv.DictReader(open(fn)):
                f = fam(r["Kernel_Name"])
                acc2[f][r["Counter_Name"]] += float(r["Counter_Value"])
                if r["Dispatch_Id"] not in seen[f] and r["Dispatch_Id"] in tr:
                    seen[f].add(r["Dispatch_Id"])
                    t = tr[r["Dispatch_Id"]]
                    acc2[f]["_dur_us"] += (int(t["End_Timestamp"]) - int(t["Start_Timestamp"])) / 1e3
            for f, c in acc2.items():
                if f in cp and c.get("GRBM_GUI_ACTIVE", 0) > 0 and c.get("_dur_us", 0) > 0 and f.startswith("mfma_gemm"):
                    gui = c["GRBM_GUI_ACTIVE"] / 8.0
                    cp[f]["mfma_util"] = c["SQ_VALU_MFMA_BUSY_CYCLES"] / (gui * 1024.0)
                    cp[f]["clock_ghz"] = gui / c["_dur_us"] / 1e3
        summ["_captions_after_final_nms_group8"] = cp
        print("_captions_after_final_nms_group8 (bench.py --lanes 1 --plan-mode 0 --group 8 --caption-order 1)")
        for f, e in cp.items():
            if e["total_us"] > 50:
                print("  %-52s calls=%4d avg_us=%9.1f %s" % (f[:52], e["calls"], e["avg_us"],
                      " ".join("%s=%.3g" % (k, v) for k, v in e.items() if k not in ("calls", "total_us", "avg_us"))))
    json.dump(summ, open(prefix + "_pmc_summary.json", "w"), indent=1)
    for key in ("_layers_single_image_plan", "_layers_multi_lane_plan"):
        tab = summ.get(key)
        if isinstance(tab, list):
            print(key)
            for t in tab:
                print("  %-20s %-42s %8.1f us %6.1f TF  mfma %.2f @ %.2f GHz  fabric %7.1f MB / alg %6.1f MB = %.2f" % (
                    t["layer"], t["kernel"][:42], t["us"], t["tflops"], t.get("mfma_util", 0), t.get("clock_ghz", 0),
                    t.get("fabric_bytes", 0) / 1e6, t["algorithmic_bytes"] / 1e6, t.get("fabric_over_algorithmic", 0)))
        elif tab:
            print(key, tab)
    for extra in ("mlplan_kernel_stats.csv", "bench_mlplan.json", "default_kernel_stats.csv", "bench_default_under_rocprof.json", "bench_default.json",
                  "bench_webcam_480_p50.json", "bench_config3_p300.json", "bench_config5.json", "bench_config0_720x480.json",
                  "gemm_bench_serial.txt", "gemm_bench_multilane.txt", "decode_bench.txt", "parity_report.json",
                  "split_kernel_stats.csv", "bench_split_lanes1.json", "bench_split_bf16_mode.json", "gemm_bench_split_bf16.txt",
                  "cli_throughput.json", "mfma_bf16_numerics.txt", "split_bf16_error_ratios.txt",
                  "capnms_kernel_stats.csv", "bench_capnms_under_rocprof.json", "latency_check.txt", "decode_trace_small_rows.txt",
                  "survivor_decode_trace.txt", "nms_chain_trace.txt", "fuzz_nms.txt", "webcam_layers.txt"):
        if os.path.exists(os.path.join(d, extra)):
            shutil.copy(os.path.join(d, extra), prefix + "_" + extra.replace("default_kernel_stats", "kernel_stats_default_lanes"))
    for f, e in summ.items():
        if isinstance(e, dict) and "total_us" in e and e["total_us"] > 50:
            print("%-44s calls=%4d avg_us=%9.1f %s" % (f[:44], e["calls"], e["avg_us"],
                  " ".join("%s=%.3g" % (k, v) for k, v in e.items() if k not in ("calls", "total_us", "avg_us"))))


if __name__ == "__main__":
    main()
