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
    json.dump(summ, open(prefix + "_pmc_summary.json", "w"), indent=1)
    for extra in ("default_kernel_stats.csv", "bench_default_under_rocprof.json", "bench_default.json",
                  "bench_webcam_480_p50.json", "bench_config3_p300.json", "bench_config5.json", "bench_config0_720x480.json",
                  "gemm_bench_serial.txt", "gemm_bench_multilane.txt", "decode_bench.txt", "parity_report.json"):
        if os.path.exists(os.path.join(d, extra)):
            shutil.copy(os.path.join(d, extra), prefix + "_" + extra.replace("default_kernel_stats", "kernel_stats_default_lanes"))
    for f, e in summ.items():
        if isinstance(e, dict) and e["total_us"] > 50:
            print("%-44s calls=%4d avg_us=%9.1f %s" % (f[:44], e["calls"], e["avg_us"],
                  " ".join("%s=%.3g" % (k, v) for k, v in e.items() if k not in ("calls", "total_us", "avg_us"))))


if __name__ == "__main__":
    main()
