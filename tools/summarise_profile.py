"""Condense rocprofv3 output directories (gpurun_out/prof/*) into the small summaries
committed under profiles/.  Usage: python tools/summarise_profile.py <prof_dir> <tag>"""
import collections
import csv
import glob
import json
import os
import sys

prof, tag = sys.argv[1], sys.argv[2]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out_dir = os.path.join(root, "profiles")
os.makedirs(out_dir, exist_ok=True)


def counters(subdir, match):
    files = glob.glob(os.path.join(prof, subdir, "**", "*_counter_collection.csv"), recursive=True)
    acc = collections.defaultdict(list)
    dur = collections.defaultdict(list)
    for f in files:
        for r in csv.DictReader(open(f)):
            if match in r["Kernel_Name"]:
                acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
                dur[r["Counter_Name"]].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    return {k: {"launches": len(v), "mean": sum(v) / len(v), "mean_ns": sum(dur[k]) / len(v)}
            for k, v in acc.items()}


summary = {"tag": tag}
# 1. kernel stats of bench.py
for f in glob.glob(os.path.join(prof, "bench_trace", "**", "*_kernel_stats.csv"), recursive=True):
    rows = list(csv.DictReader(open(f)))
    with open(os.path.join(out_dir, "%s_bench_kernel_stats.csv" % tag), "w") as o:
        w = csv.DictWriter(o, fieldnames=rows[0].keys())
        w.writeheader()
        w.writerows(rows)
    summary["bench_kernel_stats"] = [
        {"name": r["Name"][:90], "calls": int(r["Calls"]), "avg_ns": float(r["AverageNs"]),
         "pct": float(r["Percentage"])} for r in rows[:8]]
# 2. calibration
cal = {}
for name, width in [("k_stream<unsigned int>", 4), ("k_stream<HIP_vector_type<unsigned int, 2u>", 8),
                    ("k_stream<HIP_vector_type<unsigned int, 4u>", 16)]:
    c = counters("pmc_calib", name)
    if "FETCH_SIZE" in c:
        cal[str(width)] = c["FETCH_SIZE"]["mean"] * 1024 / (2 << 30)
summary["fetch_size_fraction_of_true_bytes_by_lane_width"] = cal
# 3. sweep kernel PMC
pmc = {}
for sub in ["pmc_fetch", "pmc_l2", "pmc_sq1", "pmc_sq2"]:
    pmc.update(counters(sub, "k_sa_sweep"))
summary["sweep_pmc"] = pmc
if "FETCH_SIZE" in pmc:
    # loads are 1/3 dword + 2/3 dwordx2 by bytes; correct each with its calibration factor
    f4, f8 = cal.get("4", 1.0), cal.get("8", 1.0)
    corr = 1.0 / ((1 / 3) * f4 + (2 / 3) * f8) if cal else 1.0
    fetch = pmc["FETCH_SIZE"]["mean"] * 1024 * corr
    write = pmc.get("WRITE_SIZE", {"mean": 0})["mean"] * 1024
    summary["sweep_hbm_bytes_per_launch"] = {"fetch_corrected": fetch, "write": write,
                                             "correction": corr, "total": fetch + write}
    with open(os.path.join(out_dir, "traffic.json"), "w") as o:
        json.dump({"tag": tag, "kernel": "k_sa_sweep", "command": "bench.py (default workload)",
                   "hbm_bytes_per_launch": fetch + write, "fetch_bytes_corrected": fetch,
                   "write_bytes": write, "fetch_correction": corr,
                   "launches_averaged": pmc["FETCH_SIZE"]["launches"],
                   "method": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE in separate passes; "
                             "FETCH_SIZE (KB) doubled per the gfx950 calibration in this file's "
                             "sibling summary (tools/fetch_calibrate.hip)"}, o, indent=1)
    hit, miss = pmc.get("TCC_HIT_sum"), pmc.get("TCC_MISS_sum")
    if hit and miss:
        summary["sweep_l2_hit_rate"] = hit["mean"] / (hit["mean"] + miss["mean"])
# 4. the VALU-issue side of the roofline: instructions per flip, measured clock
pmc.update(counters("pmc_clock", "k_sa_sweep"))
if "SQ_INSTS_VALU" in pmc:
    sys.path.insert(0, root)
    import bench  # the workload's shape: sizes, chains and sweeps per launch

    flips_per_launch = sum(bench.CLUSTER_SIZES) / len(bench.CLUSTER_SIZES) * 1024 * 128
    clock = None
    if "GRBM_GUI_ACTIVE" in pmc:
        clock = pmc["GRBM_GUI_ACTIVE"]["mean"] / 8.0 / pmc["GRBM_GUI_ACTIVE"]["mean_ns"]
    with open(os.path.join(out_dir, "sweep_counters.json"), "w") as o:
        json.dump({
            "tag": tag, "kernel": "k_sa_sweep", "command": "bench.py (default workload)",
            "valu_insts_per_launch": pmc["SQ_INSTS_VALU"]["mean"],
            "flips_per_launch": flips_per_launch,
            "valu_insts_per_flip": pmc["SQ_INSTS_VALU"]["mean"] / flips_per_launch,
            "active_valu_quad_cycles_per_launch": pmc.get("SQ_ACTIVE_INST_VALU", {}).get("mean"),
            "cycles_per_valu_inst": 4.35,
            "cycles_per_valu_inst_source": "profiles/%s_issue_rate_probe.txt: 4.2-4.4 SIMD cycles per "
                                           "wave64 instruction for every class the kernel's hot "
                                           "phases issue at 3-4 waves per SIMD (f64 FMA/add, VOP3 "
                                           "integer, SDWA, left shifts, 32x32 multiplies); plain VOP2 "
                                           "and/or/xor/add/right-shift 2.5; v_exp_f32 8.3" % tag,
            "clock_ghz": clock,
            "method": "rocprofv3 --pmc SQ_INSTS_VALU (per launch, mean over the launches of the three "
                      "cluster sizes) / flip attempts per launch; clock = GRBM_GUI_ACTIVE / 8 / kernel "
                      "time (separate pass)"}, o, indent=1)
    summary["sweep_valu"] = {"insts_per_flip": pmc["SQ_INSTS_VALU"]["mean"] / flips_per_launch,
                             "clock_ghz": clock}
# 5. build-side and batched kernels: kernel-trace stats copied as they are
for sub, name in [("build_trace", "build_kernel_stats"), ("batch_trace", "batch_kernel_stats")]:
    for f in glob.glob(os.path.join(prof, sub, "**", "*_kernel_stats.csv"), recursive=True):
        rows = list(csv.DictReader(open(f)))
        if not rows:
            continue
        with open(os.path.join(out_dir, "%s_%s.csv" % (tag, name)), "w") as o:
            w = csv.DictWriter(o, fieldnames=rows[0].keys())
            w.writeheader()
            w.writerows(rows)
with open(os.path.join(out_dir, "%s_summary.json" % tag), "w") as o:
    json.dump(summary, o, indent=1)
print(json.dumps(summary, indent=1))
