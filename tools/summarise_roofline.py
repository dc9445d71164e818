"""Condense the rocprofv3 passes of tools/gpu_profile_r4.sh into profiles/sweep_counters.json and
profiles/traffic.json (what bench.py's `roofline` objects are computed from) and copy the kernel
stats.  Usage: python tools/summarise_roofline.py <prof_dir> <tag>

Layout of <prof_dir>: cases/<case>.json (tools/profile_roofline.py), <case>/<set>/**/*_counter_collection.csv
for the counter sets `sq` (instruction counts, busy cycles, GRBM_GUI_ACTIVE), `lanes`
(SQ_THREAD_CYCLES_VALU over SQ_ACTIVE_INST_VALU: lanes with exec = 1 per VALU instruction), `fetch`
(FETCH_SIZE) and `l2` (WRITE_SIZE, TCC hits / misses), and calib/ for tools/fetch_calibrate.hip.

Every case carries the fingerprint of the sources its kernel is built from
(annealing_sign_problem_amd/build.py: KERNEL_SOURCE_SETS) as it was when the case was profiled;
bench.py refuses to compute a fraction from counters of other code.  Cases that are NOT under
<prof_dir> are carried over from the committed files as they are (with the fingerprints they were
measured on): a change to sa_shuffled.hip needs the shuffled cases re-profiled, not all twelve.
"""
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


def counters(directory, match):
    """Sum over all dispatches of kernels whose name contains `match`: {counter: total}, the
    number of dispatches and their total duration in ns (per counter set).  Reads the per-kernel
    sums of tools/reduce_counters.py (*_counter_collection.json) or rocprofv3's own CSVs."""
    total = collections.defaultdict(float)
    dispatches, ns = 0, 0.0
    for f in glob.glob(os.path.join(directory, "**", "*_counter_collection.json"), recursive=True):
        for name, k in json.load(open(f)).items():
            if match in name:
                for counter, value in k["counters"].items():
                    total[counter] += value
                dispatches += k["dispatches"]
                ns += k["ns"]
    seen = {}
    for f in glob.glob(os.path.join(directory, "**", "*_counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if match in r["Kernel_Name"]:
                total[r["Counter_Name"]] += float(r["Counter_Value"])
                # (per file: two runs of one command number their dispatches alike)
                seen[(f, r["Dispatch_Id"])] = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    return dict(total), dispatches + len(seen), ns + float(sum(seen.values()))


# FETCH_SIZE calibration on this chip: fraction of the true bytes the counter reports
calibration = {}
for name, width in [("k_stream<unsigned int>", 4), ("k_stream<HIP_vector_type<unsigned int, 2u>", 8),
                    ("k_stream<HIP_vector_type<unsigned int, 4u>", 16)]:
    c, n, _ = counters(os.path.join(prof, "calib"), name)
    if n and "FETCH_SIZE" in c:
        calibration[str(width)] = c["FETCH_SIZE"] / n * 1024 / (2 << 30)
# the sweep kernels fetch 16 bytes per lane and instruction
fetch_factor = 1.0 / calibration["16"] if "16" in calibration else 2.0

fingerprints = set()
cases = {}
for path in sorted(glob.glob(os.path.join(prof, "cases", "*.json"))):
    rec = json.load(open(path))
    case = rec["case"]
    fingerprints.add(rec.get("library_fingerprint"))
    flips = float(rec["flips"])
    entry = {"kernel": rec["kernel"], "flips_profiled": flips,
             "kernel_source_set": rec.get("kernel_source_set"),
             "source_set_fingerprint": rec.get("source_set_fingerprint"),
             "library_fingerprint": rec.get("library_fingerprint"), "tag": tag}
    for key in ("K", "dbar", "chains", "sweeps", "problems"):
        if key in rec:
            entry[key] = rec[key]
    sq, launches, ns = counters(os.path.join(prof, case, "sq"), rec["kernel"])
    if launches:
        entry["launches"] = launches
        entry["kernel_ms_per_launch"] = ns / launches * 1e-6
        entry["kernel_flips_per_s_under_profiler"] = flips / (ns * 1e-9)
        for name, key in [("SQ_INSTS_VALU", "valu_insts_per_flip"), ("SQ_INSTS_SALU", "salu_insts_per_flip"),
                          ("SQ_INSTS_LDS", "lds_insts_per_flip"), ("SQ_INSTS_VMEM_RD", "vmem_rd_insts_per_flip"),
                          ("SQ_ACTIVE_INST_VALU", "active_valu_quad_cycles_per_flip")]:
            if name in sq:
                entry[key] = sq[name] / flips
        if "GRBM_GUI_ACTIVE" in sq and ns:
            entry["clock_ghz"] = sq["GRBM_GUI_ACTIVE"] / 8.0 / ns
        if "SQ_WAVE_CYCLES" in sq and "SQ_BUSY_CYCLES" in sq and sq["SQ_BUSY_CYCLES"]:
            entry["mean_waves_per_busy_cycle"] = sq["SQ_WAVE_CYCLES"] / sq["SQ_BUSY_CYCLES"]
    lanes, n_lanes, _ = counters(os.path.join(prof, case, "lanes"), rec["kernel"])
    if n_lanes and lanes.get("SQ_ACTIVE_INST_VALU"):
        # lanes with exec = 1 per VALU instruction (VALUThreadUtilization of the counter definitions)
        entry["exec_lane_utilisation"] = lanes["SQ_THREAD_CYCLES_VALU"] / (lanes["SQ_ACTIVE_INST_VALU"] * 64.0)
    fetch, n_fetch, _ = counters(os.path.join(prof, case, "fetch"), rec["kernel"])
    l2, n_l2, _ = counters(os.path.join(prof, case, "l2"), rec["kernel"])
    if n_fetch and "FETCH_SIZE" in fetch:
        entry["hbm_fetch_bytes_per_flip"] = fetch["FETCH_SIZE"] * 1024 * fetch_factor / flips
    if n_l2 and "WRITE_SIZE" in l2:
        entry["hbm_write_bytes_per_flip"] = l2["WRITE_SIZE"] * 1024 / flips
        if l2.get("TCC_HIT_sum") is not None and l2.get("TCC_MISS_sum") is not None:
            entry["l2_hit_rate"] = l2["TCC_HIT_sum"] / max(1.0, l2["TCC_HIT_sum"] + l2["TCC_MISS_sum"])
    if "hbm_fetch_bytes_per_flip" in entry and "hbm_write_bytes_per_flip" in entry:
        entry["hbm_bytes_per_flip"] = entry["hbm_fetch_bytes_per_flip"] + entry["hbm_write_bytes_per_flip"]
    # kernels that run beside the case's main kernel (the order kernel of the shuffled sweep)
    for side in rec.get("side_kernels", []):
        s_sq, s_n, s_ns = counters(os.path.join(prof, case, "sq"), side)
        if s_n:
            entry.setdefault("side_kernels", {})[side] = {
                "launches": s_n, "ms_per_launch": s_ns / s_n * 1e-6,
                "valu_insts_per_flip": s_sq.get("SQ_INSTS_VALU", 0.0) / flips}
    cases[case] = entry
    # kernel-trace stats of the case, as rocprofv3 wrote them
    for f in glob.glob(os.path.join(prof, case, "trace", "**", "*_kernel_stats.csv"), recursive=True):
        rows = list(csv.DictReader(open(f)))
        if rows:
            with open(os.path.join(out_dir, "%s_%s_kernel_stats.csv" % (tag, case)), "w") as o:
                w = csv.DictWriter(o, fieldnames=rows[0].keys())
                w.writeheader()
                w.writerows(rows)

if len(fingerprints) != 1 or None in fingerprints:
    raise SystemExit("the cases were not all profiled on one library build: %r" % fingerprints)
fingerprint = fingerprints.pop()
# cases not re-profiled this time: carried over from the committed summary, fingerprints and all
try:
    previous = json.load(open(os.path.join(out_dir, "%s_roofline_summary.json" % tag)))["cases"]
except (OSError, ValueError, KeyError):
    previous = {}
for case, entry in previous.items():
    cases.setdefault(case, entry)
probe = "profiles/r02_issue_rate_probe.txt"
common = {
    "tag": tag,
    "library_fingerprint": fingerprint,  # of the newest cases; every case names its own
    "carried_over_cases": "cases whose library_fingerprint is 47eedac1... were profiled before the per-set "
                          "fingerprints existed: the colour set (sa_sweep.hip, sa_plan.cpp, their headers, the "
                          "flags) is byte-identical at the measured commit 3d5814c and since (git diff empty), so "
                          "they carry the fingerprint any build of those sources stamps",
    "cycles_per_valu_inst": 4.35,
    "cycles_per_valu_inst_source": probe + ": 4.2-4.4 SIMD cycles per wave64 instruction for every class "
                                   "the kernels' hot phases issue at 3-4 waves per SIMD (f64 FMA/add, VOP3 "
                                   "integer, SDWA, left shifts, 32x32 multiplies); plain VOP2 2.5; v_exp_f32 8.3",
    "method": "one rocprofv3 run per case and counter set (tools/gpu_profile_r3.sh): --pmc SQ_INSTS_VALU "
              "SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES "
              "GRBM_GUI_ACTIVE, summed over every dispatch of the case's kernel and divided by the flip "
              "attempts of the case; clock = GRBM_GUI_ACTIVE / 8 / kernel time; exec_lane_utilisation = "
              "SQ_THREAD_CYCLES_VALU / (SQ_ACTIVE_INST_VALU x 64) from a pass of its own (tools/gpu_profile_r4.sh)",
}
counters_out = dict(common)
counters_out["cases"] = {k: {kk: vv for kk, vv in v.items() if not kk.startswith("hbm_") and kk != "l2_hit_rate"}
                         for k, v in cases.items()}
with open(os.path.join(out_dir, "sweep_counters.json"), "w") as o:
    json.dump(counters_out, o, indent=1)
traffic_out = {
    "tag": tag, "library_fingerprint": fingerprint,
    "fetch_size_fraction_of_true_bytes_by_lane_width": calibration, "fetch_correction": fetch_factor,
    "method": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum in separate passes "
              "(MI355X_MICROARCH.md, HBM/rocprofv3 section); FETCH_SIZE (KB) divided by the fraction of the "
              "true bytes it reports for 16-byte-per-lane streams on this chip (tools/fetch_calibrate.hip)",
    "cases": {k: {kk: vv for kk, vv in v.items()
                  if kk.startswith("hbm_") or kk in ("l2_hit_rate", "kernel", "K", "kernel_source_set",
                                                     "source_set_fingerprint", "library_fingerprint", "tag")}
              for k, v in cases.items()},
}
with open(os.path.join(out_dir, "traffic.json"), "w") as o:
    json.dump(traffic_out, o, indent=1)
with open(os.path.join(out_dir, "%s_roofline_summary.json" % tag), "w") as o:
    json.dump({"common": common, "cases": cases, "calibration": calibration}, o, indent=1)
print(json.dumps(cases, indent=1))
