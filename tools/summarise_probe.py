"""rocprofv3 counter file of tools/issue_rate_probe.hip -> profiles/<tag>_issue_rate_probe.txt.
Usage: python tools/summarise_probe.py <rocprof output dir> <tag>

Per instruction class and waves per SIMD: SIMD cycles per wave64 instruction, with the clock of
each dispatch taken from GRBM_GUI_ACTIVE (sum over the 8 XCDs) / 8 / kernel time."""
import collections
import csv
import glob
import os
import re
import sys

prof, tag = sys.argv[1], sys.argv[2]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
NAMES = ["v_fma_f64", "v_add_f64", "v_mul_f64", "v_fma_f32", "v_bfi_b32", "v_lshlrev_b32",
         "v_lshl_or_b32", "v_xor_b32", "v_add_u32", "v_mul_hi_u32", "v_mul_lo_u32",
         "v_or_b32_sdwa", "v_exp_f32", "v_cvt_f64_u32", "v_cvt_f32_f64",
         "mix lshlrev+bfi+fma_f64 (byte layout term)", "mix or_sdwa+fma_f64 (word layout term)",
         "mix mul_hi+mul_lo+2 xor (Philox)",
         "v_mov_b32", "v_and_b32", "v_or_b32", "v_cndmask_b32 (vcc)", "v_cmp_lt_u32 (-> sgpr pair)",
         "v_lshrrev_b32", "v_sub_u32", "v_and_or_b32", "v_or3_b32", "v_add3_u32", "v_bfe_u32",
         "v_perm_b32", "v_alignbit_b32", "v_mad_u64_u32", "v_xor_b32_sdwa (byte select)",
         "v_fmac_f64 (VOP2)", "v_add_f32", "v_mul_f32", "mix xor(hi word) + add_f64",
         "v_xad_u32", "v_lshl_add_u32",
         "v_cndmask_b32_e64 (sgpr-pair mask)", "mix v_cmp_lt_u32 vcc + v_cndmask_b32 vcc",
         "v_lshlrev_b32 by 31", "v_lshlrev_b32 by vgpr", "v_cvt_f32_ubyte1", "v_fmaak_f32",
         "v_mul_u32_u24", "v_readlane_b32 + v_writelane_b32", "v_lshl_add_u64", "v_mov_b64",
         "v_mbcnt_lo + v_mbcnt_hi", "mix lshrrev+lshl_or+fma_f64 (byte layout, new)",
         "v_ashrrev_i32"]
ITERS, UNROLL = 20000, 32
dispatches = collections.OrderedDict()
for f in glob.glob(os.path.join(prof, "**", "*_counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        d = dispatches.setdefault(r["Dispatch_Id"], {})
        d[r["Counter_Name"]] = float(r["Counter_Value"])
        d["kernel"], d["threads"] = r["Kernel_Name"], int(r["Workgroup_Size"])
        d["ns"] = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
table = collections.defaultdict(dict)
for d in dispatches.values():
    if d["ns"] < 5e5:  # warm-up launches
        continue
    kind = int(re.search(r"probe<(\d+)>", d["kernel"]).group(1))
    w = d["threads"] // 256
    cycles = d["GRBM_GUI_ACTIVE"] / 8
    executed = d["SQ_INSTS_VALU"] / (256 * 4 * w)  # per wave, as the hardware counted them
    table[kind][w] = (cycles / (executed * w), d["SQ_ACTIVE_INST_VALU"] / d["SQ_INSTS_VALU"],
                      cycles / d["ns"], executed / (ITERS * UNROLL))
lines = ["# tools/issue_rate_probe.hip under rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE",
         "# SIMD cycles per executed wave64 VALU instruction (every CU busy, one workgroup per CU),",
         "# clock per dispatch = GRBM_GUI_ACTIVE / 8 / kernel time; 'quads/inst' = SQ_ACTIVE_INST_VALU / SQ_INSTS_VALU",
         "%-46s %6s %6s %6s %6s  %10s %9s %10s" % ("instruction", "1 w/S", "2 w/S", "3 w/S", "4 w/S",
                                                    "quads/inst", "clock GHz", "inst/unit")]
for kind in sorted(table):
    row = table[kind]
    lines.append("%-46s %6.2f %6.2f %6.2f %6.2f  %10.3f %9.2f %10.2f" % (
        NAMES[kind], row[1][0], row[2][0], row[3][0], row[4][0], row[4][1], row[4][2], row[4][3]))
text = "\n".join(lines) + "\n"
with open(os.path.join(root, "profiles", "%s_issue_rate_probe.txt" % tag), "w") as f:
    f.write(text)
print(text)
