#!/bin/bash
# HBM-side fetch traffic of the two matrix products of the exact diagonalisation (FETCH_SIZE, own pass)
set -o pipefail
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/r2edpmc
rm -rf $OUT; mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout -k 10 400 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/plain -- python3 tools/time_plain_matvec.py > $OUT/plain.log 2>&1; echo "plain rc=$?" | tee -a $OUT/status.txt
timeout -k 10 400 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/sector -- python3 -m annealing_sign_problem_amd.sector_ed --model heisenberg_kagome_36 --output /tmp/k36.h5 --max-iterations 20 --tol 1e-30 > $OUT/sector.log 2>&1; echo "sector rc=$?" | tee -a $OUT/status.txt
python - <<'PY' | tee $OUT/fetch_summary.txt
import csv, glob, collections
for tag in ("plain", "sector"):
    files = glob.glob("gpurun_out/r2edpmc/%s/**/*counter_collection.csv" % tag, recursive=True)
    acc, dur = collections.defaultdict(list), collections.defaultdict(list)
    for f in files:
        for r in csv.DictReader(open(f)):
            if r.get("Counter_Name") != "FETCH_SIZE":
                continue
            for k in ("k_plain_matvec", "k_sector_matvec", "k_sector_rows", "k_sector_generate"):
                if k in r["Kernel_Name"]:
                    acc[k].append(float(r["Counter_Value"]))
                    dur[k].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    for k, v in acc.items():
        mean = sum(v) / len(v)
        ms = sum(dur[k]) / len(v) / 1e6
        gb = mean * 1024 * 2 / 1e9  # FETCH_SIZE is in KB and counts half of the true bytes on gfx950 (tools/fetch_calibrate.hip)
        print("%s: %d launches, %.2f ms, fetched %.2f GB per launch = %.2f TB/s" % (k, len(v), ms, gb, gb / ms))
PY
rm -rf $OUT/plain $OUT/sector
