"""Reduce the rocprofv3 CSVs of one profiling directory IN PLACE (run on the GPU box by
tools/gpu_profile_r4.sh): every *_counter_collection.csv becomes a *_counter_collection.json with, per
kernel name, the sum of every counter over its dispatches, the number of dispatches and their total
duration — what tools/summarise_roofline.py needs; the per-dispatch CSVs of a shuffled call (one
k_order_level launch per level and chunk) are tens of megabytes and do not travel.  Kernel traces are
dropped (the --stats summary of the trace pass stays).

    python tools/reduce_counters.py <directory>
"""
import collections
import csv
import glob
import json
import os
import sys

root = sys.argv[1]
for path in glob.glob(os.path.join(root, "**", "*_counter_collection.csv"), recursive=True):
    kernels = collections.defaultdict(lambda: {"counters": collections.defaultdict(float), "dispatches": {}})
    with open(path) as f:
        for r in csv.DictReader(f):
            k = kernels[r["Kernel_Name"]]
            k["counters"][r["Counter_Name"]] += float(r["Counter_Value"])
            k["dispatches"][r["Dispatch_Id"]] = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    out = {name: {"counters": dict(k["counters"]), "dispatches": len(k["dispatches"]),
                  "ns": float(sum(k["dispatches"].values()))} for name, k in kernels.items()}
    with open(path[:-4] + ".json", "w") as f:
        json.dump(out, f)
    os.remove(path)
for path in glob.glob(os.path.join(root, "**", "*_kernel_trace.csv"), recursive=True):
    os.remove(path)
