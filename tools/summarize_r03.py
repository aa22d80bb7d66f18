"""Runs ON the GPU box at the end of tools/collect_profiles.sh: turns the rocprofv3 output directories under
gpurun_out/prof_r03 into small summaries (gpurun copies back at most 64 MiB) and deletes the raw traces.
  stats_*  -> <name>_kernel_stats.csv (the --stats table as it is)
  pmc_*    -> pmc_summary.json: per kernel (template instance) the dispatch count and the median / sum of every counter
              collected in that pass, plus the median duration of the same dispatches
FETCH_SIZE / WRITE_SIZE are in KB; on gfx950 FETCH_SIZE counts half the bytes of a wide streaming read
(MI355X_MICROARCH.md, HBM) -- the doubling is applied where the numbers are used (profiles/README.md)."""
import csv
import glob
import json
import os
import shutil
import statistics
import sys

root = sys.argv[1]
out = {}
for d in sorted(glob.glob(os.path.join(root, "*"))):
    name = os.path.basename(d)
    if not os.path.isdir(d):
        continue
    if name.startswith("stats_"):
        hits = glob.glob(os.path.join(d, "**", "*kernel_stats.csv"), recursive=True)
        if hits:
            shutil.copy(hits[-1], os.path.join(root, name + "_kernel_stats.csv"))
    elif name.startswith("pmc_"):
        cc = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
        if not cc:
            continue
        per = {}
        with open(cc[-1]) as f:
            for r in csv.DictReader(f):
                k = r["Kernel_Name"].split("(")[0].replace("void bessx::", "").replace("bessx::", "")
                e = per.setdefault(k, {})
                e.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
                if "Start_Timestamp" in r and "End_Timestamp" in r:
                    e.setdefault("_dur_ns", {})[r.get("Dispatch_Id", str(len(e)))] = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
        summ = {}
        for k, e in per.items():
            row = {}
            for cn, vals in e.items():
                if cn == "_dur_ns":
                    dv = list(vals.values())
                    row["duration_ns_median"] = statistics.median(dv)
                    row["duration_ns_sum"] = sum(dv)
                else:
                    big = [v for v in vals if v > 0.05 * max(vals)] if max(vals) > 0 else vals
                    row[cn] = {"dispatches": len(vals), "dispatches_counted": len(big), "median": statistics.median(big),
                               "max": max(vals), "sum": sum(vals)}
            summ[k] = row
        out[name] = summ
    shutil.rmtree(d, ignore_errors=True)
json.dump(out, open(os.path.join(root, "pmc_summary.json"), "w"), indent=1)
print("summarised", len(out), "counter passes")
