"""Turn rocprofv3 output directories (gpurun_out/, scratch) into the committed summaries under profiles/.

  python tools/summarize_profiles.py stats  <dir of --kernel-trace --stats run>  profiles/<name>.csv
  python tools/summarize_profiles.py pmc    <dir of --pmc FETCH_SIZE run> <dir of --pmc WRITE_SIZE run> \
                                            profiles/pmc_traffic.json  [kernel substring: "k_xtv<8, 16, false"
                                            (default, streaming score pass) or "k_cov_panel_lds"]

Commands the directories come from (one gpurun call, each profiler pass its own process; --pmc never together
with the sys/hip/hsa traces):
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline
  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_fetch -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --kmax 20
  rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_write -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --kmax 20

gfx950 correction (MI355X_MICROARCH.md, HBM section): FETCH_SIZE reports half of the bytes of a wide coalesced
streaming read, so the fetch figure is doubled; both counters are in KB.
"""
import csv
import glob
import json
import os
import statistics
import sys


def find(d, suffix):
    hits = sorted(glob.glob(os.path.join(d, "**", "*" + suffix), recursive=True), key=os.path.getmtime)
    if not hits:
        raise SystemExit("no *%s under %s" % (suffix, d))
    return hits[-1]  # the newest: gpurun_out/ keeps the files of earlier collections next to the new ones


def stats(src, dst):
    rows = list(csv.reader(open(find(src, "kernel_stats.csv"))))
    with open(dst, "w", newline="") as f:
        csv.writer(f).writerows(rows)
    print("wrote %s (%d kernels)" % (dst, len(rows) - 1))


def counter_values(d, kernel, counter):
    vals = []
    with open(find(d, "counter_collection.csv")) as f:
        for r in csv.DictReader(f):
            if kernel in r["Kernel_Name"] and r["Counter_Name"] == counter:
                vals.append(float(r["Counter_Value"]))
    return vals


def pmc(fetch_dir, write_dir, dst, kernel, key):
    """Median FETCH_SIZE / WRITE_SIZE of the dispatches of `kernel` that really streamed X, merged into dst under
    `key` (k_xtv: one launch = one pass over X; k_cov_panel: one 32-column group = one pass, a launch runs one
    group unless more than 32 columns were missing)."""
    fv = counter_values(fetch_dir, kernel, "FETCH_SIZE")
    wv = counter_values(write_dir, kernel, "WRITE_SIZE")
    big = [v for v in fv if v > 0.5 * statistics.median([x for x in fv if x > 0.05 * max(fv)])]
    big = [v for v in big if v < 1.5 * statistics.median(big)]  # drop the rare two-group launches
    wbig = [v for v in wv if v > 0.05 * max(wv)]
    out = json.load(open(dst)) if os.path.exists(dst) else {}
    entry = {
        "kernel": kernel,
        "FETCH_SIZE_KB_median": statistics.median(big),
        "WRITE_SIZE_KB_median": statistics.median(wbig),
        "dispatches": len(big),
    }
    entry["hbm_bytes"] = 1024.0 * (2.0 * entry["FETCH_SIZE_KB_median"] + entry["WRITE_SIZE_KB_median"])
    out[key + "_detail"] = entry
    out[key] = entry["hbm_bytes"]
    out["source"] = ("rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes), python3 bench.py --steps 1 "
                     "--warmup 0 --kmax 20..60 [--score-mode streaming]")
    out["correction"] = ("gfx950: FETCH_SIZE counts half of the bytes of a wide coalesced streaming read "
                         "(MI355X_MICROARCH.md, HBM) -> doubled; WRITE_SIZE exact")
    out["algorithmic_bytes_per_launch"] = 8.0 * 50000 * 10000
    json.dump(out, open(dst, "w"), indent=1)
    print(json.dumps(entry, indent=1))


if __name__ == "__main__":
    if len(sys.argv) >= 4 and sys.argv[1] == "stats":
        stats(sys.argv[2], sys.argv[3])
    elif len(sys.argv) >= 5 and sys.argv[1] == "pmc":
        kern = sys.argv[5] if len(sys.argv) > 5 else "k_xtv<8, 16, false"
        pmc(sys.argv[2], sys.argv[3], sys.argv[4], kern,
            "k_cov_panel_hbm_bytes_per_pass" if "cov_panel" in kern else "k_xtv_hbm_bytes_per_launch")
    else:
        raise SystemExit(__doc__)
