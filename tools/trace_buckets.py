"""Per-kernel average durations along a path from a rocprofv3 --kernel-trace CSV, in 10 buckets of path time
(diagnostic: how a launch's cost grows with the sparsity level).  python tools/trace_buckets.py <kernel_trace.csv> [first kernel name]"""
import collections
import csv
import sys


def main():
    rows = list(csv.DictReader(open(sys.argv[1])))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    first = sys.argv[2] if len(sys.argv) > 2 else "k_topk"

    def nm(r):
        return r["Kernel_Name"].split("(")[0].replace("void bessx::", "").replace("bessx::", "").split("<")[0]
    names = [nm(r) for r in rows]
    i0 = names.index(first)
    t0, tend = int(rows[i0]["Start_Timestamp"]), int(rows[-1]["End_Timestamp"])
    nb = 10
    agg = [collections.defaultdict(lambda: [0, 0.0]) for _ in range(nb)]
    for r in rows[i0:]:
        b = min(nb - 1, int((int(r["Start_Timestamp"]) - t0) * nb / (tend - t0)))
        a = agg[b][nm(r)]
        a[0] += 1
        a[1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    for b in range(nb):
        print(b, " ".join("%s:%d/%.1f" % (k.replace("k_", ""), v[0], v[1] / v[0])
                          for k, v in sorted(agg[b].items(), key=lambda kv: -kv[1][1])[:9]))
    busy = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in rows[i0:])
    gaps = sum(max(0, int(b["Start_Timestamp"]) - int(a["End_Timestamp"])) for a, b in zip(rows[i0:-1], rows[i0 + 1:]))
    print("path span ms", (tend - t0) / 1e6, "busy", busy / 1e6, "gaps", gaps / 1e6)


if __name__ == "__main__":
    main()
