"""How much of a rocprofv3 --kernel-trace runs concurrently (diagnostic for the fold chains that run side by side):
for the kernels from the LAST occurrence of a marker kernel on (default k_col_normalize = session set-up), the wall
span, the union of the busy intervals, the sum of the durations, and the same per queue.
python tools/trace_overlap.py <kernel_trace.csv> [tail fraction of the trace to analyse, default 0.3]"""
import collections
import csv
import sys


def main():
    rows = list(csv.DictReader(open(sys.argv[1])))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    frac = float(sys.argv[2]) if len(sys.argv) > 2 else 0.3
    t_lo = int(rows[0]["Start_Timestamp"])
    t_hi = int(rows[-1]["End_Timestamp"])
    cut = t_hi - int((t_hi - t_lo) * frac)
    rows = [r for r in rows if int(r["Start_Timestamp"]) >= cut]

    def nm(r):
        return r["Kernel_Name"].split("(")[0].replace("void bessx::", "").replace("bessx::", "").split("<")[0]
    iv = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r.get("Queue_Id", "?"), nm(r)) for r in rows]
    span = iv[-1][1] - iv[0][0]
    total = sum(b - a for a, b, _, _ in iv)
    union, cur_a, cur_b = 0, None, None
    for a, b, _, _ in iv:
        if cur_b is None or a > cur_b:
            if cur_b is not None:
                union += cur_b - cur_a
            cur_a, cur_b = a, b
        else:
            cur_b = max(cur_b, b)
    union += cur_b - cur_a
    print("kernels %d  span %.3f ms  union busy %.3f ms  sum of durations %.3f ms" % (len(iv), span / 1e6, union / 1e6, total / 1e6))
    perq = collections.defaultdict(lambda: [0, 0])
    for a, b, q, _ in iv:
        perq[q][0] += 1
        perq[q][1] += b - a
    for q, (c, t) in sorted(perq.items()):
        print("  queue %s: %d kernels, %.3f ms" % (q, c, t / 1e6))
    pern = collections.defaultdict(lambda: [0, 0])
    for a, b, _, n in iv:
        pern[n][0] += 1
        pern[n][1] += b - a
    for n, (c, t) in sorted(pern.items(), key=lambda kv: -kv[1][1])[:14]:
        print("  %-28s %5d  %.3f ms  avg %.1f us" % (n, c, t / 1e6, t / c / 1e3))


if __name__ == "__main__":
    main()
