#!/bin/bash
# kernel trace of tools/pipe_profile_run.py: the last path as a timeline per hardware queue -- where the panel passes lie,
# what every chain's queue does meanwhile, and how long the chains' kernels take beside a panel pass and alone
#   tools/pipe_trace.sh [chains]       (BESSX_TEST_HOOKS selects the variant)
root=${GRAFT_REPO_ROOT:-/root/repo}
out=/tmp/pipetrace; rm -rf $out; mkdir -p $out $root/gpurun_out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $out -o t -- python3 $root/tools/${PIPE_TRACE_RUNNER:-pipe_profile_run.py} ${1:-4} > $root/gpurun_out/pipe_trace.log 2>&1
python3 - $out > $root/gpurun_out/pipe_trace_summary.txt <<'PY'
import csv, glob, sys, collections, statistics
rows = []
for fn in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(fn)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"][:28], r.get("Queue_Id", "?")))
rows.sort()
# paths: split by gaps > 15 ms
paths, cur = [], [rows[0]]
for a, b in zip(rows, rows[1:]):
    if b[0] - a[1] > 15_000_000:
        paths.append(cur); cur = []
    cur.append(b)
paths.append(cur)
paths = [q for q in paths if any("k_cov_panel" in r[2] for r in q)]
seg = paths[-1]
t0 = seg[0][0]
print("last path: %d kernels, span %.3f ms" % (len(seg), (seg[-1][1] - t0) / 1e6))
panels = [(s, e, n, q) for s, e, n, q in seg if "k_cov_panel" in n and e - s > 100_000]
print("panel passes:", [(round((s - t0) / 1e6, 2), round((e - s) / 1e6, 2), q) for s, e, n, q in panels])
print("panel total %.3f ms" % (sum(e - s for s, e, _, _ in panels) / 1e6))
def beside(s, e):
    return any(ps < e and s < pe for ps, pe, _, _ in panels)
byq = collections.defaultdict(list)
for r in seg:
    byq[r[3]].append(r)
for q, rs in sorted(byq.items()):
    busy = sum(e - s for s, e, _, _ in rs)
    print("queue %s: %4d kernels, first at %.2f ms, last ends %.2f ms, busy %.2f ms" % (q, len(rs), (rs[0][0] - t0) / 1e6, (rs[-1][1] - t0) / 1e6, busy / 1e6))
names = collections.defaultdict(lambda: [[], []])
for s, e, n, q in seg:
    if "k_cov_panel" in n: continue
    names[n][1 if beside(s, e) else 0].append((e - s) / 1e3)
print("%-30s %22s %22s" % ("kernel (us)", "alone: n median", "beside a panel pass: n median"))
for n, (a, b) in sorted(names.items(), key=lambda kv: -(sum(kv[1][0]) + sum(kv[1][1])))[:14]:
    print("%-30s %8d %10.1f %12d %10.1f" % (n, len(a), statistics.median(a) if a else 0, len(b), statistics.median(b) if b else 0))
# what one chain's queue runs (the busiest queue that ran no panel pass)
cq = max((q for q in byq if not any("k_cov_panel" in r[2] and r[1] - r[0] > 100_000 for r in byq[q])), key=lambda q: len(byq[q]))
cnt = collections.Counter(r[2] for r in byq[cq])
dur = collections.defaultdict(list)
for s_, e_, n_, _ in byq[cq]:
    dur[n_].append((e_ - s_) / 1e3)
print("queue %s by kernel:" % cq)
for n_, c_ in cnt.most_common():
    d_ = sorted(dur[n_])
    print("   %-30s n=%4d  median %.1f us  p10 %.1f  p90 %.1f  total %.2f ms" % (n_, c_, statistics.median(d_), d_[len(d_) // 10], d_[(9 * len(d_)) // 10], sum(d_) / 1e3))
seq_ = [r[2][7:22] for r in byq[cq][60:100]]
print("   a stretch of it:", seq_)
# per queue: gaps between consecutive kernels that overlap a panel pass vs not
for q, rs in sorted(byq.items()):
    ga, gb = [], []
    for x, y2 in zip(rs, rs[1:]):
        g = (y2[0] - x[1]) / 1e3
        (gb if beside(x[1], y2[0]) else ga).append(g)
    if len(rs) > 20:
        print("queue %s gaps us: alone n=%d median %.1f p90 %.1f | beside n=%d median %.1f p90 %.1f" % (
            q, len(ga), statistics.median(ga) if ga else 0, sorted(ga)[int(0.9 * len(ga))] if ga else 0,
            len(gb), statistics.median(gb) if gb else 0, sorted(gb)[int(0.9 * len(gb))] if gb else 0))
PY
cat $root/gpurun_out/pipe_trace_summary.txt
