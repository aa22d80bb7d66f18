#!/bin/bash
# kernel trace of tools/mc_profile_run.py (merged chunk chains): per-kernel gaps of the last path's chunk phase
root=${GRAFT_REPO_ROOT:-/root/repo}
out=/tmp/mctrace; rm -rf $out; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $out -o t -- python3 $root/tools/mc_profile_run.py ${1:-4} > $root/gpurun_out/mc_trace.log 2>&1
python3 - $out > $root/gpurun_out/mc_trace_summary.txt <<'PY'
import csv, glob, sys
rows = []
for fn in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(fn)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"][:40]))
rows.sort()
# the last path: from the last k_mc_status-free stretch... take the last 700 kernels and find the mc kernels of the last path
mc = [i for i, r in enumerate(rows) if "k_mc_" in r[2]]
# split mc kernels into paths by gaps > 3 ms
paths, cur = [], [mc[0]]
for a, b in zip(mc, mc[1:]):
    if rows[b][0] - rows[a][1] > 3_000_000:
        paths.append(cur); cur = []
    cur.append(b)
paths.append(cur)
last = paths[-1]
i0, i1 = last[0], last[-1]
seg = rows[i0:i1 + 1]
busy = sum(e - s for s, e, _ in seg)
span = seg[-1][1] - seg[0][0]
print("chunk phase of the last path: %d kernels, span %.3f ms, kernels busy %.3f ms, idle %.3f ms" % (len(seg), span / 1e6, busy / 1e6, (span - busy) / 1e6))
by = {}
for s, e, n in seg:
    by.setdefault(n, [0, 0]); by[n][0] += 1; by[n][1] += e - s
for n, (c, t) in sorted(by.items(), key=lambda kv: -kv[1][1]):
    print("  %-42s calls %4d total %.3f ms avg %.1f us" % (n, c, t / 1e6, t / c / 1e3))
gaps = sorted(((seg[i + 1][0] - seg[i][1]), seg[i][2], seg[i + 1][2]) for i in range(len(seg) - 1))
print("largest gaps (us):", [(round(g / 1e3, 1), a[:18], b[:18]) for g, a, b in gaps[-12:]])
print("median gap us:", gaps[len(gaps) // 2][0] / 1e3)
# what precedes / follows the chunk phase
print("before:", [(r[2][:24], round((r[1]-r[0])/1e3,1)) for r in rows[max(0,i0-6):i0]])
print("after:", [(r[2][:24], round((r[1]-r[0])/1e3,1), round((r[0]-rows[i1][1])/1e3,1)) for r in rows[i1+1:i1+8]])
PY
cat $root/gpurun_out/mc_trace_summary.txt
