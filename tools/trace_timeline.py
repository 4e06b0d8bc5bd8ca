#!/usr/bin/env python3
"""Summarise a rocprofv3 --kernel-trace csv: per kernel name (shortened) calls / total / mean, and for the last N milliseconds
of the trace the busy time (union of kernel intervals) against the wall time.  python tools/trace_timeline.py <dir> [last_ms]"""
import csv, glob, os, re, sys, collections
d = sys.argv[1]
last_ms = float(sys.argv[2]) if len(sys.argv) > 2 else 0
f = glob.glob(os.path.join(d, '**', '*_kernel_trace.csv'), recursive=True)[0]
rows = [(int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name']) for r in csv.DictReader(open(f))]
rows.sort()
t1 = rows[-1][1]
if last_ms > 0:
    rows = [r for r in rows if r[0] >= t1 - last_ms * 1e6]
t0 = rows[0][0]
def short(n):
    n = re.sub(r'^void ', '', n)
    m = re.search(r'stage_kernel<rdr::(?:LeanStage|MidStage)?<?rdr::([A-Za-z0-9_]+)', n) or re.search(r'stage_kernel<rdr::([A-Za-z0-9_]+)', n)
    if m: return m.group(1)
    return re.sub(r'\(.*', '', n)[:48]
agg = collections.defaultdict(lambda: [0, 0.0])
for s, e, n in rows:
    a = agg[short(n)]; a[0] += 1; a[1] += (e - s) * 1e-6
busy, cur_s, cur_e = 0.0, None, None
for s, e, _ in rows:
    if cur_e is None or s > cur_e:
        if cur_e is not None: busy += cur_e - cur_s
        cur_s, cur_e = s, e
    else: cur_e = max(cur_e, e)
busy += cur_e - cur_s
print('%d launches in %.2f ms wall, GPU busy %.2f ms, kernel time summed %.2f ms' % (len(rows), (t1 - t0) * 1e-6, busy * 1e-6, sum(a[1] for a in agg.values())))
for k, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:28]:
    print('  %-40s %5d calls  %8.3f ms  mean %7.1f us' % (k, c, t, t / c * 1e3))
