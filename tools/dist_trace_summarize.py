#!/usr/bin/env python
"""Summarise a rocprofv3 --kernel-trace CSV of `bench.py --force-dist`: per hardware queue the kernels it ran, how much of
the traced wall time had >= 2 kernels in flight (i.e. whether the tracer let the streams overlap), and for every collective
kernel (RCCL: names containing nccl / rccl / AllReduce) the share of its duration during which kernels of OTHER queues
(the student backward and its weight-gradient lane) were running.
   python tools/dist_trace_summarize.py <dir with *kernel_trace.csv> > profiles/rNN_dist_trace_summary.txt"""
import csv, glob, os, re, sys
from collections import defaultdict

d = sys.argv[1]
files = glob.glob(os.path.join(d, '**', '*kernel_trace.csv'), recursive=True)
if not files:
    sys.exit('no *kernel_trace.csv under %s' % d)
rows = []
for f in files:
    with open(f) as fh:
        for r in csv.DictReader(fh):
            rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r.get('Queue_Id', '?'), r['Kernel_Name']))
rows.sort()
t0, t1 = rows[0][0], max(r[1] for r in rows)
print('%d kernel records, %.3f ms from the first start to the last end' % (len(rows), (t1 - t0) / 1e6))
byq = defaultdict(list)
for s, e, q, n in rows:
    byq[q].append((s, e, n))
short = lambda n: re.sub(r'\(anonymous namespace\)::|void |unsigned short, ', '', n)[:70]
for q, ks in sorted(byq.items(), key=lambda kv: -len(kv[1])):
    busy = sum(e - s for s, e, _ in ks)
    names = defaultdict(int)
    for _, _, n in ks:
        names[short(n)] += 1
    top = sorted(names.items(), key=lambda kv: -kv[1])[:3]
    print('queue %-4s %6d kernels, busy %8.3f ms: %s' % (q, len(ks), busy / 1e6, '; '.join('%s x%d' % t for t in top)))
# time with >= 2 kernels in flight (sweep line)
ev = []
for s, e, q, n in rows:
    ev.append((s, 1)); ev.append((e, -1))
ev.sort()
depth, last, multi, anyk = 0, ev[0][0], 0, 0
for t, dlt in ev:
    if depth >= 1: anyk += t - last
    if depth >= 2: multi += t - last
    depth += dlt; last = t
print('time with >= 1 kernel running %.3f ms, with >= 2 running %.3f ms (%.1f %% of the busy time): the tracer %s' % (
    anyk / 1e6, multi / 1e6, 100.0 * multi / max(anyk, 1), 'lets queues overlap' if multi > 0.02 * anyk else 'SERIALISES the queues'))
coll = [(s, e, q, n) for s, e, q, n in rows if re.search(r'nccl|rccl|allreduce', n, re.I)]
print('%d collective kernel records' % len(coll))
tot = ov = 0
for s, e, q, n in coll:
    cover = sorted((max(s, s2), min(e, e2)) for s2, e2, q2, _ in rows if q2 != q and s2 < e and e2 > s)
    c, cur = 0, s
    for a, b in cover:
        a = max(a, cur)
        if b > a:
            c += b - a; cur = b
    tot += e - s; ov += c
for s, e, q, n in coll[:12]:
    print('  %-60s queue %s  +%.3f ms  %.1f us' % (short(n), q, (s - t0) / 1e6, (e - s) / 1e3))
if coll:
    print('collective kernel time %.3f ms in total, %.1f %% of it with kernels of other queues in flight' % (tot / 1e6, 100.0 * ov / max(tot, 1)))
