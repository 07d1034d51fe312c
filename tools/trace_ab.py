#!/usr/bin/env python
"""Per-(kernel, shape) A/B of two libraries on the bench step (GPU box):
   python tools/trace_ab.py <out dir> <libA.so> <libB.so> [kernel substring]
Runs `bench.py --steps 12 --warmup 4` under rocprofv3 --kernel-trace once per library (launch log next to the trace), keys the trace rows on
the launch log's shape tags (as tools/profile_summarize.py does) and prints average microseconds per launch side by side, largest
difference per step first."""
import collections, csv, glob, os, subprocess, sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import trace_align

out, libs, pat = os.path.abspath(sys.argv[1]), sys.argv[2:4], (sys.argv[4] if len(sys.argv) > 4 else '')
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run(tag, lib):
    d = os.path.join(out, tag)
    os.makedirs(d, exist_ok=True)
    env = dict(os.environ, FPD_LAUNCH_LOG=os.path.join(d, 'launch.log'), TMPDIR='/tmp')
    if lib != '-':
        env['FPD_AMD_LIB'] = os.path.abspath(lib)
    else:
        env.pop('FPD_AMD_LIB', None)
    subprocess.run(['rocprofv3', '--kernel-trace', '-f', 'csv', '-d', d, '-o', 'b', '--', sys.executable, os.path.join(root, 'bench.py'),
                    '--no-cpu-baseline', '--no-parity', '--no-phase-times', '--steps', '12', '--warmup', '4'], cwd='/tmp', env=env,
                   stdout=open(d + '.log', 'w'), stderr=subprocess.STDOUT)
    rows = [r for f in glob.glob(d + '/**/*kernel_trace.csv', recursive=True) for r in csv.DictReader(open(f))]
    log = trace_align.parse_log(open(os.path.join(d, 'launch.log')))
    tags, bad, n_trace, n_log = trace_align.align(log, rows)
    if bad or n_trace != n_log:
        print('%s: %d trace rows vs %d logged launches, %d name mismatches' % (tag, n_trace, n_log, bad))
    ours = trace_align.library_rows(rows, set(l[0] for l in log))
    steps = sum(1 for r in ours if 'adam_kernel' in r['Kernel_Name'])
    agg = collections.defaultdict(list)
    for r, l in zip(ours, log):
        if r['Dispatch_Id'] in tags:
            agg[(l[0], l[4])].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
    return agg, steps


a, sa = run('A', libs[0])
b, sb = run('B', libs[1])
rows = []
for k in set(a) | set(b):
    if pat and pat not in k[0]:
        continue
    ua = sum(a[k]) / len(a[k]) if a.get(k) else 0.0
    ub = sum(b[k]) / len(b[k]) if b.get(k) else 0.0
    n = len(b.get(k, a.get(k))) / max(sb, 1)
    rows.append(((ub - ua) * n, k, n, ua, ub))
rows.sort(key=lambda r: -abs(r[0]))
print('%-18s %-70s %6s %8s %8s %9s' % ('kernel', 'shape', 'n/step', 'A us', 'B us', 'd us/step'))
for d, k, n, ua, ub in rows[:60]:
    print('%-18s %-70s %6.1f %8.2f %8.2f %9.1f' % (k[0][:18], k[1][:70], n, ua, ub, d))
print('total difference (B - A): %.1f us per step of serialised kernel time; steps %d / %d' % (sum(r[0] for r in rows), sa, sb))
