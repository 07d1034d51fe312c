#!/usr/bin/env python
"""Aggregates the rocprofv3 output of tools/profile.sh into the small tables kept under profiles/ (rNN = the prefix argument):
  rNN_kernel_stats.csv      rocprofv3's own --stats table
  rNN_per_shape.csv         per (kernel, grid): calls per step, average us, ms per step (the --stats averages mix shapes)
  rNN_hbm_traffic.csv       per (kernel, grid): FETCH_SIZE / WRITE_SIZE per launch (separate --pmc passes), HBM bytes per launch
                            = (2 x FETCH + WRITE) x 1024 (gfx950: FETCH_SIZE counts wide coalesced reads at 1/2, MI355X_MICROARCH.md),
                            average duration from the un-counted trace, GB/s and the fraction of the 8 TB/s HBM3E peak
  rNN_pmc_bneck64.json      the same for the dominant kernel from its micro-benchmark (what bench.py's roofline.traffic quotes)"""
import collections
import csv
import glob
import json
import os
import shutil
import sys

out = sys.argv[1]
PFX = sys.argv[2] if len(sys.argv) > 2 else 'r02'
STEPS = None      # set from the trace: one adam_kernel launch per student step (eager + warm-up + timed + the host-enqueue probe step)


def short(n):
    return n.replace('(anonymous namespace)::', '').split('(')[0].replace('void ', '')[:60]


def trace(sub):
    rows = []
    for f in glob.glob('%s/%s/**/*kernel_trace.csv' % (out, sub), recursive=True):
        rows += list(csv.DictReader(open(f)))
    return rows


# 1. stats
for f in glob.glob('%s/stats/**/*kernel_stats.csv' % out, recursive=True):
    shutil.copy(f, os.path.join(out, PFX + '_kernel_stats.csv'))
dur = collections.defaultdict(list)
rows_stats = sorted(trace('stats'), key=lambda r: int(r['Start_Timestamp']))
adams = [i for i, r in enumerate(rows_stats) if 'adam_kernel' in r['Kernel_Name']]
STEPS = len(adams)
# bench.py re-times single recorded ops live after its last step (roofline.avg_us, conv_classes): not part of a step
tail = sum(1 for r in rows_stats[adams[-1] + 1:] if 'adam_tick' not in r['Kernel_Name'])
rows_stats = rows_stats[:adams[-1] + 2]
print('%d steps in the trace; %d kernel records behind the last step (live re-timing of single ops) left out' % (STEPS, tail))
for r in rows_stats:
    grid = int(r['Grid_Size_X']) * int(r['Grid_Size_Y']) * int(r['Grid_Size_Z'])
    wg = int(r['Workgroup_Size_X']) * int(r['Workgroup_Size_Y']) * int(r['Workgroup_Size_Z'])
    dur[(short(r['Kernel_Name']), str(grid), str(wg))].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
with open(os.path.join(out, PFX + '_per_shape.csv'), 'w', newline='') as f:
    w = csv.writer(f)
    w.writerow(['kernel', 'grid_threads', 'workgroup', 'calls_per_step', 'avg_us', 'ms_per_step'])
    for k, v in sorted(dur.items(), key=lambda kv: -sum(kv[1])):
        w.writerow([k[0], k[1], k[2], round(len(v) / STEPS, 1), round(sum(v) / len(v), 2), round(sum(v) / STEPS / 1e3, 3)])
total = sum(sum(v) for v in dur.values()) / STEPS / 1e3
print('serialised kernel time per step: %.2f ms' % total)

# 2. counters
def counters(sub, name):
    agg = collections.defaultdict(list)
    for f in glob.glob('%s/%s/**/*counter_collection.csv' % (out, sub), recursive=True):
        for r in csv.DictReader(open(f)):
            if r['Counter_Name'] == name:
                agg[(short(r['Kernel_Name']), r['Grid_Size'])].append(float(r['Counter_Value']))
    return agg


fe, wr = counters('step_FETCH_SIZE', 'FETCH_SIZE'), counters('step_WRITE_SIZE', 'WRITE_SIZE')
with open(os.path.join(out, PFX + '_hbm_traffic.csv'), 'w', newline='') as f:
    w = csv.writer(f)
    w.writerow(['kernel', 'grid_threads', 'launches_counted', 'FETCH_KB', 'WRITE_KB', 'hbm_MB_per_launch', 'avg_us', 'GB_per_s', 'frac_of_8TBs'])
    for k in sorted(set(fe) | set(wr), key=lambda k: -sum(fe.get(k, [0])) - sum(wr.get(k, [0]))):
        a, b = fe.get(k, []), wr.get(k, [])
        if not a or not b:
            continue
        fkb, wkb = sum(a) / len(a), sum(b) / len(b)
        byt = (2 * fkb + wkb) * 1024
        us = [v for kk, v in dur.items() if kk[0] == k[0] and kk[1] == k[1]]
        us = sum(us[0]) / len(us[0]) if us else 0.0
        gbs = byt / us / 1e3 if us else 0.0
        w.writerow([k[0], k[1], len(a), round(fkb, 1), round(wkb, 1), round(byt / 1e6, 2), round(us, 2), round(gbs, 1), round(gbs / 8000.0, 3)])
res = {}
for c in ('FETCH_SIZE', 'WRITE_SIZE'):
    vals = [v for k, vs in counters('bneck_' + c, c).items() if 'bneck_eval_kernel' in k[0] for v in vs]
    res[c] = sum(vals) / max(len(vals), 1)
    res[c + '_launches'] = len(vals)
res['hbm_bytes_per_launch'] = (2.0 * res['FETCH_SIZE'] + res['WRITE_SIZE']) * 1024
res['grid_cap'] = int(os.environ.get('FPD_BNECK_BLOCKS', '128'))       # the persistent grid the micro-benchmark launched (bench.py refuses a mismatch)
res['note'] = ('fused teacher Bottleneck N=32 64x64 C=256 P=128 (weight tiles by LDS-DMA, conv1 rows kept in a ring), launched at the grid cap of the step; rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE in '
               'separate passes, mean over launches, KB -> bytes, read side doubled per the gfx950 calibration in MI355X_MICROARCH.md')
json.dump(res, open(os.path.join(out, PFX + '_pmc_bneck64.json'), 'w'), indent=1)
print(json.dumps(res))
