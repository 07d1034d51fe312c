#!/usr/bin/env python
"""Aggregates the rocprofv3 output of tools/profile.sh into the small tables kept under profiles/ (rNN = the prefix argument):
  rNN_kernel_stats.csv      rocprofv3's own --stats table
  rNN_per_shape.csv         per (kernel, grid, SHAPE): calls per step, average us, ms per step.  The shape (plan op kind, N/H/W/C/K/R,
                            forward / data gradient, fused extras) is not in the trace; it comes from the library's launch log of the
                            same process (FPD_LAUNCH_LOG, include: csrc/common.h FPD_LAUNCH): one line per kernel launch in host
                            order, matched 1:1 with the trace rows of the library's kernels sorted by Dispatch_Id (the same host
                            order) and verified by kernel name.  So e.g. the fused Bottleneck's 64x64 launches and its 32x32
                            launches (same persistent grid) are separate rows, and every figure bench.py quotes for a (kernel, shape)
                            can be recomputed from this file.
  rNN_hbm_traffic.csv       per (kernel, grid, shape): FETCH_SIZE / WRITE_SIZE per launch (separate --pmc passes, each with its own
                            launch log), HBM bytes per launch = (2 x FETCH + WRITE) x 1024 (gfx950: FETCH_SIZE counts wide coalesced
                            reads at 1/2, MI355X_MICROARCH.md), average duration from the un-counted trace, GB/s and the fraction
                            of the 8 TB/s HBM3E peak
  rNN_pmc_bneck64.json      the same for the dominant kernel from its micro-benchmark (what bench.py's roofline.traffic quotes)"""
import collections
import csv
import glob
import json
import os
import re
import shutil
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import trace_align

out = sys.argv[1]
PFX = sys.argv[2] if len(sys.argv) > 2 else 'r04'
SUB = sys.argv[3] if len(sys.argv) > 3 else ''      # 'hrnet_': the passes over `bench.py --config hrnet` (directories hrnet_stats, hrnet_step_*)
STEPS = None      # set from the trace: one adam_kernel launch per student step (eager + warm-up + timed + the host-enqueue probe step)


def short(n):
    return n.replace('(anonymous namespace)::', '').split('(')[0].replace('void ', '')[:60]


def load_log(sub):
    """[(kernel base name, grid blocks, block threads, op index, shape tag)] in host order, or None."""
    path = os.path.join(out, sub, 'launch.log')
    if not os.path.exists(path):
        return None
    return trace_align.parse_log(open(path))


def trace(sub):
    rows = []
    for f in glob.glob('%s/%s/**/*kernel_trace.csv' % (out, sub), recursive=True):
        rows += list(csv.DictReader(open(f)))
    return rows


def shapes_by_dispatch(sub, rows):
    """Dispatch_Id -> shape tag for the library's kernels of trace `rows`, by 1:1 alignment with the launch log."""
    log = load_log(sub)
    if not log:
        print('%s: no launch log -- rows are keyed on (kernel, grid) only' % sub)
        return {}
    tags, bad, n_trace, n_log = trace_align.align(log, rows, strict=os.environ.get('FPD_ALIGN_LENIENT') != '1')   # drift = wrong keys: an error
    print('%s: %d launches keyed on their shape, %d name mismatches' % (sub, len(tags), bad))
    return tags


# 1. stats
for f in glob.glob('%s/%sstats/**/*kernel_stats.csv' % (out, SUB), recursive=True):
    shutil.copy(f, os.path.join(out, PFX + '_' + SUB + 'kernel_stats.csv'))
dur = collections.defaultdict(list)
all_rows = trace(SUB + 'stats')
tags = shapes_by_dispatch(SUB + 'stats', all_rows)
rows_stats = sorted(all_rows, key=lambda r: int(r['Start_Timestamp']))
adams = [i for i, r in enumerate(rows_stats) if 'adam_kernel' in r['Kernel_Name']]
STEPS = len(adams)
# bench.py re-times single recorded ops live after its last step (roofline.avg_us, conv_classes): not part of a step
tail = sum(1 for r in rows_stats[adams[-1] + 1:] if 'adam_tick' not in r['Kernel_Name'])
rows_stats = rows_stats[:adams[-1] + 2]
print('%d steps in the trace; %d kernel records behind the last step (live re-timing of single ops) left out' % (STEPS, tail))
for r in rows_stats:
    grid = int(r['Grid_Size_X']) * int(r['Grid_Size_Y']) * int(r['Grid_Size_Z'])
    wg = int(r['Workgroup_Size_X']) * int(r['Workgroup_Size_Y']) * int(r['Workgroup_Size_Z'])
    dur[(short(r['Kernel_Name']), str(grid), str(wg), tags.get(r['Dispatch_Id'], ''))].append(
        (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
with open(os.path.join(out, PFX + '_' + SUB + 'per_shape.csv'), 'w', newline='') as f:
    w = csv.writer(f)
    w.writerow(['kernel', 'grid_threads', 'workgroup', 'shape', 'calls_per_step', 'avg_us', 'min_us', 'ms_per_step'])
    for k, v in sorted(dur.items(), key=lambda kv: -sum(kv[1])):
        w.writerow([k[0], k[1], k[2], k[3], round(len(v) / STEPS, 1), round(sum(v) / len(v), 2), round(min(v), 2), round(sum(v) / STEPS / 1e3, 3)])
total = sum(sum(v) for v in dur.values()) / STEPS / 1e3
print('serialised kernel time per step: %.2f ms' % total)


# 2. counters
def counters(sub, name):
    agg = collections.defaultdict(list)
    files = glob.glob('%s/%s/**/*counter_collection.csv' % (out, sub), recursive=True)
    rows = [r for f in files for r in csv.DictReader(open(f))]
    # one row per (dispatch, counter): the alignment wants every dispatch once
    seen, uniq = set(), []
    for r in rows:
        if r['Dispatch_Id'] not in seen:
            seen.add(r['Dispatch_Id'])
            uniq.append(r)
    t = shapes_by_dispatch(sub, uniq)
    for r in rows:
        if r['Counter_Name'] == name:
            agg[(short(r['Kernel_Name']), r['Grid_Size'], t.get(r['Dispatch_Id'], ''))].append(float(r['Counter_Value']))
    return agg


fe, wr = counters(SUB + 'step_FETCH_SIZE', 'FETCH_SIZE'), counters(SUB + 'step_WRITE_SIZE', 'WRITE_SIZE')
with open(os.path.join(out, PFX + '_' + SUB + 'hbm_traffic.csv'), 'w', newline='') as f:
    w = csv.writer(f)
    w.writerow(['kernel', 'grid_threads', 'shape', 'launches_counted', 'FETCH_KB', 'WRITE_KB', 'hbm_MB_per_launch', 'avg_us', 'GB_per_s', 'frac_of_8TBs'])
    for k in sorted(set(fe) | set(wr), key=lambda k: -sum(fe.get(k, [0])) - sum(wr.get(k, [0]))):
        a, b = fe.get(k, []), wr.get(k, [])
        if not a or not b:
            continue
        fkb, wkb = sum(a) / len(a), sum(b) / len(b)
        byt = (2 * fkb + wkb) * 1024
        us = [v for kk, v in dur.items() if kk[0] == k[0] and kk[1] == k[1] and kk[3] == k[2]]
        us = sum(us[0]) / len(us[0]) if us else 0.0
        gbs = byt / us / 1e3 if us else 0.0
        w.writerow([k[0], k[1], k[2], len(a), round(fkb, 1), round(wkb, 1), round(byt / 1e6, 2), round(us, 2), round(gbs, 1), round(gbs / 8000.0, 3)])
if SUB:
    sys.exit(0)
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
