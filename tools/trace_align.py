"""Keys rocprofv3 trace rows on the tensor shape of the launch that produced them.

The shape is not in the trace.  The library writes one line per kernel launch in HOST order (FPD_LAUNCH_LOG, csrc/common.h FPD_LAUNCH:
kernel expression, grid blocks, block threads, plan op index, shape tag); rocprofv3's Dispatch_Id is assigned in the same host order, so the
trace rows of the library's kernels sorted by Dispatch_Id line up 1:1 with the log.  Every pairing is verified by kernel name.
Used by tools/profile_summarize.py (the tables under profiles/) and tools/trace_ab.py (per-shape A/B of two libraries);
tests/test_tools_cpu.py holds it to synthetic traces."""
import re


def base_name(expr):
    """'(conv_pp_kernel<R, C, KH, BWD, WG>)' / 'adam_kernel' -> 'conv_pp_kernel' (what the trace's kernel name must contain)"""
    return re.sub(r'[(<].*', '', expr.strip().lstrip('(')).strip()


def parse_log(lines):
    """[(kernel base name, grid blocks, block threads, op index, shape tag)] in host order"""
    rows = []
    for line in lines:
        f = line.rstrip('\n').split('\t')
        if len(f) < 4:
            continue
        rows.append((base_name(f[0]), int(f[1]), int(f[2]), f[3], f[4] if len(f) > 4 else ''))
    return rows


def library_rows(rows, names):
    """the trace rows (dicts with Dispatch_Id, Kernel_Name) of kernels the log knows, in host order"""
    return [r for r in sorted(rows, key=lambda r: int(r['Dispatch_Id'])) if any(n in r['Kernel_Name'] for n in names)]


def _geometry_agrees(row, l):
    """the trace row's launch geometry against the log's (grid blocks, block threads), where the trace carries it: rocprofv3
    reports Grid_Size in THREADS (x) and Workgroup_Size; multi-dimensional grids / traces without these columns pass"""
    try:
        gs, ws = int(row['Grid_Size']), int(row['Workgroup_Size'])
    except (KeyError, ValueError):
        return True
    return ws == l[2] and gs == l[1] * l[2]


def align(log, rows, strict=False):
    """-> (Dispatch_Id -> shape tag, mismatches, library rows in the trace, logged launches).  Rows beyond the common prefix (a
    trace cut short, a log of a longer run) stay un-keyed; a pairing whose kernel names -- or, where the trace has the columns,
    (grid, workgroup) sizes -- disagree is counted and left un-keyed.  A COUNT difference means the two files are not of the
    same run or launches went past FPD_LAUNCH (a hipGraph replay logs its launches once, at capture, but traces them on every
    replay): every later pairing would be shifted by the drift without any name mismatch as long as the kernel
    repeats, so strict=True (what the tables under profiles/ are built with) raises instead of keying the common prefix."""
    names = set(l[0] for l in log)
    ours = library_rows(rows, names)
    if strict and len(ours) != len(log):
        raise ValueError('launch log and trace are not of the same run: %d trace rows of library kernels vs %d logged launches '
                         '(graph replay? truncated trace?)' % (len(ours), len(log)))
    tags, bad = {}, 0
    for r, l in zip(ours, log):
        if l[0] not in r['Kernel_Name'] or not _geometry_agrees(r, l):
            bad += 1
            continue
        tags[r['Dispatch_Id']] = l[4]
    if strict and bad:
        raise ValueError('%d pairings of launch log and trace disagree in kernel name or launch geometry' % bad)
    return tags, bad, len(ours), len(log)
