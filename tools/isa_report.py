#!/usr/bin/env python3
"""Static instruction report of the gfx950 kernels (no GPU needed: hipcc cross-compiles).

Round 5 found the big kernels of this repo bound by instruction ISSUE -- a wave issues about one instruction per 4 cycles and the waves
of a SIMD mostly alternate -- so the instruction count of a tile loop predicts its time, and it can be read off
the compiler's assembly on the build machine.  Two views:

    tools/isa_report.py blocks <unit> [kernel-substring]    per kernel: instructions, MFMAs, and its largest basic blocks (the hot loops)
    tools/isa_report.py branchy <unit> [<unit> ...]          kernels ranked by branch density (per-element control flow the compiler
                                                             made out of selects: how `bnrelu_bwd_r`'s 56-block loop body was found)

<unit> = a translation unit of csrc/ without the extension (conv_pp, wgrad3, elementwise, ...) or a path to an existing .s file.
"""
import os
import re
import subprocess
import sys
import tempfile
from collections import Counter

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, 'fast-human-pose-estimation.pytorch_amd', 'csrc')
FLAGS = ['--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-I' + os.path.join(ROOT, 'include'), '-I' + CSRC, '-munsafe-fp-atomics',
         '-Wno-unused-result', '-Wno-pass-failed', '--cuda-device-only', '-S']

_LABEL = re.compile(r'^(\.LBB[0-9_]+):')
_FUNC = re.compile(r'^(_Z\w+):')
_INSTR = re.compile(r'^\t([a-z][a-z0-9_]*)')


def assembly(unit):
    """Path of the device assembly of `unit` (compiled into a temporary directory unless `unit` is a .s file already)."""
    if unit.endswith('.s') and os.path.exists(unit):
        return unit
    src = os.path.join(CSRC, unit + '.hip')
    if not os.path.exists(src):
        raise SystemExit('isa_report: no such unit: %s' % src)
    out = os.path.join(tempfile.mkdtemp(prefix='isa_'), unit + '.s')
    subprocess.run(['hipcc'] + FLAGS + [src, '-o', out], check=True, stderr=subprocess.DEVNULL)
    return out


def demangle(names):
    try:
        out = subprocess.run(['c++filt'] + list(names), capture_output=True, text=True, check=True).stdout.strip().split('\n')
    except (OSError, subprocess.CalledProcessError):
        out = list(names)
    return [re.sub(r'\(anonymous namespace\)::', '', re.sub(r'^void ', '', n)) for n in out]


def kernels(lines):
    """[(mangled name, [(block label, [mnemonic, ...]), ...])] for every function of an assembly listing."""
    out = []
    k = 0
    while k < len(lines):
        m = _FUNC.match(lines[k])
        if not m:
            k += 1
            continue
        j = k + 1
        blocks, cur, lab = [], [], 'entry'
        while j < len(lines) and '.Lfunc_end' not in lines[j]:
            lm = _LABEL.match(lines[j])
            if lm:
                blocks.append((lab, cur))
                cur, lab = [], lm.group(1)
            else:
                im = _INSTR.match(lines[j])
                if im:
                    cur.append(im.group(1))
            j += 1
        blocks.append((lab, cur))
        out.append((m.group(1), blocks))
        k = j
    return out


def summary(blocks):
    ins = [x for _, b in blocks for x in b]
    return {
        'instructions': len(ins),
        'mfma': sum(1 for x in ins if x.startswith('v_mfma')),
        'blocks': len(blocks),
        'branches': sum(1 for x in ins if x.startswith('s_cbranch')),
        'saveexec': sum(1 for x in ins if 'saveexec' in x),
        'lds_reads': sum(1 for x in ins if x.startswith('ds_read') or x.startswith('ds_load')),
        'global_loads': sum(1 for x in ins if x.startswith('global_load') or x.startswith('buffer_load')),
    }


def cmd_blocks(unit, pattern=''):
    ks = kernels(open(assembly(unit)).read().split('\n'))
    names = demangle([n for n, _ in ks])
    for (_, blocks), name in zip(ks, names):
        if pattern and pattern not in name:
            continue
        s = summary(blocks)
        big = sorted(blocks, key=lambda b: -len(b[1]))[:4]
        print('%s\n    %d instructions, %d MFMA, %d LDS reads, %d global loads, %d blocks, %d branches' %
              (name[:110], s['instructions'], s['mfma'], s['lds_reads'], s['global_loads'], s['blocks'], s['branches']))
        for lab, b in big:
            c = Counter(b)
            print('    %-12s %4d instr, %3d MFMA | %s' % (lab, len(b), sum(1 for x in b if x.startswith('v_mfma')),
                                                         ' '.join('%s:%d' % kv for kv in c.most_common(6))))


def cmd_branchy(units):
    rows = []
    for u in units:
        ks = kernels(open(assembly(u)).read().split('\n'))
        for (n, blocks) in ks:
            s = summary(blocks)
            if s['instructions'] >= 64:
                rows.append((s['branches'] / s['instructions'], s, n, u))
    rows.sort(key=lambda r: -r[0])
    names = demangle([r[2] for r in rows[:40]])
    for (d, s, _, u), name in zip(rows[:40], names):
        print('%.3f  %5d instr %4d blocks %4d branches %4d saveexec  %-14s %s' % (d, s['instructions'], s['blocks'], s['branches'], s['saveexec'],
                                                                                 os.path.basename(u), name[:90]))


if __name__ == '__main__':
    if len(sys.argv) < 3 or sys.argv[1] not in ('blocks', 'branchy'):
        raise SystemExit(__doc__)
    if sys.argv[1] == 'blocks':
        cmd_blocks(sys.argv[2], sys.argv[3] if len(sys.argv) > 3 else '')
    else:
        cmd_branchy(sys.argv[2:])
