"""Host-side evidence tooling (no GPU): the alignment of the library's launch log with a rocprofv3 kernel trace that keys every
row of profiles/*_per_shape.csv / *_hbm_traffic.csv on the tensor shape (tools/trace_align.py, used by tools/profile_summarize.py
and tools/trace_ab.py)."""
import os
import random

import pytest
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tools'))
import trace_align  # noqa: E402


def _log_line(expr, grid, block, op, tag):
    return '%s\t%d\t%d\t%d\t%s\n' % (expr, grid, block, op, tag)


def _trace_row(did, name, t0=0, t1=1000):
    return {'Dispatch_Id': str(did), 'Kernel_Name': name, 'Start_Timestamp': str(t0), 'End_Timestamp': str(t1)}


LAUNCHES = [   # (launch expression as FPD_LAUNCH logs it, the demangled name the tracer reports, shape tag)
    ('(conv_pp_kernel<R, C, KH, BWD, WG>)', 'void (anonymous namespace)::conv_pp_kernel<1, 128, 2, false, false>((anonymous namespace)::PPArgs)', 'conv N=32 H=64 W=64 C=128 K=64 R=1 s=1 fwd'),
    ('(conv_tile_kernel<T, TN, BK, ALLW>)', 'void (anonymous namespace)::conv_tile_kernel<unsigned short, 1, 64, true, false>(fpd_conv_t, (anonymous namespace)::TileGeo)', 'conv N=32 H=8 W=8 C=64 K=64 R=3 s=1 fwd'),
    ('(bneck_eval_kernel<P, DMA>)', 'void (anonymous namespace)::bneck_eval_kernel<128, true>(fpd_bneck_t, int, int, int)', 'bneck N=32 H=64 W=64 C=256 P=128'),
    ('(bneck_eval_kernel<P, DMA>)', 'void (anonymous namespace)::bneck_eval_kernel<128, true>(fpd_bneck_t, int, int, int)', 'bneck N=32 H=32 W=32 C=256 P=128'),
    ('adam_kernel', '(anonymous namespace)::adam_kernel(fpd_adam_t)', 'adam n=3290000'),
    ('(ew_kernel<T, OP>)', 'void (anonymous namespace)::ew_kernel<unsigned short, 2>(fpd_ew_t)', 'ew bn_bwd_apply N=32 H=64 W=64 C=128'),
]
FOREIGN = ['void at::native::vectorized_elementwise_kernel<4, at::native::FillFunctor<float>>()', '__amd_rocclr_copyBuffer', '__amd_rocclr_fillBufferAligned']


def _synthetic(nsteps=3, seed=0):
    """a trace with foreign kernels interleaved and its rows shuffled (csv row order is not host order), plus the matching log"""
    rng = random.Random(seed)
    log_lines, rows, want = [], [], {}
    did = 0
    for _ in range(nsteps):
        for expr, name, tag in LAUNCHES:
            for _ in range(rng.randint(0, 2)):                       # torch / runtime kernels between ours: not in the log
                did += 1
                rows.append(_trace_row(did, rng.choice(FOREIGN)))
            did += 1
            rows.append(_trace_row(did, name))
            want[str(did)] = tag
            log_lines.append(_log_line(expr, 256, 512, len(log_lines), tag))
    rng.shuffle(rows)
    return log_lines, rows, want


def test_base_name_strips_templates_and_parentheses():
    assert trace_align.base_name('(conv_pp_kernel<R, C, KH, BWD, WG>)') == 'conv_pp_kernel'
    assert trace_align.base_name('adam_kernel') == 'adam_kernel'
    assert trace_align.base_name('(wgrad_tile_kernel<T, R, TP, H4, SMALL>)') == 'wgrad_tile_kernel'


def test_log_aligns_with_trace_by_dispatch_order_and_separates_shapes_of_one_kernel():
    log_lines, rows, want = _synthetic()
    log = trace_align.parse_log(log_lines)
    tags, bad, n_trace, n_log = trace_align.align(log, rows)
    assert bad == 0 and n_trace == n_log == len(log_lines)
    assert tags == want
    # the two Bottleneck shapes share kernel name AND grid: only the log tells them apart
    bn = [tags[r['Dispatch_Id']] for r in rows if 'bneck_eval_kernel' in r['Kernel_Name']]
    assert sorted(set(bn)) == ['bneck N=32 H=32 W=32 C=256 P=128', 'bneck N=32 H=64 W=64 C=256 P=128']
    assert bn.count(bn[0]) == len(bn) // 2


def test_a_wrong_pairing_is_counted_not_keyed():
    log_lines, rows, want = _synthetic(nsteps=2, seed=3)
    log = trace_align.parse_log(log_lines)
    # the log claims a different kernel for its second launch (log and trace of different runs): that pairing is refused, and no
    # row is ever keyed on a tag whose kernel name disagrees with the trace
    bad_log = list(log)
    bad_log[1] = ('stem_fwd_mfma_kernel',) + bad_log[1][1:]
    tags, bad, _, _ = trace_align.align(bad_log, rows)
    assert bad == 1 and len(tags) == len(want) - 1
    assert all(want[k] == v for k, v in tags.items())


def test_truncated_trace_keys_the_common_prefix_only():
    log_lines, rows, want = _synthetic(nsteps=2, seed=5)
    log = trace_align.parse_log(log_lines)
    ours = trace_align.library_rows(rows, set(l[0] for l in log))
    keep = set(r['Dispatch_Id'] for r in ours[:7])
    cut = [r for r in rows if r['Dispatch_Id'] in keep or not any(n in r['Kernel_Name'] for n in set(l[0] for l in log))]
    tags, bad, n_trace, n_log = trace_align.align(log, cut)
    assert bad == 0 and n_trace == 7 and n_log == len(log)
    assert tags == {k: v for k, v in want.items() if k in keep}


def test_log_lines_without_a_shape_tag_or_too_short_are_tolerated():
    log = trace_align.parse_log(['adam_kernel\t1\t256\t7\n', 'garbage\n', '(ew_kernel<T, OP>)\t4\t256\t8\tew add N=1 H=2 W=2 C=16\n'])
    assert log == [('adam_kernel', 1, 256, '7', ''), ('ew_kernel', 4, 256, '8', 'ew add N=1 H=2 W=2 C=16')]



def test_strict_alignment_refuses_count_drift_and_geometry_mismatch():
    """a count difference between log and trace shifts every later pairing without a name mismatch as long as the
    kernel repeats (graph replays, truncated traces); the tables under profiles/ are built with strict=True, which raises.  Where
    the trace carries Grid_Size / Workgroup_Size the pairing is also held to the logged launch geometry."""
    log_lines, rows, want = _synthetic(nsteps=2, seed=7)
    log = trace_align.parse_log(log_lines)
    tags, bad, _, _ = trace_align.align(log, rows, strict=True)
    assert bad == 0 and tags == want
    with pytest.raises(ValueError):
        trace_align.align(log[:-1], rows, strict=True)                  # one launch missing from the log: drift
    geo = [dict(r, Grid_Size=str(256 * 512), Workgroup_Size='512') for r in rows]
    assert trace_align.align(log, geo, strict=True)[1] == 0
    wrong = [dict(r, Grid_Size=str(128 * 512), Workgroup_Size='512') if r['Dispatch_Id'] == sorted(want, key=int)[0] else r for r in geo]
    tags2, bad2, _, _ = trace_align.align(log, wrong)
    assert bad2 == 1 and len(tags2) == len(want) - 1
    with pytest.raises(ValueError):
        trace_align.align(log, wrong, strict=True)


@pytest.mark.parametrize('argv,env,want', [
    ([], {}, '8'),
    (['--gpus', '1'], {}, '8'),
    (['--force-dist'], {}, '6'),
    (['--gpus', '4'], {}, '6'),                      # re-launches itself under torch.distributed.run: the ranks inherit the value
    (['--gpus=2'], {}, '6'),
    (['--gpus', '1'], {'WORLD_SIZE': '8'}, '6'),     # a rank started by the driver's launcher
    ([], {'GPU_MAX_HW_QUEUES': '4'}, '4'),           # an explicit setting always wins
])
def test_bench_picks_the_hardware_queue_count_before_torch_loads(argv, env, want):
    """bench.py sets GPU_MAX_HW_QUEUES before the HIP runtime loads: 8 for the plain step, 6 whenever the RCCL process group will
    exist (profiles/r05_late_ab.txt calls 31-32)."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = ("import sys, os; sys.argv = ['bench.py'] + %r; os.environ.pop('GPU_MAX_HW_QUEUES', None); os.environ.pop('WORLD_SIZE', None); "
            "os.environ.update(%r); p = %r; src = open(p).read(); src = src[:src.index('import torch')]; "
            "exec(compile(src, p, 'exec'), {'__file__': p, '__name__': 'bench_head'}); print(os.environ['GPU_MAX_HW_QUEUES'])"
            % (argv, env, os.path.join(root, 'bench.py')))
    out = subprocess.run([sys.executable, '-c', code], capture_output=True, text=True, timeout=60)
    assert out.returncode == 0, out.stderr[-400:]
    assert out.stdout.strip() == want


def test_isa_report_splits_a_listing_into_kernels_and_basic_blocks():
    """tools/isa_report.py (static instruction counts: the issue-bound model of DESIGN.md section 8 makes them a time estimate)."""
    import isa_report
    listing = '''\t.text
_ZN1a6kernelEv:                          ; @_ZN1a6kernelEv
; %bb.0:
\ts_load_dwordx2 s[0:1], s[4:5], 0x0
\tv_mov_b32_e32 v1, 0
.LBB0_1:                                ; =>This Inner Loop Header
\tds_read_b128 v[2:5], v1
\tglobal_load_dwordx4 v[6:9], v1, s[0:1]
\tv_mfma_f32_32x32x16_bf16 v[10:25], v[2:5], v[6:9], v[10:25]
\tv_mfma_f32_32x32x16_bf16 v[10:25], v[2:5], v[6:9], v[10:25]
\ts_cbranch_scc1 .LBB0_1
; %bb.2:
\ts_and_saveexec_b64 s[2:3], vcc
\ts_cbranch_execz .LBB0_4
.LBB0_4:
\ts_endpgm
.Lfunc_end0:
\t.size\t_ZN1a6kernelEv, .Lfunc_end0-_ZN1a6kernelEv
_ZN1a5otherEv:
\ts_endpgm
.Lfunc_end1:
'''.split('\n')
    ks = isa_report.kernels(listing)
    assert [n for n, _ in ks] == ['_ZN1a6kernelEv', '_ZN1a5otherEv']
    blocks = ks[0][1]
    assert [lab for lab, _ in blocks] == ['entry', '.LBB0_1', '.LBB0_4']
    assert blocks[1][1] == ['ds_read_b128', 'global_load_dwordx4', 'v_mfma_f32_32x32x16_bf16', 'v_mfma_f32_32x32x16_bf16', 's_cbranch_scc1',
                            's_and_saveexec_b64', 's_cbranch_execz']
    s = isa_report.summary(blocks)
    assert s == {'instructions': 10, 'mfma': 2, 'blocks': 3, 'branches': 2, 'saveexec': 1, 'lds_reads': 1, 'global_loads': 1}
