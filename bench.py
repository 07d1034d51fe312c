#!/usr/bin/env python
"""FPD train-step benchmark (BASELINE.json metric): images/s of
    teacher forward (hg S=8 F=256, eval BN) + student forward/backward (hg S=4 F=128, train BN)
    + fused pose/KD JointsMSELoss + [RCCL gradient all-reduce] + Adam
on synthetic 256x256 crops, batch 32 per GPU.

  python bench.py --gpus N --steps K --warmup W
N>1: one rank per GPU.  Started without a torch.distributed environment (no WORLD_SIZE), `--gpus N` launches itself as
`python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port <free> bench.py ...`;
started by torch.distributed.run (the driver's form) it reads RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* from the env.

Prints ONE JSON line on rank 0 (the last line of stdout).  `roofline` prices the dominant single-shape kernel of the step
-- the fused frozen Bottleneck of the teacher at 64x64 (17 launches/step: 9 alone, 8 paired with a 32x32 one; the largest
single (kernel, shape) entry of the rocprof trace) -- with its algorithmic conv FLOPs against the dense bf16 MFMA peak,
timed live with HIP events on the launch stream; `roofline.step` prices one whole step (79.478 GFLOP per image, SURVEY.md
section 8(d)) the same way; `roofline.conv_classes` lists the costliest convolution classes of the step, each timed from
the step's own plan (`--config hrnet`: its top entry IS the roofline kernel).
`cpu_baseline` times the CPU oracle (restatement of the reference loop) on a bounded sample of the same workload on the
host cores: thread count swept per configuration, both the reference-faithful variant (teacher graph retained,
function.py:120) and the teacher-under-no_grad variant, 3 timed steps each at cfg-1 (hg2x64, B=2) and the benchmark pair (B=8).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
# Before the HIP runtime loads (it is part of torch): a step keeps three streams busy (teacher, student chain, weight
# gradients) and RCCL adds its own; with ROCm's default of 4 hardware queues two of the hot streams can end up sharing
# one and serialise (measured: 14.4 -> 17.8 ms/step as soon as the process group exists).  See DESIGN.md section 4.
# With the RCCL process group in the process (WORLD_SIZE > 1 or --force-dist) 6 queues: 5 streams are active on the default
# data-parallel path, a 6th with FPD_ALLREDUCE_BUCKETS=1, and at 8 queues that 6th serialises every lane (26.7 ms/step, r03); the
# default path itself is 0.03 (r03) - 0.08 ms/step (r05, three interleaved pairs: 9.80 vs 9.72; plain step 9.69) faster at 6.
def _dist_run():
    if int(os.environ.get('WORLD_SIZE', '1')) > 1 or '--force-dist' in sys.argv:
        return True
    for i, t in enumerate(sys.argv):            # `python bench.py --gpus N` re-launches itself under torch.distributed.run: the children inherit this
        if t == '--gpus' and i + 1 < len(sys.argv) and sys.argv[i + 1].isdigit() and int(sys.argv[i + 1]) > 1:
            return True
        if t.startswith('--gpus=') and t[7:].isdigit() and int(t[7:]) > 1:
            return True
    return False


os.environ.setdefault('GPU_MAX_HW_QUEUES', '6' if _dist_run() else '8')

import torch  # noqa: E402

GFLOP_PER_IMAGE = {'hg4x128<-hg8x256': 79.478, 'w32<-w48@256x192': 77.211, 'w32<-w48@384x288': 173.725}       # SURVEY.md section 8(d), conv 2*MAC: t-fwd + s-fwd + s-bwd
PEAK_TFLOPS = {'bf16': 2500.0, 'fp32': 157.3}          # MI355X dense MFMA peaks (MI355X_MICROARCH.md)


class AD(dict):
    __getattr__ = dict.__getitem__


def make_cfg(feats, stacks, joints, dtype):
    return AD(MODEL=AD(NUM_JOINTS=joints, DTYPE=dtype, EXTRA=AD(NUM_FEATURES=feats, NUM_STACKS=stacks, NUM_BLOCKS=1)))


def _cpu_time(pair, batch, steps, no_grad, seed=0):
    """seconds/step of the CPU oracle (port of lib/core/function.py:114-147 over the restated hourglass)."""
    from oracle import fpd_ref, hourglass_ref
    (sf, ss), (tf, ts) = pair
    torch.manual_seed(seed)
    s_sd = fpd_ref.synth_state_dict(hourglass_ref.hourglass_keys(sf, ss, 16), 1)
    t_sd = fpd_ref.synth_state_dict(hourglass_ref.hourglass_keys(tf, ts, 16), 2)
    x, tg, tw = fpd_ref.synth_batch(100, batch, 16)
    adam = {}
    fpd_ref.fpd_step(s_sd, t_sd, ss, ts, x, tg, tw, 0.5, adam_state=adam, teacher_no_grad=no_grad)      # warm-up
    t0 = time.time()
    for _ in range(steps):
        fpd_ref.fpd_step(s_sd, t_sd, ss, ts, x, tg, tw, 0.5, adam_state=adam, teacher_no_grad=no_grad)
    return (time.time() - t0) / steps


def cpu_baseline():
    """the oracle on the GPU box's host cores.  The thread count is swept SEPARATELY for cfg-1 (hg2x64,
    B=2) and for the benchmark pair (hg4x128 <- hg8x256, B=8) -- one timed step per count after a warm-up, stopping when a
    count is clearly slower than the best so far: the small pair wants few threads, the big one more -- then each pair is
    timed with its best count for 3 steps after one warm-up, both with the teacher under no_grad and reference-faithful
    (teacher graph retained, function.py:120; fewer steps, and said so, if the ~100 s budget is exhausted).  `value` =
    images/s of the benchmark pair with the teacher under no_grad (the faster variant: the stronger baseline)."""
    ncpu = os.cpu_count() or 1
    cfg1, pair = ((64, 2), (64, 2)), ((128, 4), (256, 8))
    t_begin = time.time()

    def sweep(p, batch, counts, budget_s):
        res, t0 = {}, time.time()
        for nt in sorted({min(ncpu, n) for n in counts}):
            torch.set_num_threads(nt)
            res[nt] = round(batch / _cpu_time(p, batch, 1, True), 3)
            if res[nt] < 0.7 * max(res.values()) or time.time() - t0 > budget_s:
                break
        return res, max(res, key=res.get)
    variants = []

    def run(name, p, batch, steps, no_grad, nt):
        torch.set_num_threads(nt)
        dt = _cpu_time(p, batch, steps, no_grad)
        variants.append({'config': name, 'batch': batch, 'threads': nt, 'teacher': 'no_grad' if no_grad else 'graph retained (reference-faithful)',
                         'timed_steps': steps, 's_per_step': round(dt, 3), 'images_per_s': round(batch / dt, 3)})
        return batch / dt
    sweep1, best1 = sweep(cfg1, 2, (8, 16, 32, 64), 15)          # beyond 64 threads these small convolutions only lose
    run('cfg-1 hg2x64 <- hg2x64', cfg1, 2, 3, True, best1)
    run('cfg-1 hg2x64 <- hg2x64', cfg1, 2, 3, False, best1)
    big_b = 8
    sweep2, best2 = sweep(pair, big_b, (8, 16, 32, 64, 128), 35)
    v = run('cfg-2 hg4x128 <- hg8x256', pair, big_b, 3, True, best2)
    left = 100 - (time.time() - t_begin)
    per = variants[-1]['s_per_step'] * 1.5                               # the retained teacher graph costs about 1.4x
    steps_f = 3 if per * 4 < left else max(1, int(left / per) - 1)
    run('cfg-2 hg4x128 <- hg8x256', pair, big_b, steps_f, False, best2)
    return {'value': round(v, 3), 'unit': 'images/s', 'cores': best2, 'kind': 'port',
            'sample': 'same FPD pair (hg4x128 <- hg8x256, 256x256) at batch %d, 3 timed steps after 1 warm-up, torch CPU fp32 '
                      'oracle, teacher under no_grad, %d threads (best of a sweep on this pair; %d host CPUs)' % (big_b, best2, ncpu),
            'thread_sweep_cfg1_images_per_s': sweep1, 'thread_sweep_cfg2_images_per_s': sweep2, 'variants': variants}


def hr_extra(widths):
    """MODEL.EXTRA of experiments/fpd_coco/hrnet/*.yaml for the given branch widths (W32: 32..256, W48: 48..384)."""
    st = lambda n, m: dict(NUM_MODULES=m, NUM_BRANCHES=n, BLOCK='BASIC', NUM_BLOCKS=[4] * n, NUM_CHANNELS=widths[:n], FUSE_METHOD='SUM')
    return {'FINAL_CONV_KERNEL': 1, 'PRETRAINED_LAYERS': ['*'], 'STAGE2': st(2, 1), 'STAGE3': st(3, 4), 'STAGE4': st(4, 3)}


def cpu_baseline_hrnet():
    """The same loop of the CPU oracle (function.py:114-147) over the HRNet restatement (oracle/hrnet_ref.py): W32 student,
    W48 teacher, 256x192, J=17 at batch 2 -- 3 timed steps after one warm-up, teacher under no_grad and reference-faithful."""
    from oracle import fpd_ref, hrnet_ref
    ncpu = os.cpu_count() or 1
    ex_s, ex_t = hr_extra([32, 64, 128, 256]), hr_extra([48, 96, 192, 384])
    fs = lambda sd, x, train: [hrnet_ref.hrnet_forward(sd, ex_s, x, train)]
    ft = lambda sd, x, train: [hrnet_ref.hrnet_forward(sd, ex_t, x, train)]
    B = 2
    x, tg, tw = fpd_ref.synth_batch(100, B, 17, image_size=(192, 256), heatmap_size=(48, 64))

    def t_step(steps, no_grad):
        torch.manual_seed(0)
        s_sd = fpd_ref.synth_state_dict(hrnet_ref.hrnet_keys(ex_s, 17), 1)
        t_sd = fpd_ref.synth_state_dict(hrnet_ref.hrnet_keys(ex_t, 17), 2)
        adam = {}
        kw = dict(adam_state=adam, teacher_no_grad=no_grad, student_forward=fs, teacher_forward=ft)
        fpd_ref.fpd_step(s_sd, t_sd, 1, 1, x, tg, tw, 0.5, **kw)
        t0 = time.time()
        for _ in range(steps):
            fpd_ref.fpd_step(s_sd, t_sd, 1, 1, x, tg, tw, 0.5, **kw)
        return (time.time() - t0) / steps
    t_begin = time.time()
    sweep = {}
    for nt in sorted({min(ncpu, n) for n in (8, 16, 32)}):
        torch.set_num_threads(nt)
        sweep[nt] = round(B / t_step(1, True), 3)
        if time.time() - t_begin > 30:
            break
    best = max(sweep, key=sweep.get)
    torch.set_num_threads(best)
    variants = []
    for no_grad in (True, False):
        dt = t_step(3, no_grad)
        variants.append({'config': 'configs[3] shapes W32 <- W48 256x192', 'batch': B, 'timed_steps': 3, 's_per_step': round(dt, 3),
                         'teacher': 'no_grad' if no_grad else 'graph retained (reference-faithful)', 'images_per_s': round(B / dt, 3)})
    return {'value': variants[0]['images_per_s'], 'unit': 'images/s', 'cores': best, 'kind': 'port',
            'sample': 'same HRNet pair (W32 <- W48, 256x192, J=17) at batch %d, 3 timed steps after 1 warm-up, torch CPU fp32 oracle '
                      '(oracle/hrnet_ref.py), teacher under no_grad, %d threads (best of the sweep on %d host CPUs)' % (B, best, ncpu),
            'thread_sweep_images_per_s': sweep, 'variants': variants}


def time_plan_op(R, plan, k, launches):
    """Mean device time (us) of plan op `k` re-issued back to back on the current stream between two HIP events."""
    l, st = R.lib(), R.current_stream()
    for _ in range(2):
        plan.run_op(k)
    e0, e1 = l.fpd_event_create(), l.fpd_event_create()
    l.fpd_event_record(e0, st)
    for _ in range(launches):
        plan.run_op(k)
    l.fpd_event_record(e1, st)
    return l.fpd_event_elapsed_ms(e0, e1) / launches * 1e3


def conv_classes(step, R, peak_tflops, launches=12, top=12):
    """Every (role, kind, shape) class of convolution launch of the step -- student forward / data gradient / weight
    gradient, teacher forward -- with its launch count, one recorded representative timed live (the op of the step's own
    plan, same arguments, grid and slabs), and its roofline: algorithmic FLOPs and bytes (operand in + result out, bf16),
    bound = whichever of the HBM (8 TB/s) and MFMA limits is the longer time.  Sorted by launches x time."""
    groups = {}
    for role, inst in (('student', step.student),) + tuple(('teacher', t) for t in step.teachers[:1]):
        g = inst.g
        for phase, ops in (('fwd', g.fwd), ('bwd', g.bwd if inst.train else [])):
            for o in ops:
                if o.kind not in ('conv', 'wgrad'):
                    continue
                k = inst.op_index.get(id(o))
                if k is None or inst.plan.op_type(k) == R.OP_NOP:
                    continue
                kind = 'weight gradient' if o.kind == 'wgrad' else ('data gradient' if phase == 'bwd' else 'forward')
                e = groups.setdefault((role, kind, tuple(o.dims)), [inst.plan, k, 0, o])
                e[2] += 1
    out = []
    for (role, kind, dims), (plan, k, count, o) in groups.items():
        n, h, w, C, K, Rr, S, stride, pad, P, Q = dims
        flops = 2.0 * n * P * Q * C * K * Rr * S
        nbytes = 2.0 * (n * h * w * C + n * P * Q * K)
        if o.kind == 'conv' and getattr(o, 'residual', None) is not None:
            nbytes += 2.0 * n * P * Q * K
        # the shape tag the library's launch log gives this launch (csrc/api.hip set_op_tag): the key of profiles/*_hbm_traffic.csv
        tag = ('wgrad N=%d H=%d W=%d C=%d K=%d R=%d s=%d' % (n, h, w, C, K, Rr, stride) if o.kind == 'wgrad' else
               'conv N=%d H=%d W=%d C=%d K=%d R=%d s=%d %s' % (n, h, w, C, K, Rr, stride, 'dgrad' if kind == 'data gradient' else 'fwd'))
        out.append({'role': role, 'kind': kind, 'conv': '%dx%d %d->%d stride %d on %dx%dx%d' % (Rr, S, C, K, stride, n, h, w),
                    'launches_per_step': count, 'flops': flops, 'bytes': nbytes, 'shape_tag': tag, '_plan': plan, '_k': k})
    # time the classes that can matter (largest algorithmic cost first; a class whose roofline time is tiny can still be slow,
    # so the cut is generous)
    for e in out:
        e['us'] = time_plan_op(R, e.pop('_plan'), e.pop('_k'), launches)
        t_hbm, t_mfma = e['bytes'] / 8e12 * 1e6, e['flops'] / (peak_tflops * 1e12) * 1e6
        e['bound'] = 'hbm' if t_hbm >= t_mfma else 'mfma'
        e['roofline_us'] = round(max(t_hbm, t_mfma), 2)
        e['frac'] = round(max(t_hbm, t_mfma) / e['us'], 4)
        e['ms_per_step'] = round(e['us'] * e['launches_per_step'] * 1e-3, 3)
        e['us'] = round(e['us'], 2)
    out.sort(key=lambda e: -e['ms_per_step'])
    return out[:top], round(sum(e['ms_per_step'] for e in out), 3)


def pmc_traffic_of_class(shape_tag, sub):
    """HBM bytes per launch of a convolution class from the committed rocprofv3 --pmc passes of this round, keyed on the shape
    tag of the library's launch log (profiles/r06_<sub>hbm_traffic.csv, else an earlier round's, written by tools/profile_summarize.py); (None, why)
    if that table or row is absent."""
    import csv
    for pfx in ('r06', 'r05', 'r04', 'r04a'):
        path = os.path.join(ROOT, 'profiles', '%s_%shbm_traffic.csv' % (pfx, sub))
        if not os.path.exists(path):
            continue
        rows = [r for r in csv.DictReader(open(path)) if r['shape'].startswith(shape_tag)]
        if rows:
            r = max(rows, key=lambda r: int(r['launches_counted']))
            return float(r['hbm_MB_per_launch']) * 1e6, 'profiles/%s_%shbm_traffic.csv: %s, "%s" (%s launches counted)' % (
                pfx, sub, r['kernel'], r['shape'], r['launches_counted'])
        return None, 'profiles/%s_%shbm_traffic.csv has no row for "%s"' % (pfx, sub, shape_tag)
    return None, 'no PMC pass for this configuration under profiles/'


def dominant_kernel(step, R, launches=50):
    """Live timing of the dominant (kernel, shape): the teacher's first 64x64 fused Bottleneck, re-launched `launches`
    times back to back on the current stream between two HIP events.  Returns None if the teacher graph is unfused."""
    t = step.teacher
    ops = [s_ for o in t.g.fwd for s_ in ((o.a, o.b) if o.kind == 'bneck2' else (o,)) if s_.kind == 'bneck' and s_.dims[1] == 64]
    if not ops:
        return None
    op = ops[0]
    n_single = sum(1 for o in t.g.fwd if o.kind == 'bneck' and o.dims[1] == 64)
    n_paired = sum(1 for o in t.g.fwd if o.kind == 'bneck2' and (o.a.dims[1] == 64 or o.b.dims[1] == 64))
    n, h, w, c, p = op.dims
    plan = R.Plan()
    plan.add(*t.low.op(op))
    l = R.lib()
    st = R.current_stream()
    for _ in range(3):
        plan.run(0, 1)
    e0, e1 = l.fpd_event_create(), l.fpd_event_create()
    l.fpd_event_record(e0, st)
    for _ in range(launches):
        plan.run(0, 1)
    l.fpd_event_record(e1, st)
    us = l.fpd_event_elapsed_ms(e0, e1) / launches * 1e3
    flops = 2.0 * n * h * w * (c * p + 9 * p * p + p * c)
    return {'name': 'bneck_eval_kernel<128> N=%d %dx%d C=%d P=%d (teacher Bottleneck, conv1x1+conv3x3+conv1x1 fused)' % (n, h, w, c, p),
            'us': us, 'flops': flops, 'bytes_algorithmic': 2.0 * n * h * w * c * 2, 'launches': launches,
            'per_step': (n_single, n_paired), 'grid_cap': int(os.environ.get('FPD_BNECK_BLOCKS', '128'))}


def phase_times(step, R, n=10):
    """Where a step's time goes (in the driver-visible record, not only in DESIGN.md): every phase of the
    step's own plans run ALONE -- nothing beside it on the chip -- timed with the host clock around `n` back-to-back runs between
    device synchronisations; plus the weight-gradient lane's launches (every op of the student's backward that the schedule
    puts on the lane: weight gradients, slab reductions) re-issued one by one = the lane's serialised kernel time.  Runs after
    the timed region (the buffers hold stale values: timing only)."""
    s = step.student

    def timed(fn):
        fn()
        torch.cuda.synchronize()
        t0 = time.time()
        for _ in range(n):
            fn()
        torch.cuda.synchronize()
        return round((time.time() - t0) / n * 1e3, 3)
    out = {'teacher_alone_ms': timed(lambda: step.teachers[0].run('fwd')) if step.teachers else None}
    for ph, key in (('fwd', 'student_fwd_ms'), ('mid', 'student_loss_ms'), ('bwd', 'student_bwd_ms'), ('adam', 'student_adam_ms')):
        out[key] = timed(lambda ph=ph: s.run(ph))

    def chain():
        for ph in ('prep', 'fwd', 'mid', 'bwd', 'adam'):
            s.run(ph)
    out['student_chain_alone_ms'] = timed(chain)
    lane_us, lane_n = 0.0, 0
    for o in s.g.bwd:
        if getattr(o, 'lane', 0):
            k = s.op_index.get(id(o))
            if k is not None and s.plan.op_type(k) != R.OP_NOP:
                lane_us += time_plan_op(R, s.plan, k, 3)
                lane_n += 1
    out['lane_serialised_ms'] = round(lane_us * 1e-3, 3)
    out['lane_launches'] = lane_n
    out['note'] = ('each phase of the step\'s own plans alone on the chip, host clock around %d back-to-back runs; '
                   'lane_serialised_ms = every weight-gradient-lane launch of the backward re-issued 3x on its own (HIP events)' % n)
    return out


def parity_object(build_pair, E, student, teacher, batch, dev, H, W):
    """Accuracy of the build that was just timed (bf16), measured here on the bench batch against the fp32 parity build of
    the same path (the build pinned to the reference within 1e-4 by tests/test_model_gpu.py / test_fullsize_gpu.py): same
    initial student weights, same calibrated teacher, one un-pipelined FPD iteration each.  The reference-relative
    accuracy statement (never worse than 1.5x the reference itself at bf16, fp64 referee) is tests/test_bf16_parity_gpu.py."""
    x, tg, tw = batch
    B = x.shape[0]
    res = {}

    def one(dtype):
        torch.manual_seed(1)
        s, t = build_pair(dtype)
        s.load_state_dict(student, strict=True)
        t.load_state_dict(teacher, strict=True)
        s, t = s.to(dev), t.to(dev)
        st = E.FusedFPDStep(s.device_state(), s.cfg_hg, t.device_state(), t.cfg_hg, B, H, W, alpha=0.5)
        st.set_batch(x, tg, tw)
        st.teacher_async(x)
        g = st.student
        torch.cuda.current_stream().wait_event(st.ev_t[0])
        g.run('prep'); g.run('fwd'); g.run('mid'); g.run('bwd')
        torch.cuda.synchronize()
        out = {'tmap': st.tmap[0].float().cpu(), 'map': g.output_view(len(g.g.outputs) - 1).float().cpu(), 'loss': st.losses(),
               'grad': s.device_state().A.tensor('grad').float().cpu().clone()}
        del st, s, t
        torch.cuda.empty_cache()
        return out
    a, b = one('bf16'), one('fp32')

    def rl2(u, v):
        return float((u.double() - v.double()).norm() / v.double().norm())
    # the map / gradient distances between the two builds on THIS batch are chaos figures of a random-init
    # network (0.5 / 1.3 relative L2, the reference under autocast(bf16) deviates as much) and are no longer printed; what is
    # measured live is the agreement of the losses, and the accuracy statement proper is the trained pair at the timed
    # architecture, whose measured figures the GPU tests write to profiles/ (FPD_WRITE_PARITY_JSON) and this line quotes.
    res = {'live': {'checked_against': 'fp32 parity build of the same path on the bench batch (pinned to the reference <= 1e-4 by the -m gpu tests)',
                    'loss_rel_err': round(abs(a['loss'][2] - b['loss'][2]) / abs(b['loss'][2]), 6),
                    'pose_loss_bf16_fp32': [round(a['loss'][0], 6), round(b['loss'][0], 6)],
                    'kd_loss_bf16_fp32': [round(a['loss'][1], 6), round(b['loss'][1], 6)]}}
    for name in ('r06_parity_trained.json', 'r05_parity_trained.json'):
        path = os.path.join(ROOT, 'profiles', name)
        if os.path.exists(path):
            d = json.load(open(path))
            # the record names the library it was measured on; a line timed on another build says so instead of quoting it silently
            from fpd_amd import runtime as _R
            rec_sha, cur_sha = d.get('_library_sha16'), _R.lib_sha16()
            res['trained_pair_library'] = {'record': rec_sha, 'timed': cur_sha, 'same_build': rec_sha == cur_sha}
            res['trained_pair'] = {k: {'ours_vs_fp64': round(v['ours_vs_fp64'], 5), 'reference_at_bf16_vs_fp64': round(v['reference_at_bf16_vs_fp64'], 5),
                                       'ceiling': v['ceiling']} for k, v in d.items() if k.startswith('trained')}
            res['trained_pair_source'] = ('profiles/%s: relative L2 (losses: relative error) against the fp64 oracle, written by tests/test_fullsize_gpu.py::'
                                          'test_full_architecture_trained_pair_bf16_vs_fp64 (hg8x256 teacher / hg4x128 student trained 240 + 240 Adam steps '
                                          'on the blob task, B = 8, 256x256) and tests/test_bf16_parity_gpu.py (the committed reference-trained tiny pair) '
                                          'on an MI355X; "reference_at_bf16" = the reference\'s own arithmetic under torch.autocast(bfloat16)' % name)
            break
    return res


def free_port():
    import socket
    with socket.socket() as so:
        so.bind(('127.0.0.1', 0))
        return so.getsockname()[1]


def self_launch(n):
    """`python bench.py --gpus N` without a torch.distributed environment: re-exec under torch.distributed.run, one rank per
    GPU (the form the driver uses for N>1); rank 0's JSON line is the child's stdout, passed through."""
    import subprocess
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(n), '--master-addr', '127.0.0.1',
           '--master-port', str(free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')      # dmabuf IPC: RCCL across processes needs it on this driver
    raise SystemExit(subprocess.call(cmd, env=env))


def stub_main(args, world, rank):
    """Launch-path check without GPUs (tests/test_host_cpu.py, FPD_BENCH_STUB=1): gloo process group, a 1 ms stand-in
    step + the bucketed all-reduce hook on a CPU tensor, the same barrier / max-over-ranks timing and JSON contract.
    The line is labelled as a stub -- it is never a measurement."""
    import torch.distributed as dist
    from fpd_amd import dist as fdist
    dist.init_process_group('gloo', rank=rank, world_size=world)
    hook = fdist.make_allreduce(dist)
    g = torch.full((1024,), float(rank + 1))
    for _ in range(args.warmup):
        hook(g.clone(), [(0, 512, None), (512, 1024, None)])()
    dist.barrier()
    t0 = time.time()
    for _ in range(args.steps):
        time.sleep(1e-3)
        x = g.clone()
        hook(x, [(0, 512, None), (512, 1024, None)])()
    dist.barrier()
    t = torch.tensor([time.time() - t0], dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    ok = bool((x == world * (world + 1) / 2).all())
    if rank == 0:
        print(json.dumps({'metric': 'stub (launch-path check, not a measurement)', 'value': world * args.batch * args.steps / float(t),
                          'unit': 'images/s', 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
                          'ms_per_step': float(t) / args.steps * 1e3, 'higher_is_better': True, 'scaling': 'weak',
                          'vs_baseline': None, 'dtype': 'none', 'data': 'stub',
                          'config': {'workload': 'stub', 'ranks': dist.get_world_size(), 'allreduce_ok': ok}}), flush=True)
    dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--warmup', type=int, default=5)
    ap.add_argument('--batch', type=int, default=32, help='per-GPU batch')
    ap.add_argument('--dtype', default='bf16', choices=['bf16', 'fp32'])
    ap.add_argument('--backend', default='mfma', choices=['mfma', 'naive'])
    ap.add_argument('--config', default='hourglass', choices=['hourglass', 'hrnet', 'hrnet_fp8'],
                    help="hourglass = BASELINE configs[1] (the headline metric); hrnet = configs[3] shapes: HRNet-W32 student + "
                         "HRNet-W48 teacher, 256x192, J=17; hrnet_fp8 = configs[4] shapes: the same pair at 384x288 with the "
                         "student's forward convolutions on the fp8 matrix pipe (secondary lines: step-level roofline only)")
    ap.add_argument('--no-fp8', action='store_true', help='hrnet_fp8 shapes with bf16 forward convolutions (same-shape A/B of the fp8 path)')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-parity', action='store_true', help='skip the bf16-vs-fp32-build parity sub-object')
    ap.add_argument('--no-phase-times', action='store_true', help='skip the per-phase / lane timing after the timed region')
    ap.add_argument('--graphs', action='store_true',
                    help='replay each phase as a hipGraph instead of launching kernel by kernel (same GPU time; with more\n'
                         'than 4 hardware queues ROCm 7.2 graph replay of multi-stream phases is pathologically slow)')
    ap.add_argument('--no-graphs', action='store_true', help='(default; kept for older command lines)')
    ap.add_argument('--force-dist', action='store_true',
                    help='initialise the RCCL process group and use the all-reduce path even with one rank (path check)')
    args = ap.parse_args()

    if 'WORLD_SIZE' not in os.environ and args.gpus > 1:
        self_launch(args.gpus)                     # does not return
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if world != args.gpus:
        raise SystemExit('bench.py: --gpus %d but the launcher started %d ranks' % (args.gpus, world))
    if os.environ.get('FPD_BENCH_STUB'):
        return stub_main(args, world, rank)
    torch.cuda.set_device(local_rank)
    dev = torch.device('cuda', local_rank)
    dist = None
    use_dist = world > 1 or args.force_dist
    if use_dist:
        import torch.distributed as dist
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29533')
        dist.init_process_group('nccl', rank=rank, world_size=world, device_id=dev)

    from fpd_amd import executor as E, runtime as R, synth
    from fpd_amd.lib.models import hourglass
    from fpd_amd import dist as fdist
    R.lib()
    R.set_backend(R.BACKEND_NAIVE if args.backend == 'naive' else R.BACKEND_MFMA)

    hr = args.config in ('hrnet', 'hrnet_fp8')
    f8 = args.config == 'hrnet_fp8'
    B, J, H, W = (args.batch, 17, 384, 288) if f8 else ((args.batch, 17, 256, 192) if hr else (args.batch, 16, 256, 256))
    torch.manual_seed(1)                       # identical weights on every rank
    if hr:
        from fpd_amd.lib.config import _wrap
        from fpd_amd.lib.models import pose_hrnet

        def hr_cfg(w, dtype, weight_dtype=''):
            return _wrap({'MODEL': {'NAME': 'pose_hrnet', 'NUM_JOINTS': J, 'INIT_WEIGHTS': False, 'PRETRAINED': '', 'DTYPE': dtype,
                                    'WEIGHT_DTYPE': weight_dtype, 'EXTRA': hr_extra(w)}})

        def build_pair(dtype):
            s_ = pose_hrnet.get_pose_net(hr_cfg([32, 64, 128, 256], dtype, 'fp8' if (f8 and not args.no_fp8 and dtype == 'bf16') else ''), is_train=True)
            torch.manual_seed(2)
            return s_, pose_hrnet.get_pose_net(hr_cfg([48, 96, 192, 384], dtype), is_train=False)
        if f8:                                   # secondary line (parked path, README): step-level roofline only
            args.no_parity = args.no_cpu_baseline = True
    else:
        def build_pair(dtype):
            s_ = hourglass.get_pose_net(make_cfg(128, 4, J, dtype), is_train=True)
            torch.manual_seed(2)
            return s_, hourglass.get_pose_net(make_cfg(256, 8, J, dtype), is_train=False)
    student, teacher = build_pair(args.dtype)
    student, teacher = student.to(dev), teacher.to(dev)

    # synthetic teacher: calibrate BN running statistics once so eval-mode activations are sane
    x, tg, tw = synth.make_batch(1000 + rank, B, J, (W, H), (W // 4, H // 4))
    cal = E.GraphInstance(teacher.device_state(), teacher.cfg_hg, B, H, W, train=True).finalize()
    cal.image().copy_(x)
    cal.run('prep'); cal.run('fwd')
    cal.calibrate_running_stats()
    del cal
    torch.cuda.empty_cache()

    init_sd = None
    if rank == 0 and args.dtype == 'bf16' and not args.no_parity:     # snapshot for the parity sub-object (after the timed region)
        init_sd = ({k: v.detach().cpu().clone() for k, v in student.state_dict().items()},
                   {k: v.detach().cpu().clone() for k, v in teacher.state_dict().items()})
    step = E.FusedFPDStep(student.device_state(), student.cfg_hg, teacher.device_state(), teacher.cfg_hg, B, H, W,
                          alpha=0.5, lr=2.5e-4, world_size=world)
    step.set_batch(x, tg, tw)                  # data resident in HBM before timing
    allreduce = fdist.make_allreduce(dist) if use_dist else None
    if os.environ.get('FPD_FAKE_ALLREDUCE'):            # dev: deferred-Adam sequencing without any collective
        allreduce = lambda g, buckets=None: (lambda: None)
    if use_dist:
        fdist.broadcast_state(dist, student)
        fdist.broadcast_state(dist, teacher)

    def barrier():
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()
            torch.cuda.synchronize()

    step.run_pipelined(1, allreduce)                 # one eager step (sets one-time kernel attributes) before capture
    if args.graphs:
        step.enable_graphs()
    step.run_pipelined(args.warmup, allreduce)
    barrier()
    l = R.lib()
    ev0, ev1 = l.fpd_event_create(), l.fpd_event_create()
    st = R.current_stream()
    t0 = time.time()
    l.fpd_event_record(ev0, st)
    step.run_pipelined(args.steps, allreduce)      # K teacher forwards + K student steps, pipeline starts/ends empty
    l.fpd_event_record(ev1, st)
    host_busy_ms = (time.time() - t0) / args.steps * 1e3         # host time inside the timed region (includes back-pressure of full queues)
    barrier()
    wall = time.time() - t0
    ev_ms = l.fpd_event_elapsed_ms(ev0, ev1)
    pose, kd, loss = step.losses()
    # host launch path, measured where nothing can block it: ONE step enqueued into empty queues after the timed region
    # (every kernel launch, event record / wait of one teacher forward + one student step), clock stopped before any sync
    t1 = time.time()
    step.run_pipelined(1, allreduce)
    host_enqueue_ms = (time.time() - t1) * 1e3
    barrier()
    if use_dist:
        t = torch.tensor([wall], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        wall = float(t.item())
    ms_per_step = wall / args.steps * 1e3
    value = world * B * args.steps / wall
    gfl = GFLOP_PER_IMAGE['w32<-w48@384x288' if f8 else ('w32<-w48@256x192' if hr else 'hg4x128<-hg8x256')]
    flop_step = gfl * 1e9 * B
    achieved = flop_step / (ev_ms / args.steps * 1e-3) / 1e12
    peak = PEAK_TFLOPS[args.dtype]
    step_roof = {'achieved': round(achieved, 2), 'frac': round(achieved / peak, 4), 'unit': 'TFLOP/s',
                 'note': 'one replay of the whole FPD step plan on one GPU: %.3f TFLOP algorithmic conv work (%.3f '
                         'GFLOP/image x %d), HIP events on the launch stream: %.3f ms/step' % (flop_step / 1e12, gfl, B,
                                                                                              ev_ms / args.steps)}
    dom = dominant_kernel(step, R) if (rank == 0 and args.dtype == 'bf16' and not hr) else None
    classes = None
    if rank == 0 and args.dtype == 'bf16' and not args.graphs:
        top, total_ms = conv_classes(step, R, peak)
        classes = {'top': top, 'all_classes_ms_per_step_serialised': total_ms,
                   'note': 'un-paired convolution launches of the step by (role, kind, shape): one recorded op of the step\'s own '
                           'plan per class re-issued 12x back to back between HIP events after the timed region; frac = roofline '
                           'time (max of algorithmic bytes / 8 TB/s and FLOPs / dense MFMA peak) / measured time'}
    if dom is not None:
        # HBM bytes/launch from separate rocprofv3 --pmc passes (tools/pmc_bneck.sh), only if that file was measured at the
        # launch geometry timed here (grid cap): a file from another geometry is refused, not quoted
        traffic, traffic_note = None, 'no PMC file under profiles/'
        for name in ('r06_pmc_bneck64.json', 'r05_pmc_bneck64.json', 'r04_pmc_bneck64.json', 'r04a_pmc_bneck64.json', 'r03_pmc_bneck64.json', 'r02_pmc_bneck64.json'):
            pmc = os.path.join(ROOT, 'profiles', name)
            if os.path.exists(pmc):
                d = json.load(open(pmc))
                if d.get('grid_cap') == dom['grid_cap']:
                    traffic, traffic_note = d.get('hbm_bytes_per_launch'), 'profiles/%s (grid cap %d)' % (name, dom['grid_cap'])
                    break
                traffic_note = 'profiles/%s was measured at grid cap %s, this run uses %d: not quoted' % (name, d.get('grid_cap', 'unknown (r02: 147-160)'), dom['grid_cap'])
        k_ach = dom['flops'] / (dom['us'] * 1e-6) / 1e12
        roofline = {'bound': 'mfma', 'achieved': round(k_ach, 2), 'peak': peak, 'unit': 'TFLOP/s',
                    'frac': round(k_ach / peak, 4), 'traffic': traffic, 'kernel': dom['name'],
                    'avg_us': round(dom['us'], 2), 'flop_per_launch': dom['flops'],
                    'algorithmic_bytes_per_launch': dom['bytes_algorithmic'],
                    'launches_per_step': {'single': dom['per_step'][0], 'paired_with_a_32x32_one': dom['per_step'][1]},
                    'grid_cap': dom['grid_cap'], 'traffic_source': traffic_note,
                    'note': 'dominant (kernel, shape) of the step, timed AS LAUNCHED IN THE STEP (persistent grid capped at '
                            'FPD_BNECK_BLOCKS of 256 CUs so that the concurrent student chain finds free compute units; uncapped the '
                            'same kernel runs 86.5 us = 0.26 of peak, DESIGN.md section 5); %d back-to-back launches timed with HIP '
                            'events on the launch stream after the timed region; traffic = HBM bytes/launch from rocprofv3 --pmc '
                            'passes at the same grid cap (profiles/), null if absent' % dom['launches'],
                    'step': step_roof}
    elif classes is not None and classes['top']:
        # no fused-Bottleneck teacher here (HRNet): the dominant (kernel, shape) is the costliest convolution class of the step
        d = classes['top'][0]
        if d['bound'] == 'hbm':
            ach, pk, unit = d['bytes'] / (d['us'] * 1e-6) / 1e9, 8000.0, 'GB/s'
        else:
            ach, pk, unit = d['flops'] / (d['us'] * 1e-6) / 1e12, peak, 'TFLOP/s'
        traffic, tsrc = pmc_traffic_of_class(d['shape_tag'], 'hrnet_' if hr else '')
        roofline = {'bound': d['bound'], 'achieved': round(ach, 2), 'peak': pk, 'unit': unit, 'frac': round(ach / pk, 4),
                    'traffic': traffic, 'traffic_source': tsrc,
                    'kernel': '%s %s, %s' % (d['role'], d['kind'], d['conv']), 'avg_us': d['us'],
                    'flop_per_launch': d['flops'], 'algorithmic_bytes_per_launch': d['bytes'],
                    'launches_per_step': d['launches_per_step'],
                    'note': 'the convolution class with the largest launches x time of the step (conv_classes below), timed as '
                            'recorded in the step plan; bound chosen by which roofline time is longer', 'step': step_roof}
    else:
        roofline = {'bound': 'mfma', 'achieved': step_roof['achieved'], 'peak': peak, 'unit': 'TFLOP/s',
                    'frac': step_roof['frac'], 'traffic': None, 'note': step_roof['note']}
    if classes is not None:
        roofline['conv_classes'] = classes
    phases = phase_times(step, R) if (rank == 0 and world == 1 and not args.graphs and not args.no_phase_times) else None
    out = {
        'metric': ('images/sec FPD train step (HRNet-W32 student with fp8 forward convolutions, HRNet-W48 teacher) 384x288' if f8 else
                   'images/sec FPD train step (HRNet-W32 student, HRNet-W48 teacher) 256x192' if hr else
                   'images/sec FPD train step (4-stack HG student, 8-stack teacher) 256x256'),
        'value': round(value, 2), 'unit': 'images/s', 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
        'ms_per_step': round(ms_per_step, 3), 'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
        'dtype': args.dtype, 'data': 'synthetic',
        'config': {'workload': ('configs[4] shapes with bf16 forward convolutions (A/B of the fp8 path): HRNet-W32 student + HRNet-W48 '
                                'teacher, 384x288, J=17, batch %d/GPU, ' if (f8 and args.no_fp8) else
                                'configs[4] shapes: HRNet-W32 student (e4m3 weights + activations on v_mfma_f32_32x32x16_fp8_fp8 in the '
                                'forward convolutions) + bf16 HRNet-W48 teacher, 384x288, J=17, batch %d/GPU, ' if f8 else
                                'configs[3] shapes: HRNet-W32 student + HRNet-W48 teacher, 256x192, J=17, batch %d/GPU, ' if hr else
                                'configs[1]: hourglass student S=4 F=128 + teacher S=8 F=256, 256x256, batch %d/GPU, ') % B +
                               'fused FPD step incl. Adam, teacher forward one batch ahead on a 2nd stream%s' % (' + RCCL all-reduce' if use_dist else ''),
                   'global_batch': world * B, 'parallelism': 'dp%d' % world, 'backend': args.backend,
                   'launch': 'hipGraph replay per phase' if args.graphs else 'one native plan call per phase, kernels launched eagerly on 3 streams',
                   'host_enqueue_ms_per_step': round(host_enqueue_ms, 3),
                   'host_ms_per_step_inside_timed_region': round(host_busy_ms, 3),
                   'launches_per_step': step.launches_per_step(),
                   'ranks': world,
                   'loss_last_step': round(loss, 6), 'finite': bool(loss == loss and abs(loss) < 1e6)},
        'roofline': roofline,
    }
    if phases is not None:
        out['phase_times'] = phases
    if use_dist:
        out['config']['rccl_version'] = '.'.join(str(v) for v in torch.cuda.nccl.version())
        exp = getattr(allreduce, 'exposed_us', None)
        out['config']['allreduce_exposed_us'] = round(exp(), 1) if exp else None
        bucketed = os.environ.get('FPD_ALLREDUCE_BUCKETS', '0') == '1'
        out['config']['allreduce'] = ('one asynchronous RCCL all-reduce per gradient bucket from a side stream (%d buckets, %d fp32 elements in total), waited for before Adam' % (
            len(student.table.buckets), student.table.sizes['param']) if bucketed else
            'ONE asynchronous RCCL all-reduce of the flat gradient arena (%d fp32 elements) issued behind the backward, waited for before Adam '
            '(overlaps the next teacher forward); FPD_ALLREDUCE_BUCKETS=1 = one per bucket' % student.table.sizes['param'])
        out['config']['gpu_max_hw_queues'] = os.environ.get('GPU_MAX_HW_QUEUES')
    if rank == 0:
        print('[bench] timed region done: %.3f ms/step; parity / cpu baseline next' % ms_per_step, file=sys.stderr, flush=True)
    if init_sd is not None:
        del step
        torch.cuda.empty_cache()
        out['parity'] = parity_object(build_pair, E, init_sd[0], init_sd[1], (x, tg, tw), dev, H, W)
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        t_cpu = time.time()
        out['cpu_baseline'] = cpu_baseline_hrnet() if hr else cpu_baseline()
        print('[bench] cpu baseline took %.1f s' % (time.time() - t_cpu), file=sys.stderr, flush=True)
    # the JSON line must be the LAST line of stdout: RCCL prints a version banner through C stdio, which (block-buffered when
    # stdout is a pipe or file) would otherwise be flushed at process exit, i.e. after this line -- every rank empties its C
    # buffer before the final barrier, rank 0 prints behind it
    import ctypes
    sys.stdout.flush()
    ctypes.CDLL(None).fflush(None)
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        ctypes.CDLL(None).fflush(None)
        print(json.dumps(out), flush=True)


if __name__ == '__main__':
    main()
