"""Host logic of the product (graph wiring, fused-BN algebra, reverse mode, memory planning) executed by the
CPU interpreter and compared with the reference outputs in tests/golden."""
import numpy as np
import pytest
import torch

from fpd_amd import graph as G
from oracle import hourglass_ref, plan_interp as PI
from tests import _cases, _interp_util as U


def build(name, train, which='s', wlp_is_master=True, **kw):
    c = _cases.CONFIGS[name]
    feats, stacks = c[which]
    # the product's layout: one gradient bucket per stack (lib/models/hourglass.py)
    table = G.ParamTable(hourglass_ref.hourglass_keys(feats, stacks, c['joints']), bucket_of=G.hourglass_bucket_of(stacks))
    g = G.HourglassGraph(table, feats, stacks, c['joints'], c['batch'], c['image'][1], c['image'][0], train,
                         wlp_is_master=wlp_is_master, **kw)
    return c, table, g


@pytest.mark.parametrize('name', ['tiny', 'cfg1'])
def test_teacher_eval_forward(name):
    c, table, g = build(name, train=False, which='t')
    gold = _cases.load_golden(name)
    _, t_sd = _cases.state_dicts(name, gold)
    act = G.plan_memory(g.fwd)
    A = U.make_arenas(g, table, act)
    U.load_params(A, table, t_sd)
    x, _, _ = _cases.batch(name)
    A.t['image'].copy_(x.reshape(-1))
    PI.run(A, g.fwd)
    out = A.view(g.outputs[-1].buf).permute(0, 3, 1, 2)
    _cases.assert_parity(out.numpy(), gold['toutput'], _cases.truth64(name)['toutput'].numpy(), 'teacher map')
    # liveness-based planning must beat "one buffer per tensor" by a wide margin for inference
    total = sum(o.y.numel for o in g.fwd if hasattr(o, 'y'))
    assert act < 0.25 * total


@pytest.mark.parametrize('name,master', [('tiny', True), ('tiny', False), ('cfg1', True)])
def test_student_train_step(name, master):
    c, table, g = build(name, train=True, which='s', wlp_is_master=master)
    gold = _cases.load_golden(name)
    s_sd, _ = _cases.state_dicts(name, gold)
    x, tg, tw = _cases.batch(name)
    B, J = c['batch'], c['joints']
    hh, hw = c['heat'][1], c['heat'][0]
    teacher = torch.from_numpy(gold['toutput']).permute(0, 2, 3, 1).contiguous()
    loss = G.Op('loss', outs=g.outputs, douts=g.out_grads, teacher=teacher, alpha=0.5, grad_scale=1.0, B=B, J=J,
                target_shape=(B, J, hh, hw), extra_in=list(g.outputs), extra_out=list(g.out_grads))
    ops = g.fwd + [loss] + g.bwd
    act = G.plan_memory(ops)
    A = U.make_arenas(g, table, act, extra={'target': tg.numel(), 'weight': tw.numel()})
    U.load_params(A, table, s_sd)
    A.t['image'].copy_(x.reshape(-1))
    A.t['target'].copy_(tg.reshape(-1))
    A.t['weight'].copy_(tw.reshape(-1))
    PI.run(A, [U.wprep_op(g, table)] + ops)
    tr = _cases.truth64(name)
    for i, o in enumerate(g.outputs):
        _cases.assert_parity(A.view(o.buf).permute(0, 3, 1, 2).numpy(), gold['output%d' % i],
                             tr['outputs'][i].numpy(), 'student map %d' % i)
    # the loss op was fed the reference's fp32 teacher map, so compare with the reference's fp32 losses
    assert abs(A.t['losses'][0].item() - float(gold['pose'])) < 1e-5
    assert abs(A.t['losses'][1].item() - float(gold['kd'])) < 1e-5
    flat = U.flat_grads_oihw(A, table)
    stride = int(gold['grad_stride'])
    names = [k for k in table.trainable_keys()]
    t64 = torch.cat([tr['grads'][k].reshape(-1) for k in names])[::stride].numpy()
    _cases.assert_parity(flat[::stride].numpy(), gold['grad_flat'], t64, 'student gradients', floor=2e-6, atol=1e-5)
    for k in table.entries:
        if 'running' in k:
            np.testing.assert_allclose(A.view(table[k]).numpy(), gold['s_after/' + k], rtol=1e-4, atol=1e-5)
        if k.endswith('num_batches_tracked'):
            assert A.view(table[k]).item() == 1


def test_memory_plan_no_live_overlap():
    """No two simultaneously-live tensors may share bytes (checked independently of plan_memory's allocator)."""
    c, table, g = build('tiny', train=True)
    ops = g.fwd + g.bwd
    G.plan_memory(ops)
    first, last, acts = {}, {}, {}
    for i, op in enumerate(ops):
        for a in op.acts_in() + op.acts_out():
            first.setdefault(id(a), i)
            last[id(a)] = i
            acts[id(a)] = a
    items = sorted(acts.values(), key=lambda a: a.buf.off)
    for i, a in enumerate(items):
        for b in items[i + 1:]:
            if b.buf.off >= a.buf.off + a.numel:
                break
            la = 10 ** 9 if a.persistent else last[id(a)]
            lb = 10 ** 9 if b.persistent else last[id(b)]
            assert la < first[id(b)] or lb < first[id(a)], (a.name, b.name)


def test_param_table_layout():
    keys = hourglass_ref.hourglass_keys(32, 2, 16)
    t = G.ParamTable(keys)
    assert t['conv1.weight'].shape == (8, 7, 7, 3)
    assert t['layer1.0.conv2.weight'].shape == (8, 3, 3, 8)
    offs = [(b.off, b.numel) for b in t.entries.values() if b.arena == 'param']
    for (o1, n1), (o2, _) in zip(offs, offs[1:]):
        assert o1 + n1 <= o2 and o2 % 4 == 0
    assert len(t.trainable_keys()) == sum(1 for k, _ in keys if 'running' not in k and 'tracked' not in k)


def test_seeded_backward_without_loss_op():
    """Autograd-compat path: d(out_s) are written before the backward list starts (no loss op in the plan).  The
    planner must keep them live from the start of backward -- regression test for seeds being clobbered."""
    c, table, g = build('tiny', train=True)
    gold = _cases.load_golden('tiny')
    s_sd, _ = _cases.state_dicts('tiny', gold)
    x, tg, tw = _cases.batch('tiny')
    act = G.plan_memory(g.fwd + g.bwd)
    A = U.make_arenas(g, table, act)
    U.load_params(A, table, s_sd)
    A.t['image'].copy_(x.reshape(-1))
    PI.run(A, [U.wprep_op(g, table)] + g.fwd)
    teacher = torch.from_numpy(gold['toutput']).permute(0, 2, 3, 1)
    cnt = float(tg.numel())
    w2 = (tw.reshape(2, 16) ** 2)[:, None, None, :]
    for o, d in zip(g.outputs, g.out_grads):
        p = A.view(o.buf)
        A.view(d.buf).copy_(w2 * (0.5 * (p - tg.permute(0, 2, 3, 1)) + 0.5 * (p - teacher)) / cnt)
    PI.run(A, g.bwd)
    flat = U.flat_grads_oihw(A, table)
    tr = _cases.truth64('tiny')
    t64 = torch.cat([tr['grads'][k].reshape(-1) for k in table.trainable_keys()]).numpy()
    _cases.assert_parity(flat.numpy(), gold['grad_flat'], t64, 'seeded gradients', floor=2e-6, atol=1e-5)


@pytest.mark.parametrize('delay,levels', [(0, 2), (8, 4), (8, 0)])
def test_multi_lane_schedule_is_sound(delay, levels):
    """Brute force: every pair of ops that touch overlapping physical memory, at least one of them writing, must be
    ordered (lane order + event waits) the way the sequential list orders them -- including overlaps that only exist
    because the memory planner reused a block or the IR accumulates a gradient in place."""
    from fpd_amd.schedule import PhaseSchedule
    c, table, g = build('tiny', train=True, wlp_is_master=False, lane_levels=levels, wgrad_batch=7)
    G.plan_memory(g.fwd + g.bwd, reuse_delay=delay)
    assert g.n_lanes == 1 + g.depth + G.WGRAD_LANES
    for phase in (list(g.fwd), [None] + [o for o in g.bwd if o.kind != 'seed'] + [None]):
        entries = [((o.lane or 0), o.accesses()) if o is not None else (0, None) for o in phase]
        sch = PhaseSchedule(entries, g.n_lanes)
        assert len({l for l, _ in entries}) >= 1 + levels            # the phase really is multi-lane
        recs = []                                                      # (arena, start, end, op, write)
        for i, (lane, acc) in enumerate(entries):
            if acc is None:
                for j in range(i):
                    assert sch.happens_before(j, i), ('barrier', j, i)
                for j in range(i + 1, len(entries)):
                    assert sch.happens_before(i, j), ('after barrier', i, j)
                continue
            for bufs, wr in ((acc[0], False), (acc[1], True)):
                for b in bufs:
                    recs.append((b.arena, b.off, b.off + b.numel, i, wr))
        checked = 0
        for x, (a1, s1, e1, i1, w1) in enumerate(recs):
            for (a2, s2, e2, i2, w2) in recs[x + 1:]:
                if i1 != i2 and a1 == a2 and s1 < e2 and s2 < e1 and (w1 or w2):
                    assert sch.happens_before(min(i1, i2), max(i1, i2)), (phase[i1].kind, i1, phase[i2].kind, i2, a1)
                    checked += 1
        assert checked > len(entries)
        for i, ws in enumerate(sch.waits):                              # waits point backwards, to other lanes
            assert all(w < i and sch.lanes[w] != sch.lanes[i] for w in ws)


@pytest.mark.parametrize('levels', [0, 2])
def test_lanes_follow_the_hourglass_structure(levels):
    c, table, g = build('tiny', train=True, lane_levels=levels)
    # hg.<s>.hg.<level>.0.*: up-branch of hourglass level <level>+1; only the `lane_levels` largest get a lane
    up = [o for o in g.fwd if o.kind == 'conv' and '.hg.' in o.wkey and o.wkey.split('.')[4] == '0'
          and int(o.wkey.split('.')[3]) + 1 > g.depth - levels]
    assert (up or not levels) and all(o.lane == int(o.wkey.split('.')[3]) + 1 for o in up)
    assert all(o.lane == 0 for o in g.fwd if o not in up)
    assert all(o.lane > g.depth for o in g.bwd if o.kind in ('wgrad', 'stem_wgrad'))
    # weight gradients are deferred into batches: the gradient tensors they read must never be written twice
    nwrites = {}
    for o in g.bwd:
        if o.kind in ('conv', 'ew'):
            nwrites[id(o.y)] = nwrites.get(id(o.y), 0) + 1
    wg = [o for o in g.bwd if o.kind == 'wgrad']
    assert wg and all(nwrites.get(id(o.dy), 0) <= 1 for o in wg)


def test_fused_bottleneck_graph_matches_unfused():
    """A frozen network built with fuse_bneck=True (one 'bneck' op per Bottleneck) computes what the three-conv graph
    computes: in fp32 the two differ only by the conv-bias-into-BN-shift folding (rounding noise)."""
    from oracle import fpd_ref
    F_, S_, J = 128, 1, 4
    keys = hourglass_ref.hourglass_keys(F_, S_, J)
    sd = fpd_ref.synth_state_dict(keys, 7)
    x, _, _ = fpd_ref.synth_batch(5, 1, J, image_size=(64, 64), heatmap_size=(16, 16))
    outs = []
    for fuse in (False, True):
        table = G.ParamTable(keys)
        g = G.HourglassGraph(table, F_, S_, J, 1, 64, 64, train=False, fuse_bneck=fuse)
        kinds = {o.kind for o in g.fwd}
        assert ('bneck' in kinds or 'bneck2' in kinds) == fuse
        act = G.plan_memory(g.fwd)
        A = U.make_arenas(g, table, act)
        U.load_params(A, table, sd)
        A.t['image'].copy_(x.reshape(-1))
        PI.run(A, g.fwd)
        outs.append(A.view(g.outputs[-1].buf).clone())
    fused_ops = [s_ for o in G.HourglassGraph(G.ParamTable(keys), F_, S_, J, 1, 64, 64, train=False, fuse_bneck=True).fwd
                 for s_ in ((o.a, o.b) if o.kind == 'bneck2' else (o,)) if s_.kind == 'bneck']      # incl. paired ones
    assert len(fused_ops) == 9 and {o.dims[2] for o in fused_ops} == {16, 8, 4}      # 2x2 is outside the kernel's domain
    err = float((outs[0] - outs[1]).abs().max())
    assert err < 1e-4 * max(1.0, float(outs[0].abs().max())), err
    # training graphs never fuse (train-mode BN needs batch statistics between the convs)
    gt = G.HourglassGraph(G.ParamTable(keys), F_, S_, J, 1, 64, 64, train=True, fuse_bneck=True)
    assert all(o.kind != 'bneck' for o in gt.fwd)


def test_fused_head_graph_matches_unfused():
    """F=256, J=16 frozen network: the 'head' op (fc -> BN+ReLU -> score -> fc_ + score_ + residuals in one launch) wired
    by the builder computes what the four-conv graph computes (fp32: rounding noise only), for inner stacks and the last."""
    from oracle import fpd_ref
    F_, S_, J = 256, 2, 16
    keys = hourglass_ref.hourglass_keys(F_, S_, J)
    sd = fpd_ref.synth_state_dict(keys, 9)
    x, _, _ = fpd_ref.synth_batch(6, 1, J, image_size=(64, 64), heatmap_size=(16, 16))
    outs = []
    for fuse in (False, True):
        table = G.ParamTable(keys)
        g = G.HourglassGraph(table, F_, S_, J, 1, 64, 64, train=False, fuse_bneck=fuse)
        heads = [o for o in g.fwd if o.kind == 'head']
        assert len(heads) == (S_ if fuse else 0)
        if fuse:
            assert heads[0].next is not None and heads[-1].next is None
        act = G.plan_memory(g.fwd)
        A = U.make_arenas(g, table, act)
        U.load_params(A, table, sd)
        A.t['image'].copy_(x.reshape(-1))
        PI.run(A, g.fwd)
        outs.append([A.view(o.buf).clone() for o in g.outputs])
    for a, b in zip(*outs):
        assert float((a - b).abs().max()) < 1e-4 * max(1.0, float(a.abs().max()))


def test_gradient_buckets_are_contiguous_and_close_in_backward_order():
    """graph.ParamTable lays the param/grad arena out bucket by bucket (last stack first, stem with stack 0); the backward
    list closes each bucket with a slab reduction + a 'grad_ready' marker on the weight-gradient lane, and under the
    multi-lane schedule that marker is ordered after EVERY writer of the bucket's gradient slice (data-parallel
    all-reduce of the slice starts there, executor.FusedFPDStep.grad_buckets)."""
    from fpd_amd.schedule import PhaseSchedule
    c, table, g = build('cfg1', train=True, wlp_is_master=False)
    S = c['s'][1]
    assert len(table.buckets) == S
    assert table.buckets[0][0] == 0 and table.buckets[-1][1] == table.sizes['param']
    for (a0, a1), (b0, b1) in zip(table.buckets, table.buckets[1:]):
        assert a1 == b0 and a0 < a1                                        # contiguous, non-empty, no gaps
    for k in table.trainable_keys():
        lo, hi = table.buckets[table.bucket[k]]
        assert lo <= table[k].off and table[k].off + table[k].numel <= hi
    assert table.bucket['conv1.weight'] == S - 1 and table.bucket['hg.%d.hg.0.0.0.conv1.weight' % (S - 1)] == 0
    assert table.bucket['fc_.0.weight'] == S - 1 and table.bucket['score.%d.weight' % (S - 1)] == 0
    # state_dict order is untouched by the placement
    assert [k for k, _ in table.keys] == [k for k, _ in hourglass_ref.hourglass_keys(c['s'][0], S, c['joints'])]
    G.plan_memory(g.fwd + g.bwd, reuse_delay=8)
    phase = [None] + [o for o in g.bwd if o.kind != 'seed']
    ready = [i for i, o in enumerate(phase) if o is not None and o.kind == 'grad_ready']
    assert [phase[i].bucket for i in ready] == list(range(S))
    sch = PhaseSchedule([((o.lane or 0), o.accesses()) if o is not None else (0, None) for o in phase], g.n_lanes)
    for i in ready:
        lo, hi = table.buckets[phase[i].bucket]
        for j, o in enumerate(phase):
            if o is None or o.kind == 'grad_ready':
                continue
            wr = o.accesses()[1]
            if any(b.arena == 'grad' and b.off < hi and b.off + b.numel > lo for b in wr):
                assert j < i and sch.happens_before(j, i), (o.kind, j, i)
    # every weight gradient with a slab is reduced by exactly one bucket reduction, in its own bucket
    red = [o for o in g.bwd if o.kind == 'wreduce']
    seen = [id(w) for o in red for w in o.wgrads]
    assert len(seen) == len(set(seen)) == sum(1 for o in g.bwd if o.kind in ('wgrad', 'stem_wgrad'))      # the stem's too: no atomics anywhere
    for o in red:
        assert all(table.bucket[w.dw.name[5:]] == o.bucket for w in o.wgrads)
