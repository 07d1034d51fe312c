"""The strip kernel of the student's 3x3 64 -> 64 convolutions (csrc/conv_c3.hip, round 6) on the MI355X: forward,
BatchNorm-backward data gradient, folded BN-backward apply (+ the materialised operand for the separate weight-gradient
launch) and pair launches, through the C ABI, against the CPU specification (oracle/plan_interp.py, bf16 storage) and bit for
bit against conv_pp (both walk the taps and channel steps in the same order and round at the same points)."""
import numpy as np
import pytest
import torch

from tests import test_kernels_gpu as tk
from tests.test_kernels_gpu import Bench, make_bn, rnd, tensor_stats, RS, TOL

pytestmark = pytest.mark.gpu
G = R = None


def setup_module(module):
    global G, R
    tk.setup_module(tk)
    G, R = tk.G, tk.R


# N, H, W, bn, stats, blocks   (C = K = 64; strips of 256 pixels = 256 / W rows)
FWD_CASES = [
    (2, 64, 64, 'train', True, 4),          # 32 strips over 4 blocks
    (2, 64, 64, 'train', True, 5),          # uneven strip ranges
    (1, 64, 64, 'eval', False, 3),          # eval-mode prologue, no statistics
    (3, 32, 32, 'train', True, 4),          # 8 rows per strip
    (5, 16, 16, 'train', True, 2),          # a strip = one whole image
    (2, 32, 32, None, True, 256),           # no prologue, more blocks than strips
    (1, 8, 32, 'train', True, 1),           # ONE strip: both halo rows outside the image
    (2, 16, 64, 'train', True, 3),
]


def _fwd(case, gen, bt):
    N, H, W, bn_mode, use_stats, blocks = case
    C = K = 64
    x_val = rnd(gen, N, H, W, C) + 0.3
    x = bt.act((N, H, W, C), x_val, 'x')
    w = bt.buf('wlp', (K, 3, 3, C), rnd(gen, K, 3, 3, C, scale=1.0 / np.sqrt(C * 9)))
    bias = bt.buf('param', (K,), 0.1 * rnd(gen, K))
    y = bt.act((N, H, W, K), None, 'y')
    bn = None
    if bn_mode:
        bn = make_bn(bt, gen, C, bn_mode)
        bn.count = N * H * W
        if bn_mode == 'train':
            bn.stats = bt.buf('stats', (RS, 2, C), tensor_stats(x_val.to(torch.bfloat16).float()))
    ostats = bt.buf('stats', (RS, 2, K), torch.zeros(RS, 2, K, dtype=torch.float64)) if use_stats else None
    op = G.Op('conv', x=x, w=w, wkey='w', bias=bias, bkey='b', residual=None, y=y, out_stats=ostats, bn=bn, epi='plain',
              epi_x=None, epi_bn=None, epi_stats=None, dims=(N, H, W, C, K, 3, 3, 1, 1, H, W))
    return op, y, ostats


@pytest.mark.parametrize('case', FWD_CASES)
def test_c3_forward(case):
    N, H, W, bn_mode, use_stats, blocks = case
    gen = torch.Generator().manual_seed(301 + sum(v for v in case if isinstance(v, int) and not isinstance(v, bool)))
    bt = Bench(1)
    op, y, ostats = _fwd(case, gen, bt)
    bt.realise().run([op], ('c3', blocks))
    assert bt.n_c3 == 1, 'the strip kernel did not take the launch'
    bt.compare(y, label='c3 y %s' % (case,), **TOL[1])
    if use_stats:
        bt.compare(ostats, atol=TOL[1]['atol'] * N * H * W, rtol=TOL[1]['rtol'], label='c3 out_stats')


@pytest.mark.parametrize('case', [(2, 64, 64, 'train', True, 5), (3, 32, 32, 'train', True, 4), (5, 16, 16, 'eval', True, 2)])
def test_c3_equals_conv_pp_bitwise(case):
    outs, stats = [], []
    for backend in (('pp', 8), ('c3', case[5])):
        g2 = torch.Generator().manual_seed(331)
        bt = Bench(1)
        op, y, ostats = _fwd(case, g2, bt)
        bt.realise().run([op], backend)
        outs.append(bt.gpu.view(y.buf).cpu().view(torch.int16).clone())
        stats.append(bt.gpu.stats_read(ostats).cpu().sum(0))
    assert torch.equal(outs[0], outs[1])
    scale = stats[0].abs().max(1, keepdim=True).values
    assert ((stats[0] - stats[1]).abs() <= 2e-6 * scale + 1e-9).all(), float(((stats[0] - stats[1]).abs() / scale).max())


BWD_CASES = [(2, 64, 64, 5, True), (4, 32, 32, 4, True), (8, 16, 16, 3, False), (1, 8, 32, 1, True), (2, 64, 64, 256, False)]


def _dgrad_ops(bt, gen, case, fold):
    N, H, W, blocks, bias = case
    C = K = 64
    x_val = rnd(gen, N, H, W, C)
    x = bt.act((N, H, W, C), x_val, 'x')                  # forward input of the convolution (pre BN+ReLU)
    wm = bt.buf('param', (K, 3, 3, C), rnd(gen, K, 3, 3, C, scale=1.0 / np.sqrt(C * 9)))
    wb = bt.buf('wlp', (C, 3, 3, K))
    dz = bt.act((N, H, W, C), None, 'dz')
    bn = make_bn(bt, gen, C, 'train')
    bn.count = N * H * W
    bn.stats = bt.buf('stats', (RS, 2, C), tensor_stats(x_val.to(torch.bfloat16).float()))
    bst = bt.buf('stats', (RS, 2, C), torch.zeros(RS, 2, C, dtype=torch.float64))
    dw = bt.buf('grad', (K, 3, 3, C), torch.zeros(K, 3, 3, C))
    db = bt.buf('grad', (K,), torch.zeros(K)) if bias else None
    out = dict(dz=dz, bst=bst, dw=dw, db=db)
    ops = [G.Op('wprep', entries=[{'w': wm, 'w_fwd': None, 'w_bwd': wb}])]
    if fold:
        u_val = rnd(gen, N, H, W, K)
        u = bt.act((N, H, W, K), u_val, 'u')
        g_val = rnd(gen, N, H, W, K, scale=0.1)
        g = bt.act((N, H, W, K), g_val, 'g')
        du = bt.act((N, H, W, K), None, 'du')
        bn2 = make_bn(bt, gen, K, 'train')
        bn2.count = N * H * W
        bn2.stats = bt.buf('stats', (RS, 2, K), tensor_stats(u_val.to(torch.bfloat16).float()))
        gq, uq = g_val.to(torch.bfloat16).float(), u_val.to(torch.bfloat16).float()
        mean, var = uq.mean((0, 1, 2)), uq.var((0, 1, 2), unbiased=False)
        xhat = (uq - mean) / torch.sqrt(var + 1e-5)
        sums = torch.zeros(RS, 2, K, dtype=torch.float64)
        sums[0, 0], sums[0, 1] = gq.double().sum((0, 1, 2)), (gq.double() * xhat.double()).sum((0, 1, 2))
        bst2 = bt.buf('stats', (RS, 2, K), sums)
        dgam, dbet = bt.buf('grad', (K,), torch.zeros(K)), bt.buf('grad', (K,), torch.zeros(K))
        ap = G.Op('ew', op='bn_bwd_apply', dims=(N, H, W, K), x=u, x2=None, dy=g, add=None, y=du, out_stats=None, bstats=bst2,
                  dgamma=dgam, dbeta=dbet, bn=bn2)
        ops.append(ap)
        dy = du
        out.update(du=du, dgam=dgam, dbet=dbet)
    else:
        dy = bt.act((N, H, W, K), rnd(gen, N, H, W, K, scale=0.1), 'dy')
    wg = G.Op('wgrad', x=x, dy=dy, dw=dw, dbias=db, bn=bn, dims=(N, H, W, C, K, 3, 3, 1, 1, H, W))
    dg = G.Op('conv', x=dy, w=wb, wkey='w', bias=None, bkey=None, residual=None, y=dz, out_stats=None, bn=None,
              epi='bnrelu_bwd', epi_x=x, epi_bn=bn, epi_stats=bst, dims=(N, H, W, K, C, 3, 3, 1, 1, H, W))
    if fold:
        dg.fold_apply, dg.fold_wgrad = ap, wg
    ops += [dg, wg]
    out.update(dg=dg)
    return ops, out


@pytest.mark.parametrize('fold', [False, True])
@pytest.mark.parametrize('case', BWD_CASES)
def test_c3_dgrad(case, fold):
    """BatchNorm-backward data gradient with / without the folded BN-backward apply on its operand; the weight gradient of the
    3x3 convolution is a separate launch that reads the operand the data gradient materialised (fold_out)."""
    N, H, W, blocks, bias = case
    gen = torch.Generator().manual_seed(351 + sum(case[:4]) + 2 * fold)
    bt = Bench(1)
    ops, o = _dgrad_ops(bt, gen, case, fold)
    bt.realise().run(ops, ('c3', blocks), partials=True)
    assert bt.n_c3 == 1, 'the strip kernel did not take the data gradient'
    if fold:
        assert bt.n_folded == 1 and getattr(o['dg'], 'fold_active', False), 'the BN-backward apply was not folded'
    bt.compare(o['dz'], label='c3 dgrad dz %s' % (case,), **TOL[1])
    bt.compare(o['bst'], atol=TOL[1]['atol'] * N * H * W, rtol=TOL[1]['rtol'], label='c3 dgrad bn sums')
    m = N * H * W
    tol = dict(atol=2e-2 + 2e-5 * m, rtol=3e-2)
    bt.compare(o['dw'], label='wgrad dw behind c3 %s' % (case,), **tol)
    if bias:
        bt.compare(o['db'], label='wgrad dbias behind c3', **tol)
    if fold:
        bt.compare(o['dgam'], atol=1e-3, rtol=1e-4, label='dgamma of the folded BN')
        bt.compare(o['dbet'], atol=1e-3, rtol=1e-4, label='dbeta of the folded BN')
        bt.compare(o['du'], label='materialised operand', **TOL[1])


@pytest.mark.parametrize('case', [(2, 64, 64, 5, True), (4, 32, 32, 3, True)])
def test_c3_dgrad_equals_conv_pp_bitwise(case):
    res = []
    for backend in (('pp', 8), ('c3', case[3])):
        gen = torch.Generator().manual_seed(371)
        bt = Bench(1)
        ops, o = _dgrad_ops(bt, gen, case, True)
        bt.realise().run(ops, backend, partials=True)
        res.append((bt.gpu.view(o['dz'].buf).cpu().view(torch.int16).clone(), bt.gpu.view(o['du'].buf).cpu().view(torch.int16).clone(),
                    bt.gpu.stats_read(o['bst']).cpu().sum(0)))
    assert torch.equal(res[0][0], res[1][0]) and torch.equal(res[0][1], res[1][1])
    a, b = res[0][2], res[1][2]
    assert ((a - b).abs() <= 5e-5 * a.abs().max() + 1e-9).all(), (float((a - b).abs().max()), float(a.abs().max()))


def test_c3_is_bit_repeatable():
    outs = []
    for rep in range(3):
        gen = torch.Generator().manual_seed(391)
        bt = Bench(1)
        ops, o = _dgrad_ops(bt, gen, (2, 64, 64, 5, True), True)
        bt.realise().run(ops, ('c3', 5), partials=True)
        outs.append([bt.gpu.view(o['dz'].buf).cpu().view(torch.int16).clone(), bt.gpu.stats_read(o['bst']).cpu().clone()])
    for r in outs[1:]:
        for a, b in zip(outs[0], r):
            assert torch.equal(a, b)


@pytest.mark.parametrize('case', [(2, 64, 64, 'train', 'plain', 6), (3, 32, 32, 'train', 'plain', 3), (2, 64, 64, None, 'bnrelu_bwd', 5)])
def test_c3_pair(case):
    """'conv2' op (up- / low-branch convolutions of an hourglass level, full and half resolution) as ONE launch."""
    N, H, W, bnm, epi, blocks = case
    C = K = 64
    gen = torch.Generator().manual_seed(411 + sum(v for v in case if isinstance(v, int)))
    b = Bench(1)
    subs = []
    rd = lambda t: t.to(torch.bfloat16).float()
    for (h, w) in ((H, W), (H // 2, W // 2)):
        x_val = rnd(gen, N, h, w, C) + 0.3
        x = b.act((N, h, w, C), x_val)
        y = b.act((N, h, w, K), torch.zeros(N, h, w, K))
        wt = b.buf('wlp', (K, 3, 3, C), rnd(gen, K, 3, 3, C, scale=(2.0 / (9 * C)) ** 0.5))
        bias = b.buf('param', (K,), 0.1 * rnd(gen, K)) if epi == 'plain' else None
        bn = None
        if bnm is not None:
            bn = make_bn(b, gen, C, bnm)
            bn.stats = b.buf('stats', (RS, 2, C), tensor_stats(rd(x_val)))
            bn.count = N * h * w
        kw = dict(epi='plain', epi_x=None, epi_bn=None, epi_stats=None,
                  out_stats=b.buf('stats', (RS, 2, K), torch.zeros(RS, 2, K, dtype=torch.float64)))
        if epi == 'bnrelu_bwd':
            xf = rnd(gen, N, h, w, K)
            ex = b.act((N, h, w, K), xf)
            ebn = make_bn(b, gen, K, 'train', 'ebn')
            ebn.stats = b.buf('stats', (RS, 2, K), tensor_stats(rd(xf)))
            ebn.count = N * h * w
            kw = dict(epi='bnrelu_bwd', epi_x=ex, epi_bn=ebn, out_stats=None,
                      epi_stats=b.buf('stats', (RS, 2, K), torch.zeros(RS, 2, K, dtype=torch.float64)))
        subs.append(G.Op('conv', x=x, w=wt, wkey='', bias=bias, bkey='', residual=None, y=y, bn=bn,
                         dims=(N, h, w, C, K, 3, 3, 1, 1, h, w), **kw))
    op = G.Op('conv2', a=subs[0], b=subs[1])
    b.realise()
    b.run([op], ('c3', blocks))
    assert b.n_c3 == 1, 'the pair did not go out as ONE launch of the strip kernel'
    for i, s in enumerate(subs):
        b.compare(s.y, label='c3 pair[%d] y %r' % (i, case), **TOL[1])
        st = s.out_stats if s.out_stats is not None else s.epi_stats
        b.compare(st, atol=TOL[1]['atol'] * N * s.dims[1] * s.dims[2], rtol=TOL[1]['rtol'], label='c3 pair[%d] stats %r' % (i, case))


def test_c3_declines_what_it_does_not_serve():
    """Other channel counts, a residual, 128-wide rows, maps whose height is not a whole number of strips: the dispatcher falls
    through to conv_pp / conv_tile with unchanged results."""
    for (N, H, W, C, K, res) in [(2, 16, 16, 32, 64, False), (2, 16, 16, 64, 64, True), (1, 4, 128, 64, 64, False), (1, 6, 32, 64, 64, False)]:
        gen = torch.Generator().manual_seed(9 + H + W + C)
        bt = Bench(1)
        x = bt.act((N, H, W, C), rnd(gen, N, H, W, C), 'x')
        w = bt.buf('wlp', (K, 3, 3, C), rnd(gen, K, 3, 3, C, scale=0.05))
        r = bt.act((N, H, W, K), rnd(gen, N, H, W, K), 'r') if res else None
        y = bt.act((N, H, W, K), None, 'y')
        op = G.Op('conv', x=x, w=w, wkey='w', bias=None, bkey=None, residual=r, y=y, out_stats=None, bn=None, epi='plain',
                  epi_x=None, epi_bn=None, epi_stats=None, dims=(N, H, W, C, K, 3, 3, 1, 1, H, W))
        bt.realise().run([op], ('c3', 4))
        assert bt.n_c3 == 0
        bt.compare(y, label='fallback y', **TOL[1])
