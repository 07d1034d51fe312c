"""The streaming 1x1 convolution (csrc/conv_c1.hip, round 6) on the MI355X: forward, BatchNorm-backward data gradient, folded
BN-backward apply, fused weight / bias gradient and pair launches, through the C ABI, against the CPU specification
(oracle/plan_interp.py, bf16 storage: the same rounding points) and -- where both kernels round at the same points -- bit for
bit against conv_pp / conv_tile.  Tolerances as in tests/test_kernels_gpu.py (bf16: abs / rel 3e-2 on O(1) values)."""
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


# N, H, W, C, K, bn, residual, stats, blocks  (N*H*W a multiple of 32; tiles = N*H*W / 32, rounds of 8 tiles)
FWD_CASES = [
    (2, 64, 64, 128, 64, 'train', False, True, 4),        # conv1 of a student Bottleneck: 256 tiles = 32 rounds over 4 blocks
    (2, 64, 64, 64, 128, 'train', True, True, 5),         # conv3 + residual, uneven round ranges
    (2, 64, 64, 128, 128, None, False, True, 16),         # fc: no prologue
    (2, 64, 64, 128, 128, 'train', True, True, 3),        # fc_ + residual
    (1, 128, 128, 32, 64, 'train', True, True, 7),        # layer1 conv3 + downsample residual @128^2
    (1, 128, 128, 32, 32, 'train', False, True, 6),
    (2, 32, 32, 64, 64, 'eval', False, False, 2),         # eval-mode prologue, no statistics
    (2, 32, 32, 64, 32, 'train', True, False, 3),
    (2, 32, 32, 128, 32, None, True, True, 2),
    (2, 32, 32, 32, 128, 'train', False, True, 256),      # more blocks than rounds
    (1, 4, 40, 128, 64, 'train', True, True, 2),          # 5 tiles: a ragged last round (3 waves of the block idle)
    (3, 4, 8, 64, 128, 'train', True, True, 1),           # 3 tiles in ONE round
    (1, 8, 52, 64, 64, 'train', False, True, 3),          # 13 tiles, 2 rounds over 3 blocks (an idle block)
    (2, 64, 64, 16, 128, None, True, True, 4),            # score_: C = J = 16 (one k-step) + residual
    (2, 32, 32, 16, 128, None, False, False, 3),
]


@pytest.mark.parametrize('case', FWD_CASES)
def test_c1_forward(case):
    N, H, W, C, K, bn_mode, use_res, use_stats, blocks = case
    gen = torch.Generator().manual_seed(101 + sum(v for v in case if isinstance(v, int) and not isinstance(v, bool)))
    bt = Bench(1)
    x_val = rnd(gen, N, H, W, C) + 0.3
    x = bt.act((N, H, W, C), x_val, 'x')
    w = bt.buf('wlp', (K, 1, 1, C), rnd(gen, K, 1, 1, C, scale=1.0 / np.sqrt(C)))
    bias = bt.buf('param', (K,), 0.1 * rnd(gen, K))
    res = bt.act((N, H, W, K), rnd(gen, N, H, W, K), 'res') if use_res else None
    y = bt.act((N, H, W, K), None, 'y')
    bn = None
    if bn_mode:
        bn = make_bn(bt, gen, C, bn_mode)
        bn.count = N * H * W
        if bn_mode == 'train':
            bn.stats = bt.buf('stats', (RS, 2, C), tensor_stats(x_val.to(torch.bfloat16).float()))
    ostats = bt.buf('stats', (RS, 2, K), torch.zeros(RS, 2, K, dtype=torch.float64)) if use_stats else None
    op = G.Op('conv', x=x, w=w, wkey='w', bias=bias, bkey='b', residual=res, y=y, out_stats=ostats, bn=bn, epi='plain',
              epi_x=None, epi_bn=None, epi_stats=None, dims=(N, H, W, C, K, 1, 1, 1, 0, H, W))
    bt.realise().run([op], ('c1', blocks))
    assert bt.n_c1 == 1, 'the streaming kernel did not take the launch'
    bt.compare(y, label='c1 y %s' % (case,), **TOL[1])
    if use_stats:
        bt.compare(ostats, atol=TOL[1]['atol'] * N * H * W, rtol=TOL[1]['rtol'], label='c1 out_stats')


@pytest.mark.parametrize('case', [(2, 64, 64, 128, 64, 'train', False, 5), (2, 64, 64, 64, 128, 'train', True, 8),
                                  (2, 64, 64, 128, 128, None, True, 3), (1, 128, 128, 32, 64, 'train', True, 7)])
def test_c1_equals_conv_pp_bitwise(case):
    """Both kernels round at the same points (fp32 accumulate over the channel steps in the same order, + residual, + bias, one
    rounding), so y must agree bit for bit; the statistics are the same exact sums of the same rounded values up to the
    fp32 grouping of the partial sums (1e-6 relative)."""
    N, H, W, C, K, bn_mode, use_res, blocks = case
    outs, stats = [], []
    for backend in (('pp', 8), ('c1', blocks)):
        g2 = torch.Generator().manual_seed(131)
        bt = Bench(1)
        x_val = rnd(g2, N, H, W, C) + 0.3
        x = bt.act((N, H, W, C), x_val, 'x')
        w = bt.buf('wlp', (K, 1, 1, C), rnd(g2, K, 1, 1, C, scale=1.0 / np.sqrt(C)))
        bias = bt.buf('param', (K,), 0.1 * rnd(g2, K))
        res = bt.act((N, H, W, K), rnd(g2, N, H, W, K), 'res') if use_res else None
        y = bt.act((N, H, W, K), None, 'y')
        bn = None
        if bn_mode:
            bn = make_bn(bt, g2, C, bn_mode)
            bn.count = N * H * W
            bn.stats = bt.buf('stats', (RS, 2, C), tensor_stats(x_val.to(torch.bfloat16).float()))
        ostats = bt.buf('stats', (RS, 2, K), torch.zeros(RS, 2, K, dtype=torch.float64))
        op = G.Op('conv', x=x, w=w, wkey='w', bias=bias, bkey='b', residual=res, y=y, out_stats=ostats, bn=bn, epi='plain',
                  epi_x=None, epi_bn=None, epi_stats=None, dims=(N, H, W, C, K, 1, 1, 1, 0, H, W))
        bt.realise().run([op], backend)
        outs.append(bt.gpu.view(y.buf).cpu().view(torch.int16).clone())
        stats.append(bt.gpu.stats_read(ostats).cpu().sum(0))
    assert torch.equal(outs[0], outs[1])
    scale = stats[0].abs().max(1, keepdim=True).values
    assert ((stats[0] - stats[1]).abs() <= 2e-6 * scale + 1e-9).all(), float(((stats[0] - stats[1]).abs() / scale).max())


# forward convolution C -> K (1x1): the data gradient has K input and C output channels.  (N, H, W, C, K, blocks, bias)
BWD_CASES = [(4, 64, 64, 128, 64, 8, True), (4, 64, 64, 64, 128, 8, True), (2, 32, 32, 128, 64, 3, False),
             (1, 4, 40, 64, 128, 2, True), (3, 4, 8, 128, 64, 1, True), (1, 8, 52, 64, 128, 3, True),
             (2, 64, 64, 64, 128, 256, False)]


def _dgrad_ops(bt, gen, case, fold, fuse):
    N, H, W, C, K, blocks, bias = case
    x_val = rnd(gen, N, H, W, C)
    x = bt.act((N, H, W, C), x_val, 'x')                  # forward input of the convolution (pre BN+ReLU)
    wm = bt.buf('param', (K, 1, 1, C), rnd(gen, K, 1, 1, C, scale=1.0 / np.sqrt(C)))
    wb = bt.buf('wlp', (C, 1, 1, K))
    dz = bt.act((N, H, W, C), None, 'dz')
    bn = make_bn(bt, gen, C, 'train')                     # BN in front of the convolution (mask + sums of the data gradient)
    bn.count = N * H * W
    bn.stats = bt.buf('stats', (RS, 2, C), tensor_stats(x_val.to(torch.bfloat16).float()))
    bst = bt.buf('stats', (RS, 2, C), torch.zeros(RS, 2, C, dtype=torch.float64))
    dw = bt.buf('grad', (K, 1, 1, C), torch.zeros(K, 1, 1, C))
    db = bt.buf('grad', (K,), torch.zeros(K)) if bias else None
    out = dict(dz=dz, bst=bst, dw=dw, db=db)
    ops = [G.Op('wprep', entries=[{'w': wm, 'w_fwd': None, 'w_bwd': wb}])]
    if fold:
        u_val = rnd(gen, N, H, W, K)
        u = bt.act((N, H, W, K), u_val, 'u')              # output of the convolution = input of the next BN
        g_val = rnd(gen, N, H, W, K, scale=0.1)
        g = bt.act((N, H, W, K), g_val, 'g')              # masked gradient that reached that BN
        du = bt.act((N, H, W, K), None, 'du')
        bn2 = make_bn(bt, gen, K, 'train')                # the BN whose backward is folded
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
    wg = G.Op('wgrad', x=x, dy=dy, dw=dw, dbias=db, bn=bn, dims=(N, H, W, C, K, 1, 1, 1, 0, H, W))
    dg = G.Op('conv', x=dy, w=wb, wkey='w', bias=None, bkey=None, residual=None, y=dz, out_stats=None, bn=None,
              epi='bnrelu_bwd', epi_x=x, epi_bn=bn, epi_stats=bst, dims=(N, H, W, K, C, 1, 1, 1, 0, H, W))
    if fold:
        dg.fold_apply, dg.fold_wgrad = ap, wg
    if fuse:
        dg.fused_wgrad = wg
    ops += [dg, wg]
    out.update(dg=dg)
    return ops, out


@pytest.mark.parametrize('fuse', [False, True])
@pytest.mark.parametrize('fold', [False, True])
@pytest.mark.parametrize('case', BWD_CASES)
def test_c1_dgrad(case, fold, fuse):
    """BatchNorm-backward data gradient (ReLU mask of epi_x + the two sums), with / without the folded BN-backward apply on
    its operand and with / without the forward convolution's weight / bias gradient formed in the same launch: everything
    downstream must match the specification of the un-fused, un-folded op list."""
    N, H, W, C, K, blocks, bias = case
    gen = torch.Generator().manual_seed(151 + sum(case[:6]) + 2 * fold + fuse)
    bt = Bench(1)
    ops, o = _dgrad_ops(bt, gen, case, fold, fuse)
    bt.realise().run(ops, ('c1', blocks), partials=True)
    # launches served: the data gradient, plus -- not fused -- the 1x1 weight gradient stays with its own kernel
    assert bt.n_c1 == 1, 'the streaming kernel did not take the data gradient'
    if fold:
        assert bt.n_folded == 1 and getattr(o['dg'], 'fold_active', False), 'the BN-backward apply was not folded'
    assert bt.n_fused == (1 if fuse else 0)
    bt.compare(o['dz'], label='c1 dgrad dz %s' % (case,), **TOL[1])
    bt.compare(o['bst'], atol=TOL[1]['atol'] * N * H * W, rtol=TOL[1]['rtol'], label='c1 dgrad bn sums')
    m = N * H * W
    tol = dict(atol=2e-2 + 2e-5 * m, rtol=3e-2)
    bt.compare(o['dw'], label='c1 wgrad dw %s' % (case,), **tol)
    if bias:
        bt.compare(o['db'], label='c1 wgrad dbias', **tol)
    if fold:
        bt.compare(o['dgam'], atol=1e-3, rtol=1e-4, label='dgamma of the folded BN')
        bt.compare(o['dbet'], atol=1e-3, rtol=1e-4, label='dbeta of the folded BN')
        if not fuse:                                      # the operand was materialised for the separate weight-gradient launch
            bt.compare(o['du'], label='materialised operand', **TOL[1])


@pytest.mark.parametrize('case', [(4, 64, 64, 128, 64, 8, True), (4, 64, 64, 64, 128, 5, True)])
def test_c1_dgrad_close_to_conv_pp(case):
    """Same launch through conv_pp and conv_c1: the masked gradient must agree bit for bit (both round acc once, the mask
    commutes with the rounding), sums and weight gradient to the fp32 grouping of their partial sums."""
    res = []
    for backend in (('pp', 8), ('c1', case[5])):
        gen = torch.Generator().manual_seed(171)
        bt = Bench(1)
        ops, o = _dgrad_ops(bt, gen, case, True, True)
        bt.realise().run(ops, backend, partials=True)
        res.append((bt.gpu.view(o['dz'].buf).cpu().view(torch.int16).clone(), bt.gpu.stats_read(o['bst']).cpu().sum(0),
                    bt.gpu.view(o['dw']).cpu().double().clone(), bt.gpu.view(o['db']).cpu().double().clone()))
    assert torch.equal(res[0][0], res[1][0])
    for i, nm in ((1, 'sums'), (2, 'dw'), (3, 'dbias')):
        a, b = res[0][i], res[1][i]
        assert ((a - b).abs() <= 5e-5 * a.abs().max() + 1e-9).all(), (nm, float((a - b).abs().max()), float(a.abs().max()))


@pytest.mark.parametrize('case', [(2, 64, 64, 128, 16, 4), (3, 16, 16, 128, 16, 2), (2, 32, 32, 128, 128, 3), (1, 64, 64, 128, 128, 5)])
def test_c1_dgrad_of_the_score_convolution(case):
    """Forward convolution 128 -> J = 16 (hourglass.py:136): its data gradient has 16 input channels (one k-step), the usual
    BatchNorm-backward epilogue and NO fused weight gradient (no 32 x 32 tile of dW exists): the separate launch must agree.
    The same for 128 -> 128 (fc_, hourglass.py:137): sixteen tiles of dW do not fit beside the tiles of eight waves."""
    N, H, W, C, K, blocks = case
    gen = torch.Generator().manual_seed(181 + sum(case))
    bt = Bench(1)
    ops, o = _dgrad_ops(bt, gen, (N, H, W, C, K, blocks, True), False, True)
    bt.realise().run(ops, ('c1', blocks), partials=True)
    assert bt.n_c1 == 1 and bt.n_fused == 0
    bt.compare(o['dz'], label='score dgrad dz %s' % (case,), **TOL[1])
    bt.compare(o['bst'], atol=TOL[1]['atol'] * N * H * W, rtol=TOL[1]['rtol'], label='score dgrad bn sums')
    tol = dict(atol=2e-2 + 2e-5 * N * H * W, rtol=3e-2)
    bt.compare(o['dw'], label='score wgrad dw', **tol)
    bt.compare(o['db'], label='score wgrad dbias', **tol)


def test_c1_dgrad_is_bit_repeatable():
    """No floating-point atomics: slabs in block order, exact statistics limbs -- identical bytes over repeated launches."""
    case = (4, 64, 64, 64, 128, 8, True)
    outs = []
    for rep in range(3):
        gen = torch.Generator().manual_seed(191)
        bt = Bench(1)
        ops, o = _dgrad_ops(bt, gen, case, True, True)
        bt.realise().run(ops, ('c1', 8), partials=True)
        outs.append([bt.gpu.view(o['dz'].buf).cpu().view(torch.int16).clone(), bt.gpu.stats_read(o['bst']).cpu().clone(),
                     bt.gpu.view(o['dw']).cpu().clone(), bt.gpu.view(o['db']).cpu().clone()])
    for r in outs[1:]:
        for a, b in zip(outs[0], r):
            assert torch.equal(a, b)


PAIR_CASES = [(2, 32, 32, 128, 64, 'train', 'plain', 6), (3, 16, 16, 64, 128, 'train', 'plain', 3),
              (2, 32, 32, 128, 64, None, 'bnrelu_bwd', 5), (2, 32, 32, 64, 128, None, 'bnrelu_bwd', 2)]


@pytest.mark.parametrize('case', PAIR_CASES)
def test_c1_pair(case):
    """'conv2' op (the up- / low-branch convolutions of an hourglass level in one launch) vs the two convolutions one after
    the other in the interpreter."""
    N, H, W, C, K, bnm, epi, blocks = case
    gen = torch.Generator().manual_seed(211 + sum(v for v in case if isinstance(v, int)))
    b = Bench(1)
    subs = []
    rd = lambda t: t.to(torch.bfloat16).float()
    for (h, w) in ((H, W), (H // 2, W // 2)):
        x_val = rnd(gen, N, h, w, C) + 0.3
        x = b.act((N, h, w, C), x_val)
        y = b.act((N, h, w, K), torch.zeros(N, h, w, K))
        wt = b.buf('wlp', (K, 1, 1, C), rnd(gen, K, 1, 1, C, scale=(2.0 / C) ** 0.5))
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
                         dims=(N, h, w, C, K, 1, 1, 1, 0, h, w), **kw))
    op = G.Op('conv2', a=subs[0], b=subs[1])
    b.realise()
    b.run([op], ('c1', blocks))
    assert b.n_c1 == 1, 'the pair did not go out as ONE launch of the streaming kernel'
    for i, s in enumerate(subs):
        b.compare(s.y, label='c1 pair[%d] y %r' % (i, case), **TOL[1])
        st = s.out_stats if s.out_stats is not None else s.epi_stats
        b.compare(st, atol=TOL[1]['atol'] * N * s.dims[1] * s.dims[2], rtol=TOL[1]['rtol'], label='c1 pair[%d] stats %r' % (i, case))


def test_c1_declines_what_it_does_not_serve():
    """3x3, fp32, K = 16, ragged N*H*W, an accumulate source on a data gradient: the dispatcher must fall through to the other
    kernels (launch counter unchanged), with the same results as without the streaming kernel."""
    gen = torch.Generator().manual_seed(7)
    bt = Bench(1)
    N, H, W, C, K = 1, 5, 7, 64, 64                       # 35 pixels: not a multiple of 32
    x = bt.act((N, H, W, C), rnd(gen, N, H, W, C), 'x')
    w = bt.buf('wlp', (K, 1, 1, C), rnd(gen, K, 1, 1, C, scale=0.1))
    y = bt.act((N, H, W, K), None, 'y')
    op = G.Op('conv', x=x, w=w, wkey='w', bias=None, bkey=None, residual=None, y=y, out_stats=None, bn=None, epi='plain',
              epi_x=None, epi_bn=None, epi_stats=None, dims=(N, H, W, C, K, 1, 1, 1, 0, H, W))
    bt.realise().run([op], ('c1', 4))
    assert bt.n_c1 == 0
    bt.compare(y, label='fallback y', **TOL[1])
