"""Exact-input parity: inputs on which every intermediate is a small dyadic rational (exactly representable in bf16 AND
fp32), so the bf16 build must agree with the fp32 specification BIT FOR BIT -- no rounding tolerance to hide a dropped
halo row, a wrong border redirect, a mis-indexed weight tap or a missing tile at small magnitude.  (The random-input
kernel tests in test_kernels_gpu.py allow bf16 rounding noise of 3e-2; this file closes that gap, VERDICT r1 weak #4.)

Construction: activations in {-1, -0.5, 0, 0.5, 1}; every output channel of a convolution has TWO non-zero weights
(values +-1, +-0.5) at random (tap, input channel) positions -- the convolution becomes a signed sum of two shifted input
planes, which exercises every tap, halo row, border column and channel chunk while all partial sums stay below 2^8 in
magnitude with <= 8 significant bits.  The checker is the CPU interpreter (oracle/plan_interp.py) in fp32."""
import numpy as np
import pytest
import torch

from oracle import plan_interp as PI
from tests.test_kernels_gpu import Bench, RS, setup_module as _setup  # noqa: F401

pytestmark = pytest.mark.gpu
G = None


def setup_module(module):
    from tests import test_kernels_gpu as T
    T.setup_module(T)
    global G
    G = T.G


def small(gen, *shape, density=0.6):
    v = torch.randint(-2, 3, shape, generator=gen).float() * 0.5
    return v * (torch.rand(shape, generator=gen) < density).float()


def sparse_weights(gen, K, R, C, per_out=2):
    w = torch.zeros(K, R, R, C)
    for k in range(K):
        for _ in range(per_out):
            r, s, c = (int(torch.randint(0, n, (1,), generator=gen)) for n in (R, R, C))
            w[k, r, s, c] = float(torch.randint(0, 2, (1,), generator=gen) * 2 - 1) * (0.5 if torch.rand(1, generator=gen) < 0.3 else 1.0)
    return w


def exact_equal(bench, b, label):
    b = b.buf if hasattr(b, 'buf') else b
    c = bench.cpu.view(b).double()
    if b.arena == 'stats':                  # device: exact integer limbs (include/fpd_amd.h fpd_stat_t), decoded per replica
        c, g = c.sum(0), bench.gpu.stats_read(b).cpu().sum(0)
    else:
        g = bench.gpu.view(b).cpu().double()
    bad = c != g
    assert not bad.any(), '%s: %d/%d elements differ (max |diff| %.3e), first at %s' % (
        label, int(bad.sum()), bad.numel(), float((c - g).abs().max()), [int(i) for i in torch.nonzero(bad)[0]])
    assert c.abs().max() > 0, label + ': degenerate (all-zero) case'


CONV = [  # N, H, W, C, K, R
    (2, 16, 16, 64, 64, 3), (1, 64, 64, 64, 64, 3), (3, 4, 4, 64, 64, 3), (5, 8, 8, 128, 128, 3), (2, 32, 32, 128, 64, 1),
    (2, 8, 8, 64, 128, 1), (1, 128, 128, 32, 32, 3), (2, 16, 16, 256, 16, 1), (2, 16, 16, 16, 256, 1), (1, 6, 5, 16, 32, 3),
    # HRNet map widths (row tiles of 96 / 120 / 126 pixels, tiles straddling image boundaries)
    (2, 32, 24, 64, 64, 3), (1, 64, 48, 32, 32, 3), (3, 8, 6, 128, 128, 3), (2, 16, 12, 64, 128, 1), (5, 7, 6, 384, 96, 3),
]


@pytest.mark.parametrize('backend', [0, 1, 2])
@pytest.mark.parametrize('dtype', [0, 1])
@pytest.mark.parametrize('case', CONV)
def test_conv_forward_exact(case, dtype, backend):
    """conv + bias + residual + output statistics; also used as the data gradient (same kernel, flipped weights)."""
    N, H, W, C, K, Rr = case
    pad = (Rr - 1) // 2
    gen = torch.Generator().manual_seed(100 + sum(case))
    bt = Bench(dtype)
    x = bt.act((N, H, W, C), small(gen, N, H, W, C), 'x')
    w = bt.buf('wlp', (K, Rr, Rr, C), sparse_weights(gen, K, Rr, C))
    bias = bt.buf('param', (K,), small(gen, K))
    res = bt.act((N, H, W, K), small(gen, N, H, W, K), 'res')
    y = bt.act((N, H, W, K), None, 'y')
    ostats = bt.buf('stats', (RS, 2, K), torch.zeros(RS, 2, K, dtype=torch.float64))
    op = G.Op('conv', x=x, w=w, wkey='w', bias=bias, bkey='b', residual=res, y=y, out_stats=ostats, bn=None, epi='plain',
              epi_x=None, epi_bn=None, epi_stats=None, dims=(N, H, W, C, K, Rr, Rr, 1, pad, H, W))
    bt.realise().run([op], backend)
    exact_equal(bt, y, 'conv y %s dtype %d backend %d' % (case, dtype, backend))
    exact_equal(bt, ostats, 'conv out_stats %s' % (case,))                 # sums of small dyadic values: exact in fp64


@pytest.mark.parametrize('dtype', [0, 1])
@pytest.mark.parametrize('case', [(2, 32, 32, 64, 64, 3), (2, 16, 16, 128, 64, 1), (3, 8, 8, 64, 128, 1), (2, 64, 64, 64, 64, 3)])
def test_conv_pair_exact(case, dtype):
    """Two convolutions in one launch (up-branch at full, low-branch at half resolution)."""
    N, H, W, C, K, Rr = case
    pad = (Rr - 1) // 2
    gen = torch.Generator().manual_seed(300 + sum(case))
    bt = Bench(dtype)
    subs = []
    for (h, w_) in ((H, W), (H // 2, W // 2)):
        x = bt.act((N, h, w_, C), small(gen, N, h, w_, C), 'x')
        w = bt.buf('wlp', (K, Rr, Rr, C), sparse_weights(gen, K, Rr, C))
        bias = bt.buf('param', (K,), small(gen, K))
        y = bt.act((N, h, w_, K), None, 'y')
        subs.append(G.Op('conv', x=x, w=w, wkey='w', bias=bias, bkey='b', residual=None, y=y, out_stats=None, bn=None, epi='plain',
                         epi_x=None, epi_bn=None, epi_stats=None, dims=(N, h, w_, C, K, Rr, Rr, 1, pad, h, w_)))
    bt.realise().run([G.Op('conv2', a=subs[0], b=subs[1])], 0)
    exact_equal(bt, subs[0].y, 'pair a %s' % (case,))
    exact_equal(bt, subs[1].y, 'pair b %s' % (case,))


@pytest.mark.parametrize('backend', [0, 1, 2, 'partials'])
@pytest.mark.parametrize('dtype', [0, 1])
@pytest.mark.parametrize('case', [(2, 16, 16, 64, 64, 3), (4, 32, 32, 64, 64, 3), (2, 8, 8, 128, 64, 1), (32, 4, 4, 64, 64, 3),
                                  (1, 64, 64, 64, 128, 1), (2, 64, 64, 64, 64, 3), (2, 64, 48, 32, 32, 3), (3, 16, 48, 64, 64, 1),
                                  (2, 32, 24, 64, 64, 3)])
def test_conv_wgrad_exact(case, dtype, backend):
    """dw = sum over pixels of dy (x) x over sparse small-integer planes: integer sums far below 2^24 -> exact in fp32."""
    N, H, W, C, K, Rr = case
    pad = (Rr - 1) // 2
    gen = torch.Generator().manual_seed(500 + sum(case))
    bt = Bench(dtype)
    x = bt.act((N, H, W, C), small(gen, N, H, W, C, density=0.3), 'x')
    dy = bt.act((N, H, W, K), small(gen, N, H, W, K, density=0.3), 'dy')
    dw = bt.buf('grad', (K, Rr, Rr, C), torch.zeros(K, Rr, Rr, C))
    db = bt.buf('grad', (K,), torch.zeros(K))
    op = G.Op('wgrad', x=x, dy=dy, dw=dw, dbias=db, bn=None, dims=(N, H, W, C, K, Rr, Rr, 1, pad, H, W))
    if backend == 'partials':
        bt.realise().run([op], 0, partials=True)
    else:
        bt.realise().run([op], backend)
    exact_equal(bt, dw, 'wgrad dw %s' % (case,))
    exact_equal(bt, db, 'wgrad dbias %s' % (case,))


def unit_bn(bt, C, name):
    """Eval-mode BN whose folded scale/shift are exactly (1, 0) in fp32: gamma 1, beta 0, mean 0, var 1 - eps."""
    gamma, beta = bt.buf('param', (C,), torch.ones(C)), bt.buf('param', (C,), torch.zeros(C))
    rmean, rvar = bt.buf('rstat', (C,), torch.zeros(C)), bt.buf('rstat', (C,), torch.full((C,), 1.0 - 1e-5))
    return G.BN(name, 'eval', C, gamma, beta, rmean, rvar, bt.buf('nbt', ()), relu=True)


@pytest.mark.parametrize('fold', [False, True])
@pytest.mark.parametrize('case', [(2, 64, 64, 64), (2, 64, 64, 128), (3, 32, 32, 128), (5, 16, 16, 64), (7, 8, 8, 128), (33, 4, 4, 64)])
def test_bottleneck_fused_exact(case, fold):
    """Whole fused Bottleneck (conv1 over tile + halo, a2 image in LDS, 9 taps, conv3, residual) with unit BNs on sparse
    dyadic inputs: the bf16 kernel equals the fp32 specification bit for bit -- every halo row / border column / tile
    seam of the LDS image is covered exactly."""
    N, H, W, P = case
    C = 2 * P
    gen = torch.Generator().manual_seed(700 + sum(case))
    b = Bench(1)
    x = b.act((N, H, W, C), small(gen, N, H, W, C))
    y = b.act((N, H, W, C), torch.zeros(N, H, W, C))
    w1 = b.buf('wlp', (P, 1, 1, C), sparse_weights(gen, P, 1, C))
    w2 = b.buf('wlp', (P, 3, 3, P), sparse_weights(gen, P, 3, P))
    w3 = b.buf('wlp', (C, 1, 1, P), sparse_weights(gen, C, 1, P))
    b1, b2, b3 = (b.buf('param', (n,), small(gen, n)) for n in (P, P, C))
    op = G.Op('bneck', x=x, y=y, dims=(N, H, W, C, P), w1=w1, b1=b1, w2=w2, b2=b2, w3=w3, b3=b3,
              bn1=unit_bn(b, C, 'bn1'), bn2=unit_bn(b, P, 'bn2'), bn3=unit_bn(b, P, 'bn3'))
    ops = [op]
    if fold:
        op.folded = b.buf('fold', (3 * C + 4 * P,), torch.full((3 * C + 4 * P,), float('nan')))
        ops = [G.Op('bneck_fold', target=op), op]
    b.realise()
    b.run(ops, 0)
    exact_equal(b, y, 'fused bottleneck %r' % (case,))
    # and the fp32 arithmetic gives the same numbers: the specification itself is exact on these inputs
    f = Bench(0)
    f.sizes, f.fills = dict(b.sizes), list(b.fills)
    f.cpu = PI.Arenas(f.sizes, torch.float32)
    for bb, val in f.fills:
        v = f.cpu.view(bb)
        v.copy_(val.reshape(v.shape).to(v.dtype))
    PI.run(f.cpu, [op])
    assert torch.equal(f.cpu.view(y.buf).double(), b.cpu.view(y.buf).double())


@pytest.mark.parametrize('dtype', [0, 1])
@pytest.mark.parametrize('case', [(2, 64, 64, 128), (3, 8, 8, 64), (1, 128, 128, 32)])
def test_elementwise_exact(case, dtype):
    """max-pool (+ first-maximum backward), nearest up-add, 2x2 sum-pool, add: pure data movement / small sums."""
    N, H, W, C = case
    gen = torch.Generator().manual_seed(900 + sum(case))
    bt = Bench(dtype)
    x = bt.act((N, H, W, C), small(gen, N, H, W, C), 'x')
    lo = bt.act((N, H // 2, W // 2, C), small(gen, N, H // 2, W // 2, C), 'lo')
    dy = bt.act((N, H // 2, W // 2, C), small(gen, N, H // 2, W // 2, C), 'dy')
    pool, up, mp_b, sp, ad = (bt.act(s, None, n) for s, n in (((N, H // 2, W // 2, C), 'pool'), ((N, H, W, C), 'up'), ((N, H, W, C), 'mpb'),
                                                              ((N, H // 2, W // 2, C), 'sp'), ((N, H, W, C), 'add')))
    st1 = bt.buf('stats', (RS, 2, C), torch.zeros(RS, 2, C, dtype=torch.float64))
    st2 = bt.buf('stats', (RS, 2, C), torch.zeros(RS, 2, C, dtype=torch.float64))

    def ew(name, dims, **kw):
        f = dict(x=None, x2=None, dy=None, add=None, y=None, out_stats=None, bstats=None, dgamma=None, dbeta=None, bn=None)
        f.update(kw)
        return G.Op('ew', op=name, dims=dims, **f)
    ops = [ew('maxpool_fwd', (N, H, W, C), x=x, y=pool, out_stats=st1),
           ew('upadd_fwd', (N, H, W, C), x=x, x2=lo, y=up, out_stats=st2),
           ew('maxpool_bwd', (N, H, W, C), x=x, dy=dy, y=mp_b),
           ew('sumpool', (N, H, W, C), x=x, add=lo, y=sp),
           ew('add', (N, H, W, C), x=x, x2=up, y=ad)]
    bt.realise().run(ops, 0)
    for t, n in ((pool, 'maxpool'), (up, 'upadd'), (mp_b, 'maxpool_bwd'), (sp, 'sumpool'), (ad, 'add'), (st1, 'pool stats'), (st2, 'upadd stats')):
        exact_equal(bt, t, '%s %s dtype %d' % (n, case, dtype))
