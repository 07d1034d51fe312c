"""Per-kernel parity on the MI355X: every HIP op, called through the C ABI, against its CPU specification
(oracle/plan_interp.py) on seeded random inputs; both storage types, both HIP back ends (MFMA / direct).

Tolerances (stated per test): fp32 build -- abs 2e-4 on O(1..10) values (accumulation-order noise only; the
fp32 MFMA is an exact fp32 fma chain); bf16 build -- compared with the interpreter run with bf16 storage
(same rounding points), abs/rel 2e-2."""
import os

import numpy as np
import pytest
import torch

from oracle import plan_interp as PI
from tests.conftest import ROOT

pytestmark = pytest.mark.gpu

G = E = R = None


def setup_module(module):
    global G, E, R
    from fpd_amd import executor, graph, runtime
    G, E, R = graph, executor, runtime
    R.lib()


class Bench:
    """Pair of identical arenas (CPU interpreter / GPU executor) with a bump allocator for test tensors."""

    def __init__(self, dtype):
        self.dtype = dtype
        self.sizes = {}
        self.fills = []

    def buf(self, arena, shape, fill=None):
        n = int(np.prod(shape)) if len(shape) else 1
        off = self.sizes.get(arena, 0)
        self.sizes[arena] = off + (n + 63) // 64 * 64
        b = G.Buf(arena, off, shape, arena)
        if fill is not None:
            self.fills.append((b, fill))
        return b

    def act(self, shape, fill=None, name='t'):
        a = G.Act(shape, name)
        a.buf = self.buf('act', shape, fill)
        return a

    def realise(self):
        tdt = torch.bfloat16 if self.dtype == 1 else torch.float32
        self.cpu = PI.Arenas(self.sizes, tdt)
        dev = torch.device('cuda:0')
        self.gpu = E.Arenas(dev, self.dtype)
        for name, n in self.sizes.items():
            self.gpu.alloc(name, n)
        for b, val in self.fills:
            v = self.cpu.view(b)
            v.copy_(val.reshape(v.shape).to(v.dtype))
            if b.arena == 'stats':           # device statistics are exact integer limbs (include/fpd_amd.h fpd_stat_t)
                self.gpu.stats_write(b, v)
            else:
                self.gpu.view(b).copy_(v)
        return self

    def run(self, ops, backend, partials=False):
        if backend == 'pp' or (isinstance(backend, tuple) and backend[0] == 'pp'):
            # the persistent ping-pong convolution (csrc/conv_pp.hip) forced for every in-domain bf16 shape; ('pp', n): n blocks,
            # so that a block owns many tiles (uneven ranges, ragged last tiles, long ping-pong loops) even on small tensors
            blocks = backend[1] if isinstance(backend, tuple) else 256
            pm, pb, cm, c3m = R.set_option('conv_pp', 2), R.set_option('conv_pp_blocks', blocks), R.set_option('conv_c1', 0), R.set_option('conv_c3', 0)
            try:
                return self.run(ops, 0, partials)
            finally:
                R.set_option('conv_pp', pm)
                R.set_option('conv_pp_blocks', pb)
                R.set_option('conv_c1', cm)
                R.set_option('conv_c3', c3m)
        if backend == 'c3' or (isinstance(backend, tuple) and backend[0] == 'c3'):
            # the strip kernel for the 3x3 64 -> 64 convolutions (csrc/conv_c3.hip) forced for every in-domain shape; ('c3', n): n blocks
            blocks = backend[1] if isinstance(backend, tuple) else 256
            cm, cb = R.set_option('conv_c3', 2), R.set_option('conv_c3_blocks', blocks)
            n0 = R.set_option('conv_c3_launches', 0)
            try:
                return self.run(ops, 0, partials)
            finally:
                self.n_c3 = R.set_option('conv_c3_launches', 0) - n0
                R.set_option('conv_c3', cm)
                R.set_option('conv_c3_blocks', cb)
        if backend == 'c1' or (isinstance(backend, tuple) and backend[0] == 'c1'):
            # the streaming 1x1 convolution (csrc/conv_c1.hip) forced for every in-domain shape; ('c1', n): n blocks, so that a
            # block owns several rounds (uneven ranges, a ragged last round) even on small tensors.  self.n_c1 = launches it served
            blocks = backend[1] if isinstance(backend, tuple) else 256
            cm, cb = R.set_option('conv_c1', 2), R.set_option('conv_c1_blocks', blocks)
            n0 = R.set_option('conv_c1_launches', 0)
            try:
                return self.run(ops, 0, partials)
            finally:
                self.n_c1 = R.set_option('conv_c1_launches', 0) - n0
                R.set_option('conv_c1', cm)
                R.set_option('conv_c1_blocks', cb)
        if partials:                         # slab reduction of all weight gradients of the list (one gradient bucket)
            wg = [o for o in ops if o.kind == 'wgrad']
            ops = list(ops) + [G.Op('wreduce', bucket=0, wgrads=wg, bufs=[x for w in wg for x in (w.dw, w.dbias) if x is not None])]
        PI.run(self.cpu, ops)
        low = E.Lowering(self.gpu, self.dtype)
        plan = R.Plan()
        prev = R.set_backend(backend)
        low.use_partials = partials          # two-stage weight-gradient reduction (slabs + fpd_wgrad_reduce)
        low.plan_folds([o for o in ops if o.kind != 'wprep'])      # BN-backward applies evaluated by their consumer
        self.n_folded = sum(1 for o in ops if getattr(o, 'folded', False))
        lowered = []
        for op in ops:
            if op.kind == 'wprep':
                lowered.append(low.wprep([(e['w'], e.get('w_fwd'), e.get('w_bwd')) for e in op.entries]))
            else:
                lowered.append(low.op(op))
        low.finish_partials()
        for code, st in lowered:
            plan.add(code, st)
        self.n_partial_ops = len(low.partials)
        self.n_fused = len(low.fused)
        R.set_backend(prev)
        prev = R.set_backend(backend)
        try:
            plan.run(0, len(plan))
            torch.cuda.synchronize()
        finally:
            R.set_backend(prev)

    def compare(self, b, atol, rtol, label):
        b = b.buf if isinstance(b, G.Act) else b
        c = self.cpu.view(b).double()
        if b.arena == 'stats':              # replicated statistics: only the sum over replicas is defined
            c, g = c.sum(0), self.gpu.stats_read(b).cpu().sum(0)
        else:
            g = self.gpu.view(b).cpu().double()
        assert torch.isfinite(g).all(), label + ': non-finite values'
        err = (c - g).abs()
        tol = atol + rtol * c.abs()
        bad = err > tol
        assert not bad.any(), '%s: %d/%d mismatches, max err %.3e (ref max %.3e) first at %s' % (
            label, int(bad.sum()), bad.numel(), float(err.max()), float(c.abs().max()),
            [int(i) for i in torch.nonzero(bad)[0]])


def rnd(gen, *shape, scale=1.0):
    return torch.randn(*shape, generator=gen) * scale


def make_bn(bench, gen, C, mode, name='bn', relu=True, stats_from=None):
    gamma = bench.buf('param', (C,), 1 + 0.2 * rnd(gen, C))
    beta = bench.buf('param', (C,), 0.2 * rnd(gen, C))
    rmean = bench.buf('rstat', (C,), 0.1 * rnd(gen, C))
    rvar = bench.buf('rstat', (C,), 1 + 0.2 * torch.rand(C, generator=gen))
    nbt = bench.buf('nbt', ())
    bn = G.BN(name, mode, C, gamma, beta, rmean, rvar, nbt, relu=relu)
    return bn


RS = 4    # FPD_STATS_REPLICAS: statistics buffers are [R][2][C]; the CPU side keeps everything in replica 0


def tensor_stats(x):   # x [N,H,W,C] -> [R,2,C] fp64 (replica 0 filled)
    xd = x.double()
    out = torch.zeros(RS, 2, x.shape[-1], dtype=torch.float64)
    out[0] = torch.stack([xd.sum((0, 1, 2)), (xd * xd).sum((0, 1, 2))])
    return out


TOL = {0: dict(atol=2e-4, rtol=2e-4), 1: dict(atol=3e-2, rtol=3e-2)}
BACKENDS = [0, 1, 2]      # halo-tile MFMA (default dispatch) / direct kernels / generic MFMA kernel
DTYPES = [0, 1]

CONV_CASES = [
    # N, H, W, C, K, R, pad, bn_mode, residual, stats
    (2, 16, 16, 32, 64, 3, 1, 'train', True, True),
    (2, 16, 16, 64, 32, 1, 0, 'eval', False, True),
    (1, 5, 7, 16, 16, 3, 1, 'train', True, False),      # ragged M, smallest channel counts (score_/score convs)
    (2, 8, 8, 128, 128, 3, 1, 'train', False, True),
    (3, 8, 8, 128, 16, 1, 0, None, False, False),       # score conv: K = J = 16
    (2, 4, 4, 16, 128, 1, 0, None, True, True),         # score_ conv: C = J = 16
    (1, 64, 64, 64, 64, 3, 1, 'train', True, True),     # cfg-1 student 3x3 @64^2
    (2, 8, 8, 256, 128, 1, 0, 'eval', False, False),    # teacher 1x1 256->128
    (2, 8, 8, 48, 96, 3, 1, 'train', False, True),      # HRNet-W48 widths (not multiples of 32/64)
    (2, 16, 16, 3, 64, 3, 1, None, False, True),        # conv_smallc: HRNet's first stem conv (C = 3), statistics for bn1
    (2, 16, 12, 17, 32, 1, 0, None, False, False),      # conv_smallc: data gradient of the J = 17 final layer
    (1, 9, 7, 5, 8, 1, 0, None, True, False),           # conv_smallc: J = 5 test networks, ragged M, accumulate source
    (1, 6, 128, 64, 64, 3, 1, 'eval', False, False),    # teacher layer1 3x3 on 128-wide rows: 32-channel chunks (the 64-channel halo does not fit)
]


@pytest.mark.parametrize('backend', BACKENDS + ['pp'])
@pytest.mark.parametrize('dtype', DTYPES)
@pytest.mark.parametrize('case', CONV_CASES)
def test_conv_forward(case, dtype, backend):
    N, H, W, C, K, Rr, pad, bn_mode, use_res, use_stats = case
    gen = torch.Generator().manual_seed(17 + sum(v for v in case if isinstance(v, int) and not isinstance(v, bool)))
    bt = Bench(dtype)
    x_val = rnd(gen, N, H, W, C) + 0.3
    x = bt.act((N, H, W, C), x_val, 'x')
    w = bt.buf('wlp', (K, Rr, Rr, C), rnd(gen, K, Rr, Rr, C, scale=1.0 / np.sqrt(C * Rr * Rr)))
    bias = bt.buf('param', (K,), 0.1 * rnd(gen, K))
    P, Q = H + 2 * pad - Rr + 1, W + 2 * pad - Rr + 1
    res = bt.act((N, P, Q, K), rnd(gen, N, P, Q, K), 'res') if use_res else None
    y = bt.act((N, P, Q, K), None, 'y')
    bn = None
    if bn_mode:
        bn = make_bn(bt, gen, C, bn_mode)
        bn.count = N * H * W
        if bn_mode == 'train':
            xs = x_val.to(torch.bfloat16).float() if dtype == 1 else x_val
            bn.stats = bt.buf('stats', (RS, 2, C), tensor_stats(xs))
    ostats = bt.buf('stats', (RS, 2, K), torch.zeros(RS, 2, K, dtype=torch.float64)) if use_stats else None
    op = G.Op('conv', x=x, w=w, wkey='w', bias=bias, bkey='b', residual=res, y=y, out_stats=ostats, bn=bn, epi='plain',
              epi_x=None, epi_bn=None, epi_stats=None, dims=(N, H, W, C, K, Rr, Rr, 1, pad, P, Q))
    bt.realise().run([op], backend)
    bt.compare(y, label='conv y %s' % (case,), **TOL[dtype])
    if use_stats:
        scale = N * P * Q
        bt.compare(ostats, atol=TOL[dtype]['atol'] * scale, rtol=TOL[dtype]['rtol'], label='conv out_stats')


# the persistent ping-pong kernel (csrc/conv_pp.hip) on shapes with many tiles per block: (N, H, W, C, K, R, bn, residual, blocks)
PP_CASES = [
    (4, 64, 64, 64, 64, 3, 'train', True, 8),        # student 3x3 @64^2: 16 tiles per block, 8 per group
    (4, 64, 64, 64, 64, 3, 'train', False, 5),       # uneven tile ranges (128 tiles over 5 blocks)
    (2, 64, 64, 128, 64, 1, 'train', False, 4),      # conv1 of a student Bottleneck
    (2, 64, 64, 64, 128, 1, 'train', True, 8),       # conv3 (+ residual): K = 128 = two channel slabs per tile range
    (2, 64, 64, 128, 128, 1, None, False, 16),       # fc / fc_
    (2, 64, 64, 128, 16, 1, 'eval', False, 4),       # score: K = J = 16 (half-empty accumulator tile)
    (2, 64, 64, 16, 128, 1, None, True, 4),          # score_: C = J = 16 (one k-step)
    (1, 128, 128, 32, 32, 3, 'train', False, 6),     # layer1 @128^2: one image row per tile
    (2, 128, 128, 32, 64, 1, 'train', True, 7),
    (3, 20, 48, 32, 64, 3, 'train', True, 3),        # HRNet width 48: 96-pixel tiles, tiles straddling images (20 rows, 2 per tile)
    (3, 9, 7, 16, 24, 3, 'train', True, 2),          # ragged everything: W = 7, K = 24, last tile partly outside the tensor
    (1, 64, 64, 64, 64, 3, None, False, 256),        # more blocks than tile pairs
]


@pytest.mark.parametrize('case', PP_CASES)
def test_conv_pp_forward(case):
    N, H, W, C, K, Rr, bn_mode, use_res, blocks = case
    pad = (Rr - 1) // 2
    gen = torch.Generator().manual_seed(23 + sum(v for v in case if isinstance(v, int) and not isinstance(v, bool)))
    bt = Bench(1)
    x_val = rnd(gen, N, H, W, C) + 0.3
    x = bt.act((N, H, W, C), x_val, 'x')
    w = bt.buf('wlp', (K, Rr, Rr, C), rnd(gen, K, Rr, Rr, C, scale=1.0 / np.sqrt(C * Rr * Rr)))
    bias = bt.buf('param', (K,), 0.1 * rnd(gen, K))
    res = bt.act((N, H, W, K), rnd(gen, N, H, W, K), 'res') if use_res else None
    y = bt.act((N, H, W, K), None, 'y')
    bn = None
    if bn_mode:
        bn = make_bn(bt, gen, C, bn_mode)
        bn.count = N * H * W
        if bn_mode == 'train':
            bn.stats = bt.buf('stats', (RS, 2, C), tensor_stats(x_val.to(torch.bfloat16).float()))
    ostats = bt.buf('stats', (RS, 2, K), torch.zeros(RS, 2, K, dtype=torch.float64))
    op = G.Op('conv', x=x, w=w, wkey='w', bias=bias, bkey='b', residual=res, y=y, out_stats=ostats, bn=bn, epi='plain',
              epi_x=None, epi_bn=None, epi_stats=None, dims=(N, H, W, C, K, Rr, Rr, 1, pad, H, W))
    bt.realise().run([op], ('pp', blocks))
    bt.compare(y, label='conv_pp y %s' % (case,), **TOL[1])
    bt.compare(ostats, atol=TOL[1]['atol'] * N * H * W, rtol=TOL[1]['rtol'], label='conv_pp out_stats')


@pytest.mark.parametrize('case', [(4, 64, 64, 64, 64, 3, 8), (2, 64, 64, 64, 128, 1, 8), (2, 64, 64, 128, 64, 1, 5),
                                  (1, 128, 128, 32, 32, 3, 6), (3, 20, 48, 32, 64, 3, 3), (2, 64, 64, 16, 128, 1, 4)])
def test_conv_pp_dgrad(case):
    """BatchNorm-backward epilogue (ReLU mask of epi_x + the two sums) of the ping-pong kernel, with an accumulate source
    that aliases the output; forward convolution C -> K, i.e. the data gradient has K input and C output channels."""
    N, H, W, C, K, Rr, blocks = case
    pad = (Rr - 1) // 2
    gen = torch.Generator().manual_seed(29 + sum(case))
    bt = Bench(1)
    x_val = rnd(gen, N, H, W, C)
    x = bt.act((N, H, W, C), x_val, 'x')
    dy = bt.act((N, H, W, K), rnd(gen, N, H, W, K), 'dy')
    wm = bt.buf('param', (K, Rr, Rr, C), rnd(gen, K, Rr, Rr, C, scale=1.0 / np.sqrt(C * Rr * Rr)))
    wb = bt.buf('wlp', (C, Rr, Rr, K))
    prev = bt.act((N, H, W, C), 0.5 * rnd(gen, N, H, W, C), 'prev')
    bn = make_bn(bt, gen, C, 'train')
    bn.count = N * H * W
    bn.stats = bt.buf('stats', (RS, 2, C), tensor_stats(x_val.to(torch.bfloat16).float()))
    bst = bt.buf('stats', (RS, 2, C), torch.zeros(RS, 2, C, dtype=torch.float64))
    ops = [G.Op('wprep', entries=[{'w': wm, 'w_fwd': None, 'w_bwd': wb}]),
           G.Op('conv', x=dy, w=wb, wkey='w', bias=None, bkey=None, residual=prev, y=prev, out_stats=None, bn=None,
                epi='bnrelu_bwd', epi_x=x, epi_bn=bn, epi_stats=bst, dims=(N, H, W, K, C, Rr, Rr, 1, Rr - 1 - pad, H, W))]
    bt.realise().run(ops, ('pp', blocks))
    bt.compare(prev, label='conv_pp dgrad dz %s' % (case,), **TOL[1])
    bt.compare(bst, atol=TOL[1]['atol'] * N * H * W, rtol=TOL[1]['rtol'], label='conv_pp dgrad bn sums')


@pytest.mark.parametrize('case', [(4, 64, 64, 128, 64, 8, True), (4, 64, 64, 64, 128, 8, True), (2, 64, 64, 128, 128, 5, True),
                                  (1, 128, 128, 32, 32, 6, True), (2, 128, 128, 32, 64, 7, False), (3, 20, 48, 64, 32, 3, True),
                                  (2, 32, 32, 128, 64, 4, True)])
def test_conv_pp_dgrad_with_fused_weight_gradient(case):
    """The persistent kernel's data gradient of a 1x1 convolution also forms that convolution's weight / bias gradient
    (fpd_conv_t.wg_partial: dy is its operand image, relu(bn(u)) falls out of the ReLU-mask evaluation): the separate
    'wgrad' op must become a no-op and dw / dbias must match the specification of the un-fused op."""
    N, H, W, C, K, blocks, bias = case                    # forward convolution C -> K (1x1)
    gen = torch.Generator().manual_seed(37 + sum(case[:6]))
    bt = Bench(1)
    x_val = rnd(gen, N, H, W, C)
    x = bt.act((N, H, W, C), x_val, 'x')
    dy = bt.act((N, H, W, K), rnd(gen, N, H, W, K, scale=0.1), 'dy')
    wm = bt.buf('param', (K, 1, 1, C), rnd(gen, K, 1, 1, C, scale=1.0 / np.sqrt(C)))
    wb = bt.buf('wlp', (C, 1, 1, K))
    prev = bt.act((N, H, W, C), 0.5 * rnd(gen, N, H, W, C), 'prev')
    dz = bt.act((N, H, W, C), None, 'dz')
    bn = make_bn(bt, gen, C, 'train')
    bn.count = N * H * W
    bn.stats = bt.buf('stats', (RS, 2, C), tensor_stats(x_val.to(torch.bfloat16).float()))
    bst = bt.buf('stats', (RS, 2, C), torch.zeros(RS, 2, C, dtype=torch.float64))
    dw = bt.buf('grad', (K, 1, 1, C), torch.zeros(K, 1, 1, C))
    db = bt.buf('grad', (K,), torch.zeros(K)) if bias else None
    wg = G.Op('wgrad', x=x, dy=dy, dw=dw, dbias=db, bn=bn, dims=(N, H, W, C, K, 1, 1, 1, 0, H, W))
    dg = G.Op('conv', x=dy, w=wb, wkey='w', bias=None, bkey=None, residual=prev, y=dz, out_stats=None, bn=None,
              epi='bnrelu_bwd', epi_x=x, epi_bn=bn, epi_stats=bst, dims=(N, H, W, K, C, 1, 1, 1, 0, H, W))
    dg.fused_wgrad = wg
    ops = [G.Op('wprep', entries=[{'w': wm, 'w_fwd': None, 'w_bwd': wb}]), dg, wg]
    bt.realise().run(ops, ('pp', blocks), partials=True)
    assert bt.n_fused == 1 and getattr(dg, 'fused_active', False), 'the weight gradient was not fused into the data gradient'
    bt.compare(dz, label='fused dgrad dz %s' % (case,), **TOL[1])
    bt.compare(bst, atol=TOL[1]['atol'] * N * H * W, rtol=TOL[1]['rtol'], label='fused dgrad bn sums')
    m = N * H * W
    tol = dict(atol=2e-2 + 2e-5 * m, rtol=3e-2)
    bt.compare(dw, label='fused wgrad dw %s' % (case,), **tol)
    if bias:
        bt.compare(db, label='fused wgrad dbias', **tol)


FOLD_CASES = [(4, 64, 64, 128, 64, 1, 8, True), (4, 64, 64, 64, 128, 1, 8, False), (2, 64, 64, 64, 64, 3, 5, False),
              (1, 128, 128, 32, 32, 3, 6, False), (3, 20, 48, 64, 32, 1, 3, True), (2, 32, 32, 64, 64, 3, 4, False),
              (2, 32, 32, 128, 64, 1, 4, True),
              # small maps: served by the halo-tile kernel under the default dispatch (its FOLD variants, TN <= 2)
              (32, 8, 8, 64, 64, 3, 4, False), (32, 16, 16, 128, 64, 1, 4, True), (32, 4, 4, 128, 64, 1, 2, True),
              (8, 16, 16, 64, 64, 3, 3, False)]


@pytest.mark.parametrize('forced', [True, False])
@pytest.mark.parametrize('case', FOLD_CASES)
def test_conv_pp_dgrad_with_folded_bn_backward_apply(case, forced):
    """A data gradient whose operand is the output of a BN-backward apply evaluates the apply itself on its operand load
    (fpd_conv_t.fold_x): the stand-alone apply must become a no-op and everything downstream -- the data gradient with its
    ReLU mask and sums, the convolution's weight / bias gradient (fused into the launch for 1x1, a separate launch reading
    the materialised operand for 3x3), the BN's dgamma / dbeta -- must match the specification of the un-folded op list."""
    N, H, W, C, K, Rr, blocks, bias = case                # forward convolution u = conv(a(x)), C -> K, then bn_next(u)
    pad = (Rr - 1) // 2
    gen = torch.Generator().manual_seed(53 + sum(case[:7]))
    bt = Bench(1)
    x_val = rnd(gen, N, H, W, C)
    x = bt.act((N, H, W, C), x_val, 'x')                  # forward input of the convolution (pre BN+ReLU)
    u_val = rnd(gen, N, H, W, K)
    u = bt.act((N, H, W, K), u_val, 'u')                  # its output = input of the next BN
    g_val = rnd(gen, N, H, W, K, scale=0.1)
    g = bt.act((N, H, W, K), g_val, 'g')                  # masked gradient that reached that BN
    wm = bt.buf('param', (K, Rr, Rr, C), rnd(gen, K, Rr, Rr, C, scale=1.0 / np.sqrt(C * Rr * Rr)))
    wb = bt.buf('wlp', (C, Rr, Rr, K))
    du = bt.act((N, H, W, K), None, 'du')
    dz = bt.act((N, H, W, C), None, 'dz')
    bn = make_bn(bt, gen, C, 'train')                     # BN in front of the convolution (mask + sums of the data gradient)
    bn.count = N * H * W
    bn.stats = bt.buf('stats', (RS, 2, C), tensor_stats(x_val.to(torch.bfloat16).float()))
    bn2 = make_bn(bt, gen, K, 'train')                    # the BN whose backward is folded
    bn2.count = N * H * W
    bn2.stats = bt.buf('stats', (RS, 2, K), tensor_stats(u_val.to(torch.bfloat16).float()))
    gq, uq = g_val.to(torch.bfloat16).float(), u_val.to(torch.bfloat16).float()
    mean, var = uq.mean((0, 1, 2)), uq.var((0, 1, 2), unbiased=False)
    xhat = (uq - mean) / torch.sqrt(var + 1e-5)
    sums = torch.zeros(RS, 2, K, dtype=torch.float64)
    sums[0, 0], sums[0, 1] = gq.double().sum((0, 1, 2)), (gq.double() * xhat.double()).sum((0, 1, 2))
    bst2 = bt.buf('stats', (RS, 2, K), sums)
    dgam, dbet = bt.buf('grad', (K,), torch.zeros(K)), bt.buf('grad', (K,), torch.zeros(K))
    bst = bt.buf('stats', (RS, 2, C), torch.zeros(RS, 2, C, dtype=torch.float64))
    dw = bt.buf('grad', (K, Rr, Rr, C), torch.zeros(K, Rr, Rr, C))
    db = bt.buf('grad', (K,), torch.zeros(K)) if bias else None
    ap = G.Op('ew', op='bn_bwd_apply', dims=(N, H, W, K), x=u, x2=None, dy=g, add=None, y=du, out_stats=None, bstats=bst2,
              dgamma=dgam, dbeta=dbet, bn=bn2)
    wg = G.Op('wgrad', x=x, dy=du, dw=dw, dbias=db, bn=bn, dims=(N, H, W, C, K, Rr, Rr, 1, pad, H, W))
    dg = G.Op('conv', x=du, w=wb, wkey='w', bias=None, bkey=None, residual=None, y=dz, out_stats=None, bn=None,
              epi='bnrelu_bwd', epi_x=x, epi_bn=bn, epi_stats=bst, dims=(N, H, W, K, C, Rr, Rr, 1, Rr - 1 - pad, H, W))
    dg.fold_apply, dg.fold_wgrad = ap, wg
    if Rr == 1:
        dg.fused_wgrad = wg
    ops = [G.Op('wprep', entries=[{'w': wm, 'w_fwd': None, 'w_bwd': wb}]), ap, dg, wg]
    # forced: the persistent kernel for every shape; else the default dispatch (persistent kernel from 256 tiles, halo-tile below)
    bt.realise().run(ops, ('pp', blocks) if forced else 0, partials=True)
    # (default dispatch: a mid-sized launch of the halo-tile kernel with 128 output channels per block has no FOLD variant --
    # the library then declines and the apply runs as its own launch; the results below must hold either way)
    if forced or case in FOLD_CASES[-4:]:
        assert bt.n_folded == 1 and getattr(dg, 'fold_active', False), 'the BN-backward apply was not folded into the data gradient'
    if forced:
        assert (bt.n_fused == 1) == (Rr == 1)
    bt.compare(dz, label='folded dgrad dz %s' % (case,), **TOL[1])
    bt.compare(bst, atol=TOL[1]['atol'] * N * H * W, rtol=TOL[1]['rtol'], label='folded dgrad bn sums')
    bt.compare(dgam, atol=1e-3, rtol=1e-4, label='dgamma of the folded BN')
    bt.compare(dbet, atol=1e-3, rtol=1e-4, label='dbeta of the folded BN')
    m = N * H * W
    tol = dict(atol=2e-2 + 2e-5 * m, rtol=3e-2)
    bt.compare(dw, label='wgrad dw behind the fold %s' % (case,), **tol)
    if bias:
        bt.compare(db, label='wgrad dbias behind the fold', **tol)
    if bt.n_fused == 0:                                   # the operand was materialised for the separate weight-gradient launch
        bt.compare(du, label='materialised operand', **TOL[1])


def test_conv_pp_equals_conv_tile_bitwise():
    """Both kernels round at the same points (fp32 accumulate, + residual, + bias, one rounding; statistics of the rounded
    values), so on one tensor they must agree to the accumulation order of the MFMA chain: identical here because both
    walk the taps and channel chunks in the same order."""
    N, H, W, C, K = 2, 64, 64, 64, 64
    gen = torch.Generator().manual_seed(31)
    outs = []
    for backend in (0, ('pp', 8)):
        g2 = torch.Generator().manual_seed(31)
        bt = Bench(1)
        x_val = rnd(g2, N, H, W, C) + 0.3
        x = bt.act((N, H, W, C), x_val, 'x')
        w = bt.buf('wlp', (K, 3, 3, C), rnd(g2, K, 3, 3, C, scale=1.0 / np.sqrt(C * 9)))
        bias = bt.buf('param', (K,), 0.1 * rnd(g2, K))
        res = bt.act((N, H, W, K), rnd(g2, N, H, W, K), 'res')
        y = bt.act((N, H, W, K), None, 'y')
        bn = make_bn(bt, g2, C, 'train')
        bn.count = N * H * W
        bn.stats = bt.buf('stats', (RS, 2, C), tensor_stats(x_val.to(torch.bfloat16).float()))
        op = G.Op('conv', x=x, w=w, wkey='w', bias=bias, bkey='b', residual=res, y=y, out_stats=None, bn=bn, epi='plain',
                  epi_x=None, epi_bn=None, epi_stats=None, dims=(N, H, W, C, K, 3, 3, 1, 1, H, W))
        bt.realise().run([op], backend)
        outs.append(bt.gpu.view(y.buf).cpu().view(torch.int16).clone())
    assert torch.equal(outs[0], outs[1])


DGRAD_CASES = [(2, 16, 16, 32, 64, 3, 1), (2, 8, 8, 64, 128, 1, 0), (1, 6, 5, 16, 32, 3, 1), (2, 32, 32, 64, 64, 3, 1)]


@pytest.mark.parametrize('backend', BACKENDS + ['pp', ('pp', 3)])
@pytest.mark.parametrize('dtype', DTYPES)
@pytest.mark.parametrize('case', DGRAD_CASES)
def test_conv_dgrad_bnrelu_epilogue(case, dtype, backend):
    """Data gradient of y = conv(relu(bn(x))) w.r.t. bn output: flipped weights via wprep, ReLU mask and BN sums."""
    N, H, W, C, K, Rr, pad = case            # forward conv C -> K
    gen = torch.Generator().manual_seed(7 + sum(case))
    bt = Bench(dtype)
    P, Q = H + 2 * pad - Rr + 1, W + 2 * pad - Rr + 1
    x_val = rnd(gen, N, H, W, C)
    x = bt.act((N, H, W, C), x_val, 'x')
    dy = bt.act((N, P, Q, K), rnd(gen, N, P, Q, K), 'dy')
    wm = bt.buf('param', (K, Rr, Rr, C), rnd(gen, K, Rr, Rr, C, scale=1.0 / np.sqrt(C * Rr * Rr)))
    wb = bt.buf('wlp', (C, Rr, Rr, K))
    prev = bt.act((N, H, W, C), 0.5 * rnd(gen, N, H, W, C), 'prev')     # accumulate source (aliases dz)
    bn = make_bn(bt, gen, C, 'train')
    bn.count = N * H * W
    xs = x_val.to(torch.bfloat16).float() if dtype == 1 else x_val
    bn.stats = bt.buf('stats', (RS, 2, C), tensor_stats(xs))
    bst = bt.buf('stats', (RS, 2, C), torch.zeros(RS, 2, C, dtype=torch.float64))
    ops = [G.Op('wprep', entries=[{'w': wm, 'w_fwd': None, 'w_bwd': wb}]),
           G.Op('conv', x=dy, w=wb, wkey='w', bias=None, bkey=None, residual=prev, y=prev, out_stats=None, bn=None,
                epi='bnrelu_bwd', epi_x=x, epi_bn=bn, epi_stats=bst, dims=(N, P, Q, K, C, Rr, Rr, 1, Rr - 1 - pad, H, W))]
    bt.realise().run(ops, backend)
    bt.compare(wb, atol=1e-6 if dtype == 0 else 1e-2, rtol=0, label='wprep w_bwd')
    bt.compare(prev, label='dgrad dz %s' % (case,), **TOL[dtype])
    bt.compare(bst, atol=TOL[dtype]['atol'] * N * H * W, rtol=TOL[dtype]['rtol'], label='dgrad bn sums')


WGRAD_CASES = [(2, 16, 16, 32, 64, 3, 1, True), (2, 8, 8, 128, 64, 1, 0, True), (1, 9, 7, 16, 16, 3, 1, False),
               (4, 32, 32, 64, 64, 3, 1, True), (2, 8, 8, 64, 128, 1, 0, False), (2, 16, 16, 128, 16, 1, 0, True),
               (32, 4, 4, 64, 64, 3, 1, True), (32, 8, 8, 64, 64, 3, 1, True), (32, 4, 4, 128, 64, 1, 0, True),   # deepest hourglass levels
               (4, 32, 24, 3, 64, 3, 1, False), (2, 16, 12, 3, 16, 3, 1, False), (3, 20, 20, 4, 8, 1, 0, False),   # tiny C: wgrad_smallc
               (2, 64, 48, 32, 32, 3, 1, True), (2, 16, 16, 16, 32, 3, 1, True), (3, 32, 32, 32, 16, 3, 1, False),   # C, K <= 32: wgrad_tile SMALL
               (2, 16, 48, 32, 64, 3, 1, True), (3, 32, 24, 64, 64, 3, 1, True), (2, 64, 48, 32, 32, 1, 0, False),   # HRNet map widths
               (3, 16, 12, 128, 128, 3, 1, True), (2, 32, 24, 64, 32, 1, 0, True), (5, 16, 12, 128, 64, 1, 0, False),
               (32, 32, 24, 64, 64, 3, 1, True),
               (2, 64, 64, 64, 64, 3, 1, True), (32, 16, 16, 64, 64, 3, 1, True), (3, 9, 16, 64, 64, 3, 1, True),   # wgrad3: ring halo, ranges x pieces
               (8, 64, 64, 64, 64, 3, 1, False), (7, 32, 32, 64, 64, 3, 1, True),
               (2, 128, 128, 32, 32, 3, 1, True), (8, 32, 32, 32, 32, 3, 1, True)]      # wgrad3, one piece: the hourglass' layer1 3x3 at 128x128
TILE_ONLY_CASES = WGRAD_CASES[-14:-7]      # every HRNet width that is a multiple of 4 must be taken by the halo-tile kernel


@pytest.mark.parametrize('backend', BACKENDS + ['partials', 'tile_only'])
@pytest.mark.parametrize('dtype', DTYPES)
@pytest.mark.parametrize('case', WGRAD_CASES)
def test_conv_wgrad(case, dtype, backend):
    N, H, W, C, K, Rr, pad, use_bn = case
    if backend == 'tile_only' and (case not in TILE_ONLY_CASES or dtype != 1):
        pytest.skip('halo-tile-kernel-only run: bf16 HRNet widths')
    gen = torch.Generator().manual_seed(11 + sum(case[:7]))
    bt = Bench(dtype)
    P, Q = H + 2 * pad - Rr + 1, W + 2 * pad - Rr + 1
    x_val = rnd(gen, N, H, W, C)
    x = bt.act((N, H, W, C), x_val, 'x')
    dy = bt.act((N, P, Q, K), rnd(gen, N, P, Q, K, scale=0.1), 'dy')
    dw = bt.buf('grad', (K, Rr, Rr, C), torch.zeros(K, Rr, Rr, C))
    db = bt.buf('grad', (K,), torch.zeros(K))
    bn = None
    if use_bn:
        bn = make_bn(bt, gen, C, 'train')
        bn.count = N * H * W
        xs = x_val.to(torch.bfloat16).float() if dtype == 1 else x_val
        bn.stats = bt.buf('stats', (RS, 2, C), tensor_stats(xs))
    op = G.Op('wgrad', x=x, dy=dy, dw=dw, dbias=db, bn=bn, dims=(N, H, W, C, K, Rr, Rr, 1, pad, P, Q))
    if backend == 'partials':          # default dispatch with the two-stage (slab + reduce) flush instead of atomics
        bt.realise().run([op], 0, partials=True)
        assert bt.n_partial_ops == 1           # every weight-gradient kernel writes slabs (no floating-point atomics)
    elif backend == 'tile_only':       # the library refuses to fall through to the generic kernels: wgrad_tile must take it
        R.set_option('wgrad_tile_only', 1)
        try:
            bt.realise().run([op], 0, partials=True)
        finally:
            R.set_option('wgrad_tile_only', 0)
    else:
        bt.realise().run([op], backend)
    m = N * P * Q
    tol = dict(atol=2e-4 + 1e-6 * m, rtol=2e-4) if dtype == 0 else dict(atol=2e-2 + 2e-5 * m, rtol=3e-2)
    bt.compare(dw, label='wgrad dw %s' % (case,), **tol)
    bt.compare(db, label='wgrad dbias', **tol)


@pytest.mark.parametrize('partials', [True, False])
@pytest.mark.parametrize('dtype', DTYPES)
@pytest.mark.parametrize('case', [(8, 64, 64, 64, 64, 3, 1, True), (8, 64, 64, 128, 64, 1, 0, True), (32, 8, 8, 64, 64, 3, 1, True),
                                  (32, 4, 4, 128, 64, 1, 0, True), (4, 32, 24, 3, 64, 3, 1, False), (4, 16, 12, 32, 17, 1, 0, False)])
def test_conv_wgrad_is_bit_repeatable(case, dtype, partials):
    """No floating-point atomics in any weight-gradient kernel (halo-tile, generic MFMA, direct): every sum is formed in a
    fixed order -- slabs + fpd_wgrad_reduce, or a single block per output group when no slabs are given -- so repeated
    runs on the same inputs give IDENTICAL bytes (deterministic reduction order)."""
    N, H, W, C, K, Rr, pad, use_bn = case
    if not partials and N * H * W > 16384:
        pytest.skip('single-block flush: small problems only')
    outs = []
    for rep in range(3):
        gen = torch.Generator().manual_seed(41 + sum(case[:7]))
        bt = Bench(dtype)
        x_val = rnd(gen, N, H, W, C)
        x = bt.act((N, H, W, C), x_val, 'x')
        dy = bt.act((N, H, W, K), rnd(gen, N, H, W, K, scale=0.1), 'dy')
        dw = bt.buf('grad', (K, Rr, Rr, C), torch.zeros(K, Rr, Rr, C))
        db = bt.buf('grad', (K,), torch.zeros(K))
        bn = None
        if use_bn:
            bn = make_bn(bt, gen, C, 'train')
            bn.count = N * H * W
            xs = x_val.to(torch.bfloat16).float() if dtype == 1 else x_val
            bn.stats = bt.buf('stats', (RS, 2, C), tensor_stats(xs))
        op = G.Op('wgrad', x=x, dy=dy, dw=dw, dbias=db, bn=bn, dims=(N, H, W, C, K, Rr, Rr, 1, pad, H, W))
        bt.realise().run([op], 0, partials=partials)
        outs.append((bt.gpu.view(dw).cpu().clone(), bt.gpu.view(db).cpu().clone()))
    for a, b in outs[1:]:
        assert torch.equal(a.view(torch.int32), outs[0][0].view(torch.int32)), 'dw differs between identical runs'
        assert torch.equal(b.view(torch.int32), outs[0][1].view(torch.int32)), 'dbias differs between identical runs'


@pytest.mark.parametrize('dtype', DTYPES)
def test_stem_wgrad_is_bit_repeatable(dtype):
    N, H, W, K = 4, 128, 128, 32
    outs = []
    for rep in range(3):
        gen = torch.Generator().manual_seed(43)
        bt = Bench(dtype)
        img = bt.buf('image', (N, 3, H, W), rnd(gen, N, 3, H, W))
        dy = bt.act((N, H // 2, W // 2, K), rnd(gen, N, H // 2, W // 2, K, scale=0.1), 'dy')
        dw = bt.buf('grad', (K, 7, 7, 3), torch.zeros(K, 7, 7, 3))
        db = bt.buf('grad', (K,), torch.zeros(K))
        sw = G.Op('stem_wgrad', image=img, dy=dy, dw=dw, dbias=db, dims=(N, H, W, K, H // 2, W // 2))
        ops = [sw, G.Op('wreduce', bucket=0, wgrads=[sw], bufs=[dw, db])]
        bt.realise()
        low = E.Lowering(bt.gpu, dtype)
        low.use_partials = True
        plan = R.Plan()
        lowered = [low.op(o) for o in ops]
        low.finish_partials()
        for code, st in lowered:
            plan.add(code, st)
        assert len(low.partials) == 1
        plan.run(0, len(plan))
        torch.cuda.synchronize()
        outs.append((bt.gpu.view(dw).cpu().clone(), bt.gpu.view(db).cpu().clone()))
    for a, b in outs[1:]:
        assert torch.equal(a.view(torch.int32), outs[0][0].view(torch.int32))
        assert torch.equal(b.view(torch.int32), outs[0][1].view(torch.int32))


@pytest.mark.parametrize('backend', [0, 1, 'partials'])
@pytest.mark.parametrize('dtype', DTYPES)
@pytest.mark.parametrize('case', [(2, 64, 64, 16), (1, 128, 96, 32), (2, 32, 48, 64), (2, 128, 128, 32), (1, 256, 256, 64)])
def test_stem(case, dtype, backend):
    N, H, W, K = case
    gen = torch.Generator().manual_seed(3 + sum(case))
    bt = Bench(dtype)
    P, Q = H // 2, W // 2
    img = bt.buf('image', (N, 3, H, W), rnd(gen, N, 3, H, W))
    w = bt.buf('param', (K, 7, 7, 3), rnd(gen, K, 7, 7, 3, scale=1 / np.sqrt(147)))
    b = bt.buf('param', (K,), 0.1 * rnd(gen, K))
    y = bt.act((N, P, Q, K), None, 'y')
    st = bt.buf('stats', (RS, 2, K), torch.zeros(RS, 2, K, dtype=torch.float64))
    dy = bt.act((N, P, Q, K), rnd(gen, N, P, Q, K, scale=0.1), 'dy')
    dw = bt.buf('grad', (K, 7, 7, 3), torch.zeros(K, 7, 7, 3))
    db = bt.buf('grad', (K,), torch.zeros(K))
    sw = G.Op('stem_wgrad', image=img, dy=dy, dw=dw, dbias=db, dims=(N, H, W, K, P, Q))
    ops = [G.Op('stem_fwd', image=img, w=w, bias=b, y=y, out_stats=st, dims=(N, H, W, K, P, Q)), sw]
    if backend == 'partials':           # slabs + reduction, as in a training plan
        ops.append(G.Op('wreduce', bucket=0, wgrads=[sw], bufs=[dw, db]))
        bt.realise()
        PI.run(bt.cpu, ops)
        low = E.Lowering(bt.gpu, dtype)
        low.use_partials = True
        plan = R.Plan()
        lowered = [low.op(o) for o in ops]
        low.finish_partials()
        for code, s_ in lowered:
            plan.add(code, s_)
        plan.run(0, len(plan))
        torch.cuda.synchronize()
    else:
        bt.realise().run(ops, backend)
    bt.compare(y, label='stem y', **TOL[dtype])
    bt.compare(st, atol=TOL[dtype]['atol'] * N * P * Q, rtol=TOL[dtype]['rtol'], label='stem stats')
    m = N * P * Q
    tol = dict(atol=2e-4 + 1e-6 * m, rtol=2e-4) if dtype == 0 else dict(atol=2e-2 + 2e-5 * m, rtol=3e-2)
    bt.compare(dw, label='stem dw', **tol)
    bt.compare(db, label='stem dbias', **tol)


@pytest.mark.parametrize('dtype', DTYPES)
@pytest.mark.parametrize('shape', [(2, 16, 16, 32), (1, 6, 10, 16), (2, 8, 8, 256), (3, 4, 4, 48)])
def test_elementwise_ops(shape, dtype):
    N, H, W, C = shape
    gen = torch.Generator().manual_seed(5 + sum(shape))
    bt = Bench(dtype)
    x_val = rnd(gen, N, H, W, C)
    x = bt.act(shape, x_val, 'x')
    half = (N, H // 2, W // 2, C)
    lo = bt.act(half, rnd(gen, *half), 'lo')
    dy = bt.act(shape, rnd(gen, *shape), 'dy')
    dyh = bt.act(half, rnd(gen, *half), 'dyh')
    acc = bt.act(shape, rnd(gen, *shape), 'acc')
    acch = bt.act(half, rnd(gen, *half), 'acch')
    bn = make_bn(bt, gen, C, 'train')
    bn.count = N * H * W
    xs = x_val.to(torch.bfloat16).float() if dtype == 1 else x_val
    bn.stats = bt.buf('stats', (RS, 2, C), tensor_stats(xs))

    def z():
        return bt.buf('stats', (RS, 2, C), torch.zeros(RS, 2, C, dtype=torch.float64))
    y1, y2, y3, y4, y5, y6, y7, y8 = [bt.act(s, None, 'y%d' % i) for i, s in enumerate(
        [shape, shape, shape, half, shape, shape, half, shape])]
    s1, s4, s6, bst = z(), z(), z(), z()
    dg = bt.buf('grad', (C,), torch.zeros(C))
    dbt = bt.buf('grad', (C,), torch.zeros(C))

    def ew(op, dims, **kw):
        d = dict(x=None, x2=None, dy=None, add=None, out_stats=None, bstats=None, dgamma=None, dbeta=None, bn=None)
        d.update(kw)
        return G.Op('ew', op=op, dims=dims, **d)
    ops = [
        ew('bnrelu_fwd', shape, x=x, y=y1, bn=bn, out_stats=s1),
        ew('bnrelu_bwd_r', shape, x=x, dy=dy, y=y2, bn=bn, bstats=bst),
        ew('bn_bwd_apply', shape, x=x, dy=y2, add=acc, y=y3, bn=bn, bstats=bst, dgamma=dg, dbeta=dbt),
        ew('maxpool_fwd', shape, x=x, y=y4, out_stats=s4),
        ew('maxpool_bwd', shape, x=x, dy=dyh, add=acc, y=y5),
        ew('upadd_fwd', shape, x=x, x2=lo, y=y6, out_stats=s6),
        ew('sumpool', shape, x=dy, add=acch, y=y7),
        ew('add', shape, x=x, x2=dy, y=y8),
    ]
    bt.realise().run(ops, 0)
    for name, t in (('bnrelu_fwd', y1), ('bnrelu_bwd_r', y2), ('bn_bwd_apply', y3), ('maxpool_fwd', y4),
                    ('maxpool_bwd', y5), ('upadd_fwd', y6), ('sumpool', y7), ('add', y8)):
        bt.compare(t, label=name, **TOL[dtype])
    n = N * H * W
    for name, t in (('stats bnrelu', s1), ('stats maxpool', s4), ('stats upadd', s6), ('bstats', bst)):
        bt.compare(t, atol=TOL[dtype]['atol'] * n, rtol=TOL[dtype]['rtol'], label=name)
    bt.compare(dg, atol=TOL[dtype]['atol'] * n, rtol=TOL[dtype]['rtol'], label='dgamma')
    bt.compare(dbt, atol=TOL[dtype]['atol'] * n, rtol=TOL[dtype]['rtol'], label='dbeta')


@pytest.mark.parametrize('dtype', DTYPES)
def test_statistics_limbs_have_headroom_for_many_small_addends(dtype):
    """the exact statistics add one contribution per THREAD in the elementwise producers -- 32 768 per
    channel at C = 32 -- and each leaves a residual below one hi unit in the lo limb.  With hi in units of 2^-8 the lo limbs of
    sub-2^-9 contributions (sum y^2 of a channel with rms 0.009) wrapped 2^63 once they added up to 8; hi units of 2^-20 leave
    2^24 addends.  Pooled map of 131 072 pixels per channel, sum of squares ~ 25: must match fp64 to fp32-partial accuracy."""
    shape = (32, 128, 128, 32)
    gen = torch.Generator().manual_seed(77)
    bt = Bench(dtype)
    x = bt.act(shape, 0.009 * rnd(gen, *shape) + 0.012, 'x')
    y = bt.act((32, 64, 64, 32), None, 'y')
    st = bt.buf('stats', (RS, 2, 32), torch.zeros(RS, 2, 32, dtype=torch.float64))
    op = G.Op('ew', op='maxpool_fwd', dims=shape, x=x, x2=None, dy=None, add=None, y=y, out_stats=st, bstats=None,
              dgamma=None, dbeta=None, bn=None)
    bt.realise().run([op], 0)
    got = bt.gpu.stats_read(st).cpu().sum(0)
    yd = bt.gpu.view(y.buf).cpu().double()
    want = torch.stack([yd.sum((0, 1, 2)), (yd * yd).sum((0, 1, 2))])
    assert float(want[1].min()) > 8.0                      # the regime in which version 1 wrapped
    assert float(((got - want).abs() / want.abs()).max()) < 1e-5, (got, want)


def test_maxpool_bwd_ties_first_max():
    """All-equal windows: torch routes the gradient to the first element in scan order; so must we."""
    bt = Bench(0)
    shape = (1, 4, 4, 16)
    x = bt.act(shape, torch.ones(shape), 'x')
    dy = bt.act((1, 2, 2, 16), torch.arange(64.).reshape(1, 2, 2, 16) + 1, 'dy')
    y = bt.act(shape, None, 'y')
    op = G.Op('ew', op='maxpool_bwd', dims=shape, x=x, x2=None, dy=dy, add=None, y=y, out_stats=None, bstats=None,
              dgamma=None, dbeta=None, bn=None)
    bt.realise().run([op], 0)
    bt.compare(y, atol=0, rtol=0, label='maxpool ties')


@pytest.mark.parametrize('dtype', DTYPES)
@pytest.mark.parametrize('case', [(2, 16, 16, 16, 2, True), (3, 17, 12, 9, 1, True), (2, 16, 64, 64, 4, False)])
def test_loss(case, dtype):
    B, J, H, W, S, nchw = case
    gen = torch.Generator().manual_seed(13 + sum(case[:5]))
    dev = torch.device('cuda:0')
    tdt = torch.bfloat16 if dtype == 1 else torch.float32
    outs = [rnd(gen, B, H, W, J).to(tdt) for _ in range(S)]
    teacher = rnd(gen, B, H, W, J).to(tdt)
    target = torch.rand(B, J, H, W, generator=gen)
    wgt = (torch.rand(B, J, generator=gen) * 1.5) * (torch.rand(B, J, generator=gen) < 0.8)
    alpha = 0.3
    # specification (include/fpd_amd.h fpd_loss_t), fp64
    g = target.permute(0, 2, 3, 1).double()
    w2 = (wgt.double() ** 2)[:, None, None, :]
    cnt = B * J * H * W
    pose = sum(0.5 * (w2 * (o.double() - g) ** 2).sum() / cnt for o in outs)
    kd = sum(0.5 * (w2 * (o.double() - teacher.double()) ** 2).sum() / cnt for o in outs)
    grads = [w2 * ((1 - alpha) * (o.double() - g) + alpha * (o.double() - teacher.double())) / cnt for o in outs]
    a = R.LossT()
    a.B, a.J, a.H, a.W, a.S, a.dtype, a.target_nchw, a.alpha = B, J, H, W, S, dtype, 1 if nchw else 0, alpha
    d_outs = [o.to(dev) for o in outs]
    d_douts = [torch.zeros_like(o) for o in d_outs]
    for i in range(S):
        a.out[i], a.dout[i] = d_outs[i].data_ptr(), d_douts[i].data_ptr()
    d_t = teacher.to(dev)
    d_tg = (target if nchw else target.permute(0, 2, 3, 1).contiguous()).to(dev)
    d_w = wgt.float().to(dev)
    losses = torch.zeros(2, dtype=torch.float64, device=dev)
    a.teacher, a.target, a.weight, a.losses, a.grad_scale = d_t.data_ptr(), d_tg.data_ptr(), d_w.data_ptr(), losses.data_ptr(), 1.0
    R.check(R.lib().fpd_loss(a, R.current_stream()))
    torch.cuda.synchronize()
    l = losses.cpu()
    assert abs(l[0].item() - pose.item()) < 1e-6 * max(1, pose.item()), (l, pose)
    assert abs(l[1].item() - kd.item()) < 1e-6 * max(1, kd.item()), (l, kd)
    for i in range(S):
        got = d_douts[i].cpu().double()
        tol = 1e-9 if dtype == 0 else 1e-2 * grads[i].abs().max().item()
        assert (got - grads[i]).abs().max().item() <= tol + 1e-6 * grads[i].abs().max().item()


def test_adam_matches_torch():
    gen = torch.Generator().manual_seed(1)
    n = 100003
    dev = torch.device('cuda:0')
    p0 = rnd(gen, n)
    ref = torch.nn.Parameter(p0.clone())
    opt = torch.optim.Adam([ref], lr=2.5e-4)
    p = p0.clone().to(dev)
    m, v = torch.zeros(n, device=dev), torch.zeros(n, device=dev)
    step = torch.zeros(1, dtype=torch.int64, device=dev)
    lr = torch.full((1,), 2.5e-4, device=dev)
    for it in range(3):
        g = rnd(gen, n, scale=0.01)
        ref.grad = g.clone()
        opt.step()
        gd = g.to(dev)
        a = R.AdamT()
        a.n, a.param, a.grad, a.m, a.v = n, p.data_ptr(), gd.data_ptr(), m.data_ptr(), v.data_ptr()
        a.lr, a.beta1, a.beta2, a.eps, a.bias_corr1, a.bias_corr2, a.grad_scale = 0.0, 0.9, 0.999, 1e-8, 1.0, 1.0, 1.0
        a.lr_dev, a.step_dev = lr.data_ptr(), step.data_ptr()
        R.check(R.lib().fpd_adam(a, R.current_stream()))
    torch.cuda.synchronize()
    assert int(step.item()) == 3
    assert (p.cpu() - ref.detach()).abs().max().item() < 2e-7


def test_bn_update_running_and_layout_and_cast():
    gen = torch.Generator().manual_seed(2)
    bt = Bench(0)
    C = 48
    x = rnd(gen, 2, 5, 5, C) + 1
    bn = make_bn(bt, gen, C, 'train')
    bn.count = 50
    bn.stats = bt.buf('stats', (RS, 2, C), tensor_stats(x))
    bt.realise().run([G.Op('bnupd', bns=[bn])], 0)
    bt.compare(bn.rmean, atol=1e-6, rtol=1e-6, label='running_mean')
    bt.compare(bn.rvar, atol=1e-6, rtol=1e-6, label='running_var')
    assert int(bt.gpu.view(bn.nbt).item()) == 1
    # layout + cast round trips
    dev = torch.device('cuda:0')
    l, st = R.lib(), R.current_stream()
    src = rnd(gen, 2, 17, 6, 5).to(dev)
    for dtype, tdt in ((0, torch.float32), (1, torch.bfloat16)):
        nhwc = torch.empty((2, 6, 5, 17), dtype=tdt, device=dev)
        R.check(l.fpd_nchw_to_nhwc(src.data_ptr(), nhwc.data_ptr(), 2, 17, 6, 5, dtype, st))
        back = torch.empty_like(src)
        R.check(l.fpd_nhwc_to_nchw(nhwc.data_ptr(), back.data_ptr(), 2, 17, 6, 5, dtype, st))
        torch.cuda.synchronize()
        assert torch.equal(nhwc.float(), src.permute(0, 2, 3, 1).to(tdt).float())
        assert torch.equal(back, src.to(tdt).float())
    b16 = torch.empty(src.numel(), dtype=torch.bfloat16, device=dev)
    R.check(l.fpd_cast(src.data_ptr(), b16.data_ptr(), src.numel(), 0, 1, st))
    torch.cuda.synchronize()
    assert torch.equal(b16, src.reshape(-1).to(torch.bfloat16))


def test_error_paths_fail_loudly():
    a = R.ConvT()
    rc = R.lib().fpd_conv_forward(a, R.current_stream())
    assert rc != 0 and b'null' in R.lib().fpd_last_error()


# ---- fused frozen Bottleneck (fpd_bottleneck_forward) -------------------------------------------------------------
BNECK_CASES = [
    # N, H, W, P   (C = 2P)
    (2, 64, 64, 128),      # teacher level 64^2: two halo passes, tiles of two image rows
    (3, 32, 32, 128),
    (2, 16, 16, 128),      # image = two tiles
    (4, 8, 8, 128),        # tile = two whole images
    (5, 4, 4, 128),        # 80 pixels: one ragged tile of five whole images
    (33, 4, 4, 128),       # several tiles, last one ragged
    (2, 16, 16, 64),       # student widths
    (1, 64, 64, 64),
]


@pytest.mark.parametrize('fold', [False, True])
@pytest.mark.parametrize('case', BNECK_CASES)
def test_bottleneck_fused(case, fold):
    """Fused kernel vs its CPU specification (oracle/plan_interp.run_bneck): bf16 storage, intermediates rounded once;
    tolerance = bf16 output rounding (2^-8 relative) plus the occasional intermediate that rounds the other way."""
    N, H, W, P = case
    C = 2 * P
    gen = torch.Generator().manual_seed(hash(case) % (2 ** 31))
    b = Bench(1)
    x = b.act((N, H, W, C), rnd(gen, N, H, W, C))
    y = b.act((N, H, W, C), torch.zeros(N, H, W, C))
    w1 = b.buf('wlp', (P, 1, 1, C), rnd(gen, P, 1, 1, C, scale=(2.0 / C) ** 0.5))
    w2 = b.buf('wlp', (P, 3, 3, P), rnd(gen, P, 3, 3, P, scale=(2.0 / (9 * P)) ** 0.5))
    w3 = b.buf('wlp', (C, 1, 1, P), rnd(gen, C, 1, 1, P, scale=(2.0 / P) ** 0.5))
    b1, b2, b3 = (b.buf('param', (n,), 0.1 * rnd(gen, n)) for n in (P, P, C))
    bn1, bn2, bn3 = make_bn(b, gen, C, 'eval', 'bn1'), make_bn(b, gen, P, 'eval', 'bn2'), make_bn(b, gen, P, 'eval', 'bn3')
    op = G.Op('bneck', x=x, y=y, dims=(N, H, W, C, P), w1=w1, b1=b1, w2=w2, b2=b2, w3=w3, b3=b3, bn1=bn1, bn2=bn2, bn3=bn3)
    ops = [op]
    if fold:                  # tables folded once by fpd_bottleneck_fold() instead of by every block
        op.folded = b.buf('fold', (3 * C + 4 * P,), torch.full((3 * C + 4 * P,), float('nan')))
        ops = [G.Op('bneck_fold', target=op), op]
    b.realise()
    b.run(ops, 0)
    b.compare(y, 3e-2, 2e-2, 'bneck %r' % (case,))
    ref = b.gpu.view(y.buf).float().cpu()
    spec = b.cpu.view(y.buf).float()
    rel = float((ref - spec).norm() / spec.norm())
    assert rel < 3e-3, 'bneck %r: relative L2 vs specification %.3e' % (case, rel)


def test_bottleneck_rejects_unsupported():
    """Outside its domain the entry point fails loudly (callers issue the three conv launches instead)."""
    a = R.BneckT()
    a.N, a.H, a.W, a.C, a.P, a.dtype = 1, 8, 8, 256, 128, 0      # fp32 build is not supported
    t = torch.zeros(8 * 8 * 256, device='cuda')
    for f in ('x', 'w1', 'w2', 'w3'):
        setattr(a, f, t.data_ptr())
    a.y = torch.zeros(8 * 8 * 256, device='cuda').data_ptr()
    rc = R.lib().fpd_bottleneck_forward(a, None)
    assert rc != 0 and b'bn1' in R.lib().fpd_last_error()        # BN descriptors missing


# ---- two independent convolutions in one launch (fpd_conv_forward_pair) --------------------------------------------
PAIR_CASES = [
    # N, H, W (of a; b runs at H/2 x W/2), C, K, R, bn_mode, epi
    (2, 32, 32, 64, 64, 3, 'train', 'plain'),
    (2, 16, 16, 128, 64, 1, 'train', 'plain'),
    (3, 8, 8, 64, 128, 1, 'train', 'plain'),
    (2, 16, 16, 64, 64, 3, None, 'bnrelu_bwd'),         # the paired data gradients of the backward pass
    (2, 64, 64, 64, 64, 3, 'train', 'plain'),
]


@pytest.mark.parametrize('backend', [0, 'pp', ('pp', 6)])
@pytest.mark.parametrize('dtype', DTYPES)
@pytest.mark.parametrize('case', PAIR_CASES)
def test_conv_pair(case, dtype, backend):
    """'conv2' op (one launch) vs the interpreter running the two convolutions one after the other."""
    N, H, W, C, K, R, bnm, epi = case
    gen = torch.Generator().manual_seed(hash(case) % (2 ** 31))
    b = Bench(dtype)
    subs = []
    rd = (lambda t: t.to(torch.bfloat16).float()) if dtype == 1 else (lambda t: t)
    for (h, w) in ((H, W), (H // 2, W // 2)):
        x_val = rnd(gen, N, h, w, C) + 0.3
        x = b.act((N, h, w, C), x_val)
        y = b.act((N, h, w, K), torch.zeros(N, h, w, K))
        wt = b.buf('wlp', (K, R, R, C), rnd(gen, K, R, R, C, scale=(2.0 / (R * R * C)) ** 0.5))
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
                         dims=(N, h, w, C, K, R, R, 1, (R - 1) // 2, h, w), **kw))
    op = G.Op('conv2', a=subs[0], b=subs[1])
    b.realise()
    b.run([op], backend)
    for i, s in enumerate(subs):
        b.compare(s.y, label='pair[%d] y %r' % (i, case), **TOL[dtype])
        st = s.out_stats if s.out_stats is not None else s.epi_stats
        b.compare(st, atol=TOL[dtype]['atol'] * N * s.dims[1] * s.dims[2], rtol=TOL[dtype]['rtol'], label='pair[%d] stats %r' % (i, case))


def test_bottleneck_pair():
    """'bneck2' op: two fused frozen Bottlenecks (full and half resolution) in one launch vs the specification."""
    N, P = 3, 128
    C = 2 * P
    gen = torch.Generator().manual_seed(11)
    b = Bench(1)
    subs = []
    for hw in (16, 8):
        x = b.act((N, hw, hw, C), rnd(gen, N, hw, hw, C))
        y = b.act((N, hw, hw, C), torch.zeros(N, hw, hw, C))
        w1 = b.buf('wlp', (P, 1, 1, C), rnd(gen, P, 1, 1, C, scale=(2.0 / C) ** 0.5))
        w2 = b.buf('wlp', (P, 3, 3, P), rnd(gen, P, 3, 3, P, scale=(2.0 / (9 * P)) ** 0.5))
        w3 = b.buf('wlp', (C, 1, 1, P), rnd(gen, C, 1, 1, P, scale=(2.0 / P) ** 0.5))
        b1, b2, b3 = (b.buf('param', (n,), 0.1 * rnd(gen, n)) for n in (P, P, C))
        subs.append(G.Op('bneck', x=x, y=y, dims=(N, hw, hw, C, P), w1=w1, b1=b1, w2=w2, b2=b2, w3=w3, b3=b3,
                         bn1=make_bn(b, gen, C, 'eval', 'bn1'), bn2=make_bn(b, gen, P, 'eval', 'bn2'),
                         bn3=make_bn(b, gen, P, 'eval', 'bn3')))
    b.realise()
    b.run([G.Op('bneck2', a=subs[0], b=subs[1])], 0)
    for i, s_ in enumerate(subs):
        b.compare(s_.y, 3e-2, 2e-2, 'bneck pair[%d]' % i)


@pytest.mark.parametrize('dtype', DTYPES)
def test_pck_metric(dtype):
    """fpd_pck (arg-max + PCK@0.5 on the device) against the goldens written by the REFERENCE's own evaluate.accuracy
    (tests/golden/pck_ref.npz; square, 64x48 and 96x72 maps, ties, non-positive maps, corner targets, joints without a
    valid sample): (avg_acc, cnt) bit-exact in fp32.  bf16 storage rounds the prediction, so there the checker is the
    oracle restatement (itself pinned to the same goldens) on the rounded maps.  The loss accumulators handed to the
    kernel come back in the same log entry."""
    import numpy as np
    from fpd_amd.lib.core.evaluate import DeviceAccuracy
    from oracle import pck_ref
    g = np.load(os.path.join(ROOT, 'tests', 'golden', 'pck_ref.npz'))
    tdt = torch.bfloat16 if dtype == 1 else torch.float32
    dev = torch.device('cuda:0')
    for name in sorted({k.split('/')[0] for k in g.files}):
        out, tg = torch.from_numpy(g[name + '/output'].astype(np.float32)), torch.from_numpy(g[name + '/target'].astype(np.float32))
        B, J, H, W = out.shape
        if dtype == 1:
            _, avg, cnt, _ = pck_ref.accuracy(out.to(tdt).float().numpy(), tg.numpy())
        else:
            avg, cnt = float(g[name + '/avg_acc']), int(g[name + '/cnt'])
        out_nhwc = out.permute(0, 2, 3, 1).contiguous().to(tdt).to(dev)
        tgt_d = tg.to(dev)
        losses = torch.tensor([0.125, 0.75], dtype=torch.float64, device=dev)
        m = DeviceAccuracy(B, J, H, W, dtype, dev, slots=8).bind(out_nhwc.data_ptr(), tgt_d.data_ptr(), losses.data_ptr())
        m.enqueue(); m.enqueue()
        assert m.pending() == 2
        got = m.drain(full=True)
        assert len(got) == 2 and got[0] == got[1], got            # counters are re-zeroed between calls
        assert got[0][1] == cnt and got[0][0] == avg, (name, got[0], avg, cnt)
        assert got[0][2:] == (0.125, 0.75)
        assert m.drain() == []


@pytest.mark.parametrize('case', [(2, 16, 16, True), (3, 8, 8, True), (2, 16, 16, False), (1, 64, 64, True)])
def test_head_fused(case):
    """fpd_head_forward vs its CPU specification (oracle/plan_interp.run_head), inner stack (next) and last stack."""
    N, H, W, has_next = case
    C, J = 256, 16
    gen = torch.Generator().manual_seed(hash(case) % (2 ** 31))
    b = Bench(1)
    y0 = b.act((N, H, W, C), rnd(gen, N, H, W, C))
    x = b.act((N, H, W, C), rnd(gen, N, H, W, C)) if has_next else None
    score = b.act((N, H, W, J), torch.zeros(N, H, W, J))
    nxt = b.act((N, H, W, C), torch.zeros(N, H, W, C)) if has_next else None
    w_fc = b.buf('wlp', (C, 1, 1, C), rnd(gen, C, 1, 1, C, scale=(2.0 / C) ** 0.5))
    w_sc = b.buf('wlp', (J, 1, 1, C), rnd(gen, J, 1, 1, C, scale=(1.0 / C) ** 0.5))
    w_fc2 = b.buf('wlp', (C, 1, 1, C), rnd(gen, C, 1, 1, C, scale=(1.0 / C) ** 0.5)) if has_next else None
    w_sc2 = b.buf('wlp', (C, 1, 1, J), rnd(gen, C, 1, 1, J, scale=(1.0 / J) ** 0.5)) if has_next else None
    b_fc, b_sc = b.buf('param', (C,), 0.1 * rnd(gen, C)), b.buf('param', (J,), 0.1 * rnd(gen, J))
    b_fc2 = b.buf('param', (C,), 0.1 * rnd(gen, C)) if has_next else None
    b_sc2 = b.buf('param', (C,), 0.1 * rnd(gen, C)) if has_next else None
    op = G.Op('head', y0=y0, x=x, score=score, next=nxt, dims=(N, H, W, C, J), w_fc=w_fc, b_fc=b_fc, w_score=w_sc, b_score=b_sc,
              w_fc2=w_fc2, b_fc2=b_fc2, w_score2=w_sc2, b_score2=b_sc2, bn=make_bn(b, gen, C, 'eval', 'fcbn'))
    for fold in (False, True):
        ops = [op]
        if fold:
            op.folded = b.buf('fold', (3 * C + 32,), torch.full((3 * C + 32,), float('nan')))
            ops = [G.Op('head_fold', target=op), op]
        b.realise()
        b.run(ops, 0)
        b.compare(score, 3e-2, 2e-2, 'head score %r fold=%s' % (case, fold))
        if has_next:
            b.compare(nxt, 3e-2, 2e-2, 'head next %r fold=%s' % (case, fold))
            spec, got = b.cpu.view(nxt.buf).float(), b.gpu.view(nxt.buf).float().cpu()
            assert float((got - spec).norm() / spec.norm()) < 3e-3
