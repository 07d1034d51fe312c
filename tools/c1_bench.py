#!/usr/bin/env python
"""Micro-benchmark of the student's 1x1 and 3x3 convolutions at the benchmark's shapes: the streaming kernel (csrc/conv_c1.hip)
/ the strip kernel (csrc/conv_c3.hip) against the persistent one (csrc/conv_pp.hip), interleaved in one process (GPU only).
   python tools/c1_bench.py [--iters 30] [--rounds 3] [--only substr]
Per shape and kernel: HIP events on the launch stream around `iters` back-to-back launches; the algorithmic HBM bytes of
the launch (every tensor it must read or write once) and the fraction of 6.3 TB/s that time stands for."""
import argparse, os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fpd_amd import executor as E, graph as G, runtime as R

# name, N, H, W, C, K of the FORWARD convolution, kind: fwd / fwd+res / bwd (data gradient K -> C with fused weight gradient) / bwd+fold
SHAPES = [
    ('fwd 128>64 @64', 32, 64, 64, 128, 64, 'fwd'),
    ('fwd 64>128+res @64', 32, 64, 64, 64, 128, 'fwd+res'),
    ('fwd 128>128 @64', 32, 64, 64, 128, 128, 'fwd'),
    ('bwd+wg of 64>128 @64 (128>64)', 32, 64, 64, 64, 128, 'bwd'),
    ('bwd+wg+fold of 128>64 @64 (64>128)', 32, 64, 64, 128, 64, 'bwd+fold'),
    ('fwd 128>64 @32', 32, 32, 32, 128, 64, 'fwd'),
    ('fwd 64>128+res @32', 32, 32, 32, 64, 128, 'fwd+res'),
    ('bwd+wg of 64>128 @32', 32, 32, 32, 64, 128, 'bwd'),
    ('bwd+wg+fold of 128>64 @32', 32, 32, 32, 128, 64, 'bwd+fold'),
    ('fwd 32>64+res @128', 32, 128, 128, 32, 64, 'fwd+res'),
    ('fwd 128>64 @8', 32, 8, 8, 128, 64, 'fwd'),
    ('bwd+wg+fold of 128>64 @8', 32, 8, 8, 128, 64, 'bwd+fold'),
    ('fwd 128>64 @16', 32, 16, 16, 128, 64, 'fwd'),
    ('bwd+wg of 64>128 @16', 32, 16, 16, 64, 128, 'bwd'),
    ('bwd+wg+fold of 128>64 @16', 32, 16, 16, 128, 64, 'bwd+fold'),
    ('bwd+wg of 64>128 @8', 32, 8, 8, 64, 128, 'bwd'),
    ('fwd 64>128+res @8', 32, 8, 8, 64, 128, 'fwd+res'),
    ('fwd 128>64 @4', 32, 4, 4, 128, 64, 'fwd'),
    ('fwd 64>128+res @4', 32, 4, 4, 64, 128, 'fwd+res'),
    ('bwd+wg of 64>128 @4', 32, 4, 4, 64, 128, 'bwd'),
    ('bwd+wg+fold of 128>64 @4', 32, 4, 4, 128, 64, 'bwd+fold'),
    ('3x3 fwd 64>64 @64', 32, 64, 64, 64, 64, 'fwd', 3),
    ('3x3 bwd+fold 64>64 @64', 32, 64, 64, 64, 64, 'bwd+fold', 3),
    ('3x3 fwd 64>64 @32', 32, 32, 32, 64, 64, 'fwd', 3),
    ('3x3 bwd+fold 64>64 @32', 32, 32, 32, 64, 64, 'bwd+fold', 3),
    ('3x3 fwd 64>64 @16', 32, 16, 16, 64, 64, 'fwd', 3),
    ('3x3 bwd+fold 64>64 @16', 32, 16, 16, 64, 64, 'bwd+fold', 3),
]


def build(name, N, H, W, C, K, kind, dev, Rr=1):
    pad = (Rr - 1) // 2
    A = E.Arenas(dev, R.BF16)
    M = N * H * W
    RSn = G.STATS_REPLICAS
    sizes = {'act': M * (3 * C + 3 * K) + 1024, 'wlp': 2 * K * C * Rr * Rr, 'param': 8 * max(C, K) + K * C * Rr * Rr, 'rstat': 4 * max(C, K),
             'stats': RSn * 2 * (2 * C + 2 * K), 'nbt': 8, 'grad': K * C * Rr * Rr + 4 * max(C, K)}
    for n_, s_ in sizes.items():
        A.alloc(n_, s_)
    gen = torch.Generator().manual_seed(0)
    A.t['act'].copy_(torch.randn(A.t['act'].numel(), generator=gen).to(A.t['act'].dtype))
    A.t['wlp'].copy_((torch.randn(A.t['wlp'].numel(), generator=gen) / np.sqrt(C * Rr * Rr)).to(A.t['wlp'].dtype))
    A.t['param'].fill_(0.5)
    A.t['rstat'].fill_(1.0)
    off = [0]

    def act(ch):
        a = G.Act((N, H, W, ch))
        a.buf = G.Buf('act', off[0], a.shape)
        off[0] += M * ch
        return a

    def bn_of(t, ch, poff, soff):
        bn = G.BN('bn', 'train', ch, G.Buf('param', poff, (ch,)), G.Buf('param', poff + ch, (ch,)), G.Buf('rstat', 0, (ch,)),
                  G.Buf('rstat', ch, (ch,)), G.Buf('nbt', 0, ()))
        bn.count = M
        bn.stats = G.Buf('stats', soff, (RSn, 2, ch))
        xv = A.view(t.buf).double()
        st0 = torch.zeros(bn.stats.shape, dtype=torch.float64)
        st0[0] = torch.stack([xv.sum((0, 1, 2)), (xv * xv).sum((0, 1, 2))]).cpu()
        A.stats_write(bn.stats, st0)
        return bn

    low = E.Lowering(A, R.BF16)
    low.use_partials = True
    x = act(C)
    if kind.startswith('fwd'):
        y = act(K)
        r = act(K) if kind == 'fwd+res' else None
        bn = bn_of(x, C, 0, 0)
        op = G.Op('conv', x=x, w=G.Buf('wlp', 0, (K, Rr, Rr, C)), wkey='w', bias=G.Buf('param', 4 * max(C, K), (K,)), bkey='b',
                  residual=r, y=y, out_stats=G.Buf('stats', RSn * 2 * C, (RSn, 2, K)), bn=bn, epi='plain', epi_x=None,
                  epi_bn=None, epi_stats=None, dims=(N, H, W, C, K, Rr, Rr, 1, pad, H, W))
        ops = [op]
        nbytes = 2 * M * (C + K + (K if r is not None else 0))
    else:
        # data gradient of the forward convolution C -> K: operand dy [K], output dz [C], epi_x = x [C]
        bn = bn_of(x, C, 0, 0)
        dz = act(C)
        wb = G.Buf('wlp', K * C * Rr * Rr, (C, Rr, Rr, K))
        dw, db = G.Buf('grad', 0, (K, Rr, Rr, C)), G.Buf('grad', K * C * Rr * Rr, (K,))
        bst = G.Buf('stats', RSn * 2 * C, (RSn, 2, C))
        ops = []
        if kind == 'bwd+fold':
            u, g, du = act(K), act(K), act(K)
            bn2 = bn_of(u, K, 2 * max(C, K), RSn * 4 * C)
            bst2 = G.Buf('stats', RSn * 4 * C + RSn * 2 * K, (RSn, 2, K))
            ap = G.Op('ew', op='bn_bwd_apply', dims=(N, H, W, K), x=u, x2=None, dy=g, add=None, y=du, out_stats=None, bstats=bst2,
                      dgamma=G.Buf('grad', K * C * Rr * Rr + K, (K,)), dbeta=G.Buf('grad', K * C * Rr * Rr + 2 * K, (K,)), bn=bn2)
            ops.append(ap)
            dy = du
            nbytes = 2 * M * (2 * K + 2 * C + (K if Rr == 3 else 0))     # (3x3: + the materialised operand)
        else:
            dy = act(K)
            nbytes = 2 * M * (K + 2 * C)
        wg = G.Op('wgrad', x=x, dy=dy, dw=dw, dbias=db, bn=bn, dims=(N, H, W, C, K, Rr, Rr, 1, pad, H, W))
        dg = G.Op('conv', x=dy, w=wb, wkey='w', bias=None, bkey=None, residual=None, y=dz, out_stats=None, bn=None,
                  epi='bnrelu_bwd', epi_x=x, epi_bn=bn, epi_stats=bst, dims=(N, H, W, K, C, Rr, Rr, 1, Rr - 1 - pad, H, W))
        if Rr == 1:
            dg.fused_wgrad = wg
        if kind == 'bwd+fold':
            dg.fold_apply, dg.fold_wgrad = ap, wg
        ops += [dg, wg]
    low.plan_folds(ops)
    plan = R.Plan()
    lowered = [low.op(o) for o in ops]
    low.finish_partials()
    idx = None
    for i, (code, st_) in enumerate(lowered):
        plan.add(code, st_)
        if ops[i].kind == 'conv':
            idx = i
    return plan, idx, nbytes, A, low


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--iters', type=int, default=30)
    ap.add_argument('--rounds', type=int, default=3)
    ap.add_argument('--only', default='')
    args = ap.parse_args()
    dev = torch.device('cuda:0')
    l = R.lib()
    st = R.current_stream()
    for shp in SHAPES:
        if args.only and args.only not in shp[0]:
            continue
        res = {}
        plans = {}
        opt = 'conv_c3' if (len(shp) > 7 and shp[7] == 3) else 'conv_c1'
        for kern, mode in (('pp', 0), ('c1', 2)):
            prev = R.set_option(opt, mode)
            try:
                plans[kern] = build(*shp[:7], dev, *(shp[7:]))
            finally:
                R.set_option(opt, prev)
        for rnd_ in range(args.rounds):
            for kern in ('pp', 'c1'):
                plan, idx, nbytes, A, low = plans[kern]
                prev = R.set_option(opt, 0 if kern == 'pp' else 2)      # (the dispatcher decides at launch time)
                try:
                    n0 = R.set_option(opt + '_launches', 0)
                    for _ in range(3):
                        plan.run(idx, idx + 1, st)
                    torch.cuda.synchronize()
                    assert (R.set_option(opt + '_launches', 0) - n0 == 3) == (kern == 'c1'), 'wrong kernel served the launch'
                    e0, e1 = l.fpd_event_create(), l.fpd_event_create()
                    l.fpd_event_record(e0, st)
                    for _ in range(args.iters):
                        plan.run(idx, idx + 1, st)
                    l.fpd_event_record(e1, st)
                    ms = l.fpd_event_elapsed_ms(e0, e1) / args.iters
                finally:
                    R.set_option(opt, prev)
                res.setdefault(kern, []).append(ms * 1e3)
        nbytes = plans['pp'][2]
        floor = nbytes / 6.3e12 * 1e6
        line = '%-38s %6.1f MB (%.1f us @6.3 TB/s) ' % (shp[0], nbytes / 1e6, floor)
        for kern in ('pp', 'c1'):
            v = res[kern]
            line += ' %s %6.1f us (min %6.1f, %.2f of the HBM roofline)' % ('new' if kern == 'c1' else kern, float(np.median(v)), min(v), floor / min(v))
        print(line, flush=True)


if __name__ == '__main__':
    main()
