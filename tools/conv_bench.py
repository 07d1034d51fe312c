#!/usr/bin/env python
"""Micro-benchmark of the conv / wgrad kernels on the layer shapes of the FPD pair (GPU only).
   python tools/conv_bench.py [--dtype bf16|fp32] [--iters 20]
Prints per-shape time (HIP events on the launch stream) and algorithmic TFLOP/s."""
import argparse, os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fpd_amd import executor as E, graph as G, runtime as R

SHAPES = [
    # name, N, H, W, C, K, R, pad, bn(train/eval/None), residual
    ('t 3x3 128>128 @64', 32, 64, 64, 128, 128, 3, 1, 'eval', False),
    ('t 1x1 256>128 @64', 32, 64, 64, 256, 128, 1, 0, 'eval', False),
    ('t 1x1 128>256 @64', 32, 64, 64, 128, 256, 1, 0, 'eval', True),
    ('s 3x3 64>64 @64', 32, 64, 64, 64, 64, 3, 1, 'train', False),
    ('s 1x1 128>64 @64', 32, 64, 64, 128, 64, 1, 0, 'train', False),
    ('s 1x1 64>128 @64', 32, 64, 64, 64, 128, 1, 0, 'train', True),
    ('t 3x3 128>128 @32', 32, 32, 32, 128, 128, 3, 1, 'eval', False),
    ('t 3x3 128>128 @16', 32, 16, 16, 128, 128, 3, 1, 'eval', False),
    ('t 3x3 128>128 @8', 32, 8, 8, 128, 128, 3, 1, 'eval', False),
    ('t 3x3 128>128 @4', 32, 4, 4, 128, 128, 3, 1, 'eval', False),
    ('t 1x1 256>128 @4', 32, 4, 4, 256, 128, 1, 0, 'eval', False),
    ('s 3x3 64>64 @32', 32, 32, 32, 64, 64, 3, 1, 'train', False),
    ('s 3x3 64>64 @16', 32, 16, 16, 64, 64, 3, 1, 'train', False),
    ('s 3x3 64>64 @8', 32, 8, 8, 64, 64, 3, 1, 'train', False),
    ('s 3x3 64>64 @4', 32, 4, 4, 64, 64, 3, 1, 'train', False),
    ('s 1x1 128>64 @8', 32, 8, 8, 128, 64, 1, 0, 'train', False),
    ('s 1x1 64>128 @8', 32, 8, 8, 64, 128, 1, 0, 'train', True),
    ('l1 3x3 32>32 @128', 32, 128, 128, 32, 32, 3, 1, 'train', False),
]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--dtype', default='bf16')
    ap.add_argument('--iters', type=int, default=20)
    ap.add_argument('--wgrad', action='store_true')
    ap.add_argument('--only', default='')
    ap.add_argument('--partials', action='store_true', help='weight gradients with slabs (what a training plan uses), reduction not timed')
    ap.add_argument('--no-stats', action='store_true', help='ablation: do not accumulate output statistics')
    ap.add_argument('--graph', action='store_true', help='time a hipGraph of --iters copies (device-side time per launch)')
    args = ap.parse_args()
    dtype = R.BF16 if args.dtype == 'bf16' else R.F32
    dev = torch.device('cuda:0')
    l = R.lib()
    gen = torch.Generator().manual_seed(0)
    for (name, N, H, W, C, K, Rr, pad, bnm, res) in SHAPES:
        if args.only and args.only not in name:
            continue
        A = E.Arenas(dev, dtype)
        P, Q = H + 2 * pad - Rr + 1, W + 2 * pad - Rr + 1
        sizes = {'act': N * H * W * C + 2 * N * P * Q * K + 256, 'wlp': K * Rr * Rr * C, 'param': 4 * max(C, K) + K * Rr * Rr * C,
                 'rstat': 2 * C, 'stats': G.STATS_REPLICAS * (2 * C + 2 * K), 'nbt': 4, 'grad': K * Rr * Rr * C + K}
        for n_, s_ in sizes.items():
            A.alloc(n_, s_)
        A.t['act'].copy_(torch.randn(A.t['act'].numel(), generator=gen).to(A.t['act'].dtype))
        A.t['wlp'].copy_((torch.randn(A.t['wlp'].numel(), generator=gen) / np.sqrt(C * Rr * Rr)).to(A.t['wlp'].dtype))
        A.t['param'][:2 * C] = 1.0
        A.t['rstat'][:C] = 0.0
        A.t['rstat'][C:] = 1.0
        x = G.Act((N, H, W, C)); x.buf = G.Buf('act', 0, x.shape)
        y = G.Act((N, P, Q, K)); y.buf = G.Buf('act', N * H * W * C, y.shape)
        r = None
        if res:
            r = G.Act((N, P, Q, K)); r.buf = G.Buf('act', N * H * W * C + N * P * Q * K, r.shape)
        bn = None
        if bnm:
            bn = G.BN('bn', bnm, C, G.Buf('param', 0, (C,)), G.Buf('param', C, (C,)), G.Buf('rstat', 0, (C,)),
                      G.Buf('rstat', C, (C,)), G.Buf('nbt', 0, ()))
            bn.count = N * H * W
            if bnm == 'train':
                bn.stats = G.Buf('stats', 0, (G.STATS_REPLICAS, 2, C))
                xv = A.view(x.buf).double()
                st0 = torch.zeros(bn.stats.shape, dtype=torch.float64)
                st0[0] = torch.stack([xv.sum((0, 1, 2)), (xv * xv).sum((0, 1, 2))]).cpu()
                A.stats_write(bn.stats, st0)
        low = E.Lowering(A, dtype)
        if args.wgrad:
            op = G.Op('wgrad', x=x, dy=y, dw=G.Buf('grad', 0, (K, Rr, Rr, C)), dbias=G.Buf('grad', K * Rr * Rr * C, (K,)), bn=bn,
                      dims=(N, H, W, C, K, Rr, Rr, 1, pad, P, Q))
        else:
            op = G.Op('conv', x=x, w=G.Buf('wlp', 0, (K, Rr, Rr, C)), wkey='w', bias=G.Buf('param', 2 * C, (K,)), bkey='b', residual=r, y=y,
                      out_stats=G.Buf('stats', G.STATS_REPLICAS * 2 * C, (G.STATS_REPLICAS, 2, K)) if (bnm == 'train' and not args.no_stats) else None, bn=bn, epi='plain', epi_x=None,
                      epi_bn=None, epi_stats=None, dims=(N, H, W, C, K, Rr, Rr, 1, pad, P, Q))
        plan = R.Plan()
        reps = args.iters if args.graph else 1
        low.use_partials = args.partials and args.wgrad
        lowered = [low.op(op) for _ in range(1)] * reps
        low.finish_partials()
        for code, st_ in lowered:
            plan.add(code, st_)
        st = R.current_stream()
        for _ in range(3):
            plan.run(0, 1, st)
        torch.cuda.synchronize()
        e0, e1 = l.fpd_event_create(), l.fpd_event_create()
        if args.graph:
            import ctypes
            cs = torch.cuda.Stream()
            with torch.cuda.stream(cs):
                gid = plan.capture(0, reps, ctypes.c_void_p(cs.cuda_stream))
            torch.cuda.synchronize()
            plan.replay(gid, st)
            torch.cuda.synchronize()
            l.fpd_event_record(e0, st)
            for _ in range(5):
                plan.replay(gid, st)
            l.fpd_event_record(e1, st)
            ms = l.fpd_event_elapsed_ms(e0, e1) / (5 * reps)
        else:
            l.fpd_event_record(e0, st)
            for _ in range(args.iters):
                plan.run(0, 1, st)
            l.fpd_event_record(e1, st)
            ms = l.fpd_event_elapsed_ms(e0, e1) / args.iters
        fl = 2.0 * N * P * Q * K * C * Rr * Rr
        extra = ''
        if args.wgrad and args.partials:
            extra = '  slabs %d' % sum(r[4] for r in low.partials.values())
        print('%-22s %s %8.1f us  %7.1f TFLOP/s%s' % (name, 'wgrad' if args.wgrad else 'conv ', ms * 1e3, fl / ms / 1e9, extra), flush=True)


if __name__ == '__main__':
    main()
