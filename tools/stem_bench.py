#!/usr/bin/env python
"""Micro-benchmark of the stem forward (7x7 stride-2 convolution of the fp32 NCHW image) at the benchmark shapes (GPU only).
   python tools/stem_bench.py [--iters 20]      (the im2col kernel stem_fwd_mfma now only serves the shapes stem_s2d declines)"""
import argparse, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fpd_amd import executor as E, graph as G, runtime as R


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--iters', type=int, default=20)
    args = ap.parse_args()
    dev = torch.device('cuda:0')
    l = R.lib()
    for name, N, H, W, K in (('student stem K=32', 32, 256, 256, 32), ('teacher stem K=64', 32, 256, 256, 64)):
        P, Q = H // 2, W // 2
        A = E.Arenas(dev, R.BF16)
        for n_, s_ in {'image': N * 3 * H * W, 'param': K * 147 + K + 64, 'act': N * P * Q * K + 64, 'stats': G.STATS_REPLICAS * 2 * K}.items():
            A.alloc(n_, s_)
        A.t['image'].normal_()
        A.t['param'].normal_(std=0.08)
        img = G.Buf('image', 0, (N, 3, H, W))
        w = G.Buf('param', 0, (K, 7, 7, 3))
        b = G.Buf('param', K * 147, (K,))
        y = G.Act((N, P, Q, K)); y.buf = G.Buf('act', 0, y.shape)
        st = G.Buf('stats', 0, (G.STATS_REPLICAS, 2, K))
        op = G.Op('stem_fwd', image=img, w=w, bias=b, y=y, out_stats=st, dims=(N, H, W, K, P, Q))
        low = E.Lowering(A, R.BF16)
        plan = R.Plan()
        plan.add(*low.op(op))
        s = R.current_stream()
        for _ in range(3):
            plan.run(0, 1, s)
        torch.cuda.synchronize()
        e0, e1 = l.fpd_event_create(), l.fpd_event_create()
        l.fpd_event_record(e0, s)
        for _ in range(args.iters):
            plan.run(0, 1, s)
        l.fpd_event_record(e1, s)
        ms = l.fpd_event_elapsed_ms(e0, e1) / args.iters
        mb = (N * 3 * H * W * 4 + N * P * Q * K * 2) / 1e6
        print('%-20s %8.1f us   %6.1f MB algorithmic -> %6.0f GB/s' % (name, ms * 1e3, mb, mb / ms), flush=True)
        if K <= 32:           # the weight gradient (student only: the teacher is frozen), with slabs as in a training plan
            A.alloc('grad', K * 147 + K + 64)
            A.t['act'].normal_(std=0.1)
            wop = G.Op('stem_wgrad', image=img, dy=y, dw=G.Buf('grad', 0, (K, 7, 7, 3)), dbias=G.Buf('grad', K * 147, (K,)), dims=(N, H, W, K, P, Q))
            low2 = E.Lowering(A, R.BF16)
            low2.use_partials = True
            lw = low2.op(wop)
            low2.finish_partials()
            plan2 = R.Plan()
            plan2.add(*lw)
            for _ in range(3):
                plan2.run(0, 1, s)
            torch.cuda.synchronize()
            l.fpd_event_record(e0, s)
            for _ in range(args.iters):
                plan2.run(0, 1, s)
            l.fpd_event_record(e1, s)
            ms = l.fpd_event_elapsed_ms(e0, e1) / args.iters
            print('%-20s %8.1f us   (weight gradient, %d slabs)' % (name, ms * 1e3, sum(r[4] for r in low2.partials.values())), flush=True)


if __name__ == '__main__':
    main()
