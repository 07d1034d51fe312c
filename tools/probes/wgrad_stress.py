"""Stress one wgrad shape N times and report runs whose result deviates from the median run (race hunting)."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from fpd_amd import executor as E, graph as G, runtime as R
N, H, W, C, K, Rr, pad = [int(v) for v in (sys.argv[1:8] or [4, 32, 32, 64, 64, 3, 1])]
iters = int(sys.argv[8]) if len(sys.argv) > 8 else 300
dev = torch.device('cuda:0'); dtype = R.BF16
A = E.Arenas(dev, dtype)
M = N * H * W
for n_, s_ in {'act': M * C + M * K + 128, 'param': 2 * C, 'rstat': 2 * C, 'stats': 2 * C * G.STATS_REPLICAS, 'nbt': 4, 'grad': K * Rr * Rr * C + K}.items():
    A.alloc(n_, s_)
g = torch.Generator().manual_seed(0)
A.t['act'].copy_(torch.randn(A.t['act'].numel(), generator=g).to(torch.bfloat16))
A.t['param'][:C] = 1.0
x = G.Act((N, H, W, C)); x.buf = G.Buf('act', 0, x.shape)
dy = G.Act((N, H, W, K)); dy.buf = G.Buf('act', M * C, dy.shape)
bn = G.BN('bn', 'train', C, G.Buf('param', 0, (C,)), G.Buf('param', C, (C,)), G.Buf('rstat', 0, (C,)), G.Buf('rstat', C, (C,)), G.Buf('nbt', 0, ()))
bn.count = M; bn.stats = G.Buf('stats', 0, (G.STATS_REPLICAS, 2, C))
xv = A.view(x.buf).double(); st0 = torch.zeros(bn.stats.shape, dtype=torch.float64); st0[0] = torch.stack([xv.sum((0, 1, 2)), (xv * xv).sum((0, 1, 2))]).cpu(); A.stats_write(bn.stats, st0)
op = G.Op('wgrad', x=x, dy=dy, dw=G.Buf('grad', 0, (K, Rr, Rr, C)), dbias=G.Buf('grad', K * Rr * Rr * C, (K,)), bn=bn, dims=(N, H, W, C, K, Rr, Rr, 1, pad, H, W))
plan = R.Plan(); plan.add(*E.Lowering(A, dtype).op(op))
res = []
for it in range(iters):
    A.t['grad'].zero_()
    plan.run(0, 1)
    torch.cuda.synchronize()
    res.append(A.t['grad'][:K * Rr * Rr * C].clone())
ref = torch.stack(res[:9]).median(0).values
bad = 0
for it, r in enumerate(res):
    e = (r - ref).abs().max().item()
    if e > 1e-2 * ref.abs().max().item():
        bad += 1
        d = (r - ref).abs().view(K, Rr, Rr, C)
        idx = torch.nonzero(d > 1e-2 * ref.abs().max().item())
        print('iter', it, 'max err', e, 'n bad', len(idx), 'k range', idx[:, 0].min().item(), idx[:, 0].max().item(), 'taps', sorted(set((idx[:, 1] * Rr + idx[:, 2]).tolist())), 'c range', idx[:, 3].min().item(), idx[:, 3].max().item())
print('bad runs: %d / %d' % (bad, iters))
