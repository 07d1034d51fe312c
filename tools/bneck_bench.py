"""Dev micro-benchmark: fused frozen Bottleneck vs the three conv launches it replaces (hipGraph-timed)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from fpd_amd import executor as E, graph as G, runtime as R
R.lib()
dev = torch.device('cuda', 0)
N, P = 32, int(os.environ.get('P', '128'))
C = 2 * P
for HW in ((int(os.environ['ONLY']),) if os.environ.get('ONLY') else (64, 32, 16, 8, 4)):
    A = E.Arenas(dev, R.BF16)
    sizes = {}
    def buf(arena, shape):
        n = 1
        for s in shape: n *= s
        off = sizes.get(arena, 0); sizes[arena] = off + (n + 63) // 64 * 64
        return G.Buf(arena, off, shape, arena)
    def act(shape):
        a = G.Act(shape); a.buf = buf('act', shape); return a
    x, t1, t2, y, y2 = act((N, HW, HW, C)), act((N, HW, HW, P)), act((N, HW, HW, P)), act((N, HW, HW, C)), act((N, HW, HW, C))
    w1, w2, w3 = buf('wlp', (P, 1, 1, C)), buf('wlp', (P, 3, 3, P)), buf('wlp', (C, 1, 1, P))
    b1, b2, b3 = buf('param', (P,)), buf('param', (P,)), buf('param', (C,))
    def bn(Cn):
        return G.BN('bn', 'eval', Cn, buf('param', (Cn,)), buf('param', (Cn,)), buf('rstat', (Cn,)), buf('rstat', (Cn,)), buf('nbt', ()))
    bn1, bn2, bn3 = bn(C), bn(P), bn(P)
    for k, n in sizes.items():
        A.alloc(k, n)
    A.tensor('act').normal_(); A.tensor('wlp').normal_(0, 0.05); A.tensor('param').fill_(1.0); A.tensor('rstat').fill_(1.0)
    fused = [G.Op('bneck', x=x, y=y, dims=(N, HW, HW, C, P), w1=w1, b1=b1, w2=w2, b2=b2, w3=w3, b3=b3, bn1=bn1, bn2=bn2, bn3=bn3)]
    def conv(xx, w, b, yy, bnn, K, Cc, Rr, res=None):
        return G.Op('conv', x=xx, w=w, wkey='', bias=b, bkey='', residual=res, y=yy, out_stats=None, bn=bnn, epi='plain',
                    epi_x=None, epi_bn=None, epi_stats=None, dims=(N, HW, HW, Cc, K, Rr, Rr, 1, (Rr - 1) // 2, HW, HW))
    three = [conv(x, w1, b1, t1, bn1, P, C, 1), conv(t1, w2, b2, t2, bn2, P, P, 3), conv(t2, w3, b3, y2, bn3, C, P, 1, res=x)]
    low = E.Lowering(A, R.BF16)
    res = {}
    for name, ops in (('fused', fused), ('3 convs', three)):
        plan = R.Plan()
        for op in ops:
            for _ in range(1):
                plan.add(*low.op(op))
        reps = 20
        st = torch.cuda.Stream()
        with torch.cuda.stream(st):
            plan.run(0, len(plan)); torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(st)
            for _ in range(reps):
                plan.run(0, len(plan))
            e1.record(st)
        torch.cuda.synchronize()
        res[name] = e0.elapsed_time(e1) / reps * 1e3
    fl = 2.0 * N * HW * HW * (C * P + 9 * P * P + P * C)
    d = (A.view(y.buf).float() - A.view(y2.buf).float()).norm() / A.view(y2.buf).float().norm()
    print('%2dx%-2d P=%d  fused %7.1f us (%6.1f TF/s)   3 convs %7.1f us   speedup %.2fx   rel-L2 diff %.1e' % (
        HW, HW, P, res['fused'], fl / res['fused'] / 1e6, res['3 convs'], res['3 convs'] / res['fused'], float(d)), flush=True)
