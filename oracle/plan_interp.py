"""CPU interpreter of the product's op list (test infrastructure -- never imported by the product).

Executes the IR built by `fast-human-pose-estimation.pytorch_amd/graph.py` with plain torch CPU ops
over the SAME arenas/offsets the GPU executor uses.  Two purposes:
  1. validate the host logic (graph wiring, fused-BN algebra, reverse-mode construction, memory
     planning/aliasing) against the reference-pinned oracle without a GPU;
  2. serve as the per-op specification the HIP kernels are compared with in the `-m gpu` tests.
Each op's semantics restate include/fpd_amd.h; the arithmetic they replace is the torch.nn graph of
/root/reference/lib/models/hourglass.py and its autograd.
"""
import torch
import torch.nn.functional as F

EPS = 1e-5


class Arenas:
    """name -> flat CPU tensor; `view(buf)` returns the tensor region of a graph.Buf."""

    def __init__(self, sizes, act_dtype=torch.float32):
        self.t = {}
        self.act_dtype = act_dtype
        for name, n in sizes.items():
            if name in ('act', 'wlp'):
                dt = act_dtype
            elif name in ('stats', 'losses'):
                dt = torch.float64
            elif name == 'nbt':
                dt = torch.int64
            else:
                dt = torch.float32
            self.t[name] = torch.zeros(max(int(n), 1), dtype=dt)

    def view(self, buf):
        return self.t[buf.arena][buf.off:buf.off + buf.numel].view(buf.shape)


def _bn_coef(A, bn):
    gamma, beta = A.view(bn.gamma).double(), A.view(bn.beta).double()
    if bn.mode == 'train':
        st = A.view(bn.stats).sum(0)              # [R][2][C] replicas -> [2][C]
        mean = st[0] / bn.count
        var = (st[1] / bn.count - mean * mean).clamp_min(0)
    else:
        mean, var = A.view(bn.rmean).double(), A.view(bn.rvar).double()
    invstd = 1.0 / torch.sqrt(var + EPS)
    scale = (gamma * invstd).float()
    shift = (beta - mean * gamma * invstd).float()
    return scale, shift, mean.float(), invstd.float()


def _act(A, a):
    return A.view(a.buf).float()


def _store(A, a, val):
    A.view(a.buf).copy_(val.to(A.act_dtype))


def _rnd(A, v):
    return v.to(A.act_dtype).float()


def _stats_add(A, buf, v):          # v: [N,H,W,C] fp32 already rounded to storage precision
    st = A.view(buf)[0]                          # the interpreter accumulates into replica 0 of [R][2][C]
    vd = v.double()
    st[0] += vd.sum((0, 1, 2))
    st[1] += (vd * vd).sum((0, 1, 2))


def _prologue(A, x, bn):
    if bn is None:
        return x
    scale, shift, _, _ = _bn_coef(A, bn)
    y = torch.addcmul(shift, x, scale)
    if bn.relu:
        y = y.clamp_min(0)
    return _rnd(A, y)


def A_master(op):
    """The fp32 master weight Buf behind a convolution's working copy (same shape, 'param' arena)."""
    return op.w_master if getattr(op, 'w_master', None) is not None else op.w


def run_conv(A, op):
    n, h, w, C, K, R, S, stride, pad, P, Q = op.dims
    if getattr(op, 'w8', None) is not None:       # fp8 forward convolution (include/fpd_amd.h fpd_conv_f8_t; oracle/fp8_ref.py):
        from . import fp8_ref                     # prologue result rounded ONCE, to e4m3; e4m3 weights of the fp32 masters
        x = _act(A, op.x)
        if op.bn is not None:
            scale, shift, _, _ = _bn_coef(A, op.bn)
            x = torch.addcmul(shift, x, scale)
            if op.bn.relu:
                x = x.clamp_min(0)
        y = fp8_ref.conv_f8(x, A.view(A_master(op)).float(), stride, pad)
    else:
        x = _prologue(A, _act(A, op.x), op.bn)
        wt = A.view(op.w).float()                     # [K,R,S,C]
        y = F.conv2d(x.permute(0, 3, 1, 2), wt.permute(0, 3, 1, 2), None, stride=stride, padding=pad).permute(0, 2, 3, 1)
    if op.bias is not None:
        y = y + A.view(op.bias)
    if op.residual is not None:
        y = y + _act(A, op.residual)
    if op.epi == 'bnrelu_bwd':
        xv = _act(A, op.epi_x)
        scale, shift, mean, invstd = _bn_coef(A, op.epi_bn)
        z = torch.addcmul(shift, xv, scale)
        if op.epi_bn.relu:
            y = torch.where(z > 0, y, torch.zeros_like(y))
        yr = _rnd(A, y).double()
        st = A.view(op.epi_stats)[0]
        st[0] += yr.sum((0, 1, 2))
        st[1] += (yr * ((xv - mean) * invstd).double()).sum((0, 1, 2))
    elif op.out_stats is not None:
        _stats_add(A, op.out_stats, _rnd(A, y))
    _store(A, op.y, y)


def run_bneck(A, op):
    """Fused frozen Bottleneck (include/fpd_amd.h fpd_bneck_t; /root/reference/lib/models/hourglass.py:32-52 in eval
    mode).  Each intermediate is rounded to the storage precision ONCE, after bias + BN + ReLU."""
    def conv(v, wbuf, pad):
        wt = A.view(wbuf).float().permute(0, 3, 1, 2)
        return F.conv2d(v.permute(0, 3, 1, 2), wt, None, stride=1, padding=pad).permute(0, 2, 3, 1)

    def bn_act(v, bn, bias):
        scale, shift, _, _ = _bn_coef(A, bn)
        if bias is not None:
            shift = torch.addcmul(shift, scale, A.view(bias))      # kernel folds the conv bias into the BN shift
        return _rnd(A, torch.addcmul(shift, v, scale).clamp_min(0))
    x = _act(A, op.x)
    a1 = bn_act(x, op.bn1, None)
    a2 = bn_act(conv(a1, op.w1, 0), op.bn2, op.b1)
    a3 = bn_act(conv(a2, op.w2, 1), op.bn3, op.b2)
    y = conv(a3, op.w3, 0) + x
    if op.b3 is not None:
        y = y + A.view(op.b3)
    _store(A, op.y, y)


def run_head(A, op):
    """Fused frozen inter-stack head (include/fpd_amd.h fpd_head_t; /root/reference/lib/models/hourglass.py:134-137,
    184-190 in eval mode): a = relu(bn(fc(y0))) is rounded to the storage precision once and never stored; score is
    rounded once (it is an output and the operand of score_)."""
    def conv1x1(v, wbuf):
        wt = A.view(wbuf).float().permute(0, 3, 1, 2)
        return F.conv2d(v.permute(0, 3, 1, 2), wt, None).permute(0, 2, 3, 1)
    y0 = _act(A, op.y0)
    scale, shift, _, _ = _bn_coef(A, op.bn)
    shift = torch.addcmul(shift, scale, A.view(op.b_fc))
    a = _rnd(A, torch.addcmul(shift, conv1x1(y0, op.w_fc), scale).clamp_min(0))
    score = _rnd(A, conv1x1(a, op.w_score) + A.view(op.b_score))
    _store(A, op.score, score)
    if op.next is not None:
        out = conv1x1(a, op.w_fc2) + conv1x1(score, op.w_score2) + (A.view(op.b_fc2) + A.view(op.b_score2)) + _act(A, op.x)
        _store(A, op.next, out)


def run_wgrad(A, op):
    n, h, w, C, K, R, S, stride, pad, P, Q = op.dims
    x = _prologue(A, _act(A, op.x), op.bn).permute(0, 3, 1, 2)
    dy = _act(A, op.dy).permute(0, 3, 1, 2)
    dw = torch.nn.grad.conv2d_weight(x, (K, C, R, S), dy, stride=stride, padding=pad)   # [K,C,R,S]
    A.view(op.dw).add_(dw.permute(0, 2, 3, 1))
    if op.dbias is not None:
        A.view(op.dbias).add_(dy.sum((0, 2, 3)))


def run_stem_fwd(A, op):
    # bf16 build: the MFMA stem multiplies bf16-rounded image and weights (fp32 accumulate); fp32 build: exact fp32
    img = _rnd(A, A.view(op.image))
    wt = _rnd(A, A.view(op.w)).permute(0, 3, 1, 2)         # [K,7,7,3] -> [K,3,7,7]
    y = F.conv2d(img, wt, A.view(op.bias), stride=2, padding=3).permute(0, 2, 3, 1)
    y = _rnd(A, y)
    if op.out_stats is not None:
        _stats_add(A, op.out_stats, y)
    _store(A, op.y, y)


def run_stem_wgrad(A, op):
    img = _rnd(A, A.view(op.image))
    dy = _act(A, op.dy).permute(0, 3, 1, 2)
    K = dy.shape[1]
    dw = torch.nn.grad.conv2d_weight(img, (K, 3, 7, 7), dy, stride=2, padding=3)
    A.view(op.dw).add_(dw.permute(0, 2, 3, 1))
    A.view(op.dbias).add_(dy.sum((0, 2, 3)))


def run_ew(A, op):
    name = op.op
    if name == 'bnrelu_fwd':
        y = _prologue(A, _act(A, op.x), op.bn)
        if op.out_stats is not None:
            _stats_add(A, op.out_stats, y)
        _store(A, op.y, y)
    elif name == 'bnrelu_bwd_r':
        x, dy = _act(A, op.x), _act(A, op.dy)
        scale, shift, mean, invstd = _bn_coef(A, op.bn)
        z = torch.addcmul(shift, x, scale)
        if getattr(op, 'x2', None) is not None:   # mask source = the forward OUTPUT of an affsum op (relu of a sum), not bn(x)
            dz = torch.where(_act(A, op.x2) > 0, dy, torch.zeros_like(dy))
        else:
            dz = torch.where(z > 0, dy, torch.zeros_like(dy)) if op.bn.relu else dy
        dzr = _rnd(A, dz).double()
        st = A.view(op.bstats)[0]
        st[0] += dzr.sum((0, 1, 2))
        st[1] += (dzr * ((x - mean) * invstd).double()).sum((0, 1, 2))
        if op.y is not None:                      # y None: statistics only (BN terms of an affsum op)
            _store(A, op.y, dz)
    elif name == 'relu_mask':                     # g = dy where the forward output y (op.x) is positive
        y, dy = _act(A, op.x), _act(A, op.dy)
        _store(A, op.y, torch.where(y > 0, dy, torch.zeros_like(dy)))
    elif name == 'dilate2':                       # zero-dilation: out[n, 2p, 2q] = x[n, p, q], zero elsewhere
        x = _act(A, op.x)
        n, h, w, c = op.dims
        out = torch.zeros(n, h, w, c)
        out[:, 0::2, 0::2] = x
        _store(A, op.y, out)
    elif name == 'bn_bwd_apply':
        x, dz = _act(A, op.x), _act(A, op.dy)
        _, _, mean, invstd = _bn_coef(A, op.bn)
        st = A.view(op.bstats).sum(0)
        m1, m2 = (st[0] / op.bn.count).float(), (st[1] / op.bn.count).float()
        g = A.view(op.bn.gamma)
        out = (g * invstd) * (dz - m1 - ((x - mean) * invstd) * m2)
        if op.add is not None:
            out = out + _act(A, op.add)
        if op.dgamma is not None:
            A.view(op.dgamma).copy_(st[1].float())
        if op.dbeta is not None:
            A.view(op.dbeta).copy_(st[0].float())
        _store(A, op.y, out)
    elif name == 'maxpool_fwd':
        y = F.max_pool2d(_act(A, op.x).permute(0, 3, 1, 2), 2, stride=2).permute(0, 2, 3, 1)
        if op.out_stats is not None:
            _stats_add(A, op.out_stats, y)
        _store(A, op.y, y)
    elif name == 'maxpool_bwd':
        x = _act(A, op.x).permute(0, 3, 1, 2)
        _, idx = F.max_pool2d(x, 2, stride=2, return_indices=True)
        dy = _act(A, op.dy).permute(0, 3, 1, 2)
        dx = F.max_unpool2d(dy, idx, 2, stride=2, output_size=x.shape[2:]).permute(0, 2, 3, 1)
        if op.add is not None:
            dx = dx + _act(A, op.add)
        _store(A, op.y, dx)
    elif name == 'upadd_fwd':
        up = F.interpolate(_act(A, op.x2).permute(0, 3, 1, 2), scale_factor=2, mode='nearest').permute(0, 2, 3, 1)
        y = _rnd(A, _act(A, op.x) + up)
        if op.out_stats is not None:
            _stats_add(A, op.out_stats, y)
        _store(A, op.y, y)
    elif name == 'sumpool':
        y = 4.0 * F.avg_pool2d(_act(A, op.x).permute(0, 3, 1, 2), 2, stride=2).permute(0, 2, 3, 1)
        if op.add is not None:
            y = y + _act(A, op.add)
        _store(A, op.y, y)
    elif name == 'add':
        _store(A, op.y, _act(A, op.x) + _act(A, op.x2))
    else:
        raise AssertionError(name)


def run_affsum(A, op):
    """include/fpd_amd.h fpd_affsum_t: y = relu?(sum_j bn_j(nearest_up_fj(x_j))) -- HRNet block tails
    (/root/reference/lib/models/pose_hrnet.py:52-57,93-98), fuse layers (:252-265) and transition outputs.  Each
    normalised term is evaluated as x*scale + shift in fp32, the sum is rounded to the storage precision once."""
    acc = None
    for t, bn, f in op.terms:
        v = _act(A, t)
        if bn is not None:
            scale, shift, _, _ = _bn_coef(A, bn)
            v = torch.addcmul(shift, v, scale)
            if bn.relu:
                v = v.clamp_min(0)
        if f > 1:
            v = v.repeat_interleave(f, dim=1).repeat_interleave(f, dim=2)
        acc = v if acc is None else acc + v
    if op.relu:
        acc = acc.clamp_min(0)
    y = _rnd(A, acc)
    if op.out_stats is not None:
        _stats_add(A, op.out_stats, y)
    _store(A, op.y, y)


def run_nchw2nhwc(A, op):
    n, c, h, w = op.dims
    _store(A, op.y, A.view(op.image).view(n, c, h, w).permute(0, 2, 3, 1))


def run_bnupd(A, op, momentum=0.1):
    for bn in op.bns:
        st = A.view(bn.stats).sum(0)
        mean = st[0] / bn.count
        var = (st[1] / bn.count - mean * mean).clamp_min(0)
        unb = var * bn.count / (bn.count - 1) if bn.count > 1 else var
        rm, rv = A.view(bn.rmean), A.view(bn.rvar)
        rm.copy_(((1 - momentum) * rm.double() + momentum * mean).float())
        rv.copy_(((1 - momentum) * rv.double() + momentum * unb).float())
        A.view(bn.nbt).add_(1)


def run_loss(A, op):
    """include/fpd_amd.h fpd_loss_t: losses += {pose, kd}; dout_s = gs*w^2[(1-a)(p-g)+a(p-t)]/cnt."""
    tgt = A.t['target'].view(op.target_shape)                   # [B,J,H,W] fp32
    wgt = A.t['weight'].view(op.B, op.J)
    t = _act_any(A, op.teacher)
    cnt = float(tgt.numel())
    g = tgt.permute(0, 2, 3, 1)
    w2 = (wgt * wgt)[:, None, None, :]
    losses = A.t['losses']
    for s, o in enumerate(op.outs):
        p = _act(A, o)
        dg, dt = p - g, p - t
        losses[0] += 0.5 * (w2 * dg * dg).double().sum() / cnt
        losses[1] += 0.5 * (w2 * dt * dt).double().sum() / cnt
        if op.douts is not None:
            _store(A, op.douts[s], op.grad_scale * w2 * ((1 - op.alpha) * dg + op.alpha * dt) / cnt)


def _act_any(A, a):
    return a if isinstance(a, torch.Tensor) else _act(A, a)


def run_wprep(A, op):
    for e in op.entries:
        w = A.view(e['w'])
        if e.get('w_fwd') is not None:
            A.view(e['w_fwd']).copy_(w.to(A.act_dtype))
        if e.get('w_bwd') is not None:
            A.view(e['w_bwd']).copy_(w.flip(1, 2).permute(3, 1, 2, 0).to(A.act_dtype))


RUN = {'conv': run_conv, 'wgrad': run_wgrad, 'stem_fwd': run_stem_fwd, 'stem_wgrad': run_stem_wgrad, 'ew': run_ew,
       'bnupd': run_bnupd, 'loss': run_loss, 'wprep': run_wprep, 'bneck': run_bneck,
       'bneck_fold': lambda A, op: None,
       'conv2': lambda A, op: (run_conv(A, op.a), run_conv(A, op.b)),
       'bneck2': lambda A, op: (run_bneck(A, op.a), run_bneck(A, op.b)),
       'ew2': lambda A, op: (run_ew(A, op.a), run_ew(A, op.b)),
       'head': run_head, 'head_fold': lambda A, op: None,
       # the interpreter's wgrad accumulates straight into dw (no slabs): the slab reduction and the bucket marker of
       # the device plan have no effect on the specification
       'wreduce': lambda A, op: None, 'grad_ready': lambda A, op: None,
       'affsum': run_affsum, 'nchw2nhwc': run_nchw2nhwc}       # one launch, two independent convolutions       # device-side table preparation: no effect on the specification


def run(A, ops):
    with torch.no_grad():
        for op in ops:
            if op.kind == 'memset':
                A.t[op.arena][op.off:op.off + op.n].zero_()
            elif op.kind == 'seed':
                pass
            else:
                RUN[op.kind](A, op)
