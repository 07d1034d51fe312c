"""`models.hourglass` with the reference's module API, executed by the MI355X HIP path.

API mirrored from /root/reference/lib/models/hourglass.py:
  * get_pose_net(cfg, is_train, **kw) -> nn.Module                      (:195-197)
  * forward(x[N,3,H,W] fp32) -> list of NUM_STACKS tensors [N,J,H/4,W/4] (:170-192)
  * state_dict keys / shapes identical (conv OIHW, BN weight/bias/running_*/num_batches_tracked), so
    reference checkpoints load with strict=True (tools/fpd_train.py:139-141, lib/utils/utils.py:250-255)
  * .train()/.eval() switch BatchNorm between batch and running statistics (lib/core/function.py:110-111)
  * the module tree has the reference's classes, names and registration order (HourglassNet / Hourglass /
    Bottleneck containers; Conv2d, BatchNorm2d, ReLU, MaxPool2d, Upsample leaves), so named_modules(),
    isinstance(m, nn.Conv2d) loops and per-layer forward hooks see what they see on the reference
What differs by design: parameters are views into one flat HBM arena (conv weights channels-last) and the
forward/backward run as ONE recorded plan of hand-written gfx950 kernels -- the leaf modules carry parameters
and metadata but are never called, and calling one (or giving the model a CPU tensor) is an error: there is no
eager/CPU fallback.  What the reference does through CPU dry-runs (tools/fpd_train.py:162-167) is served by
`HourglassNet.shape_forward()`: it walks the modules in the reference's execution order and fires their forward
hooks with shape-only (meta-device) tensors, which is all `utils.get_model_summary` needs; tensorboard's
`add_graph` (a JIT trace of torch ops) has no counterpart.
"""
import math

import torch
import torch.nn as nn

from ... import graph as G
from ... import runtime as R
from ._flat import BatchNorm2d, Conv2d, FlatArenaNet, MaxPool2d, ReLU, Upsample, _no_eager  # noqa: F401

BN_MOMENTUM = 0.1


def hourglass_keys(num_feats, num_stacks, num_joints, num_blocks=1, depth=4):
    """(key, shape) of every state_dict entry in the reference's registration order
    (hourglass.py:100-168: stem, layer1-3, then ModuleLists hg, res, fc, score, fc_, score_)."""
    def conv(dst, name, co, ci, k):
        dst += [(name + '.weight', (co, ci, k, k)), (name + '.bias', (co,))]

    def bn(dst, name, c):
        dst += [(name + '.weight', (c,)), (name + '.bias', (c,)), (name + '.running_mean', (c,)),
                (name + '.running_var', (c,)), (name + '.num_batches_tracked', ())]

    def block(dst, p, cin, planes):
        bn(dst, p + 'bn1', cin); conv(dst, p + 'conv1', planes, cin, 1)
        bn(dst, p + 'bn2', planes); conv(dst, p + 'conv2', planes, planes, 3)
        bn(dst, p + 'bn3', planes); conv(dst, p + 'conv3', 2 * planes, planes, 1)
        if cin != 2 * planes:
            conv(dst, p + 'downsample.0', 2 * planes, cin, 1)

    inpl, nf = num_feats // 4, num_feats // 2
    keys = []
    conv(keys, 'conv1', inpl, 3, 7)
    bn(keys, 'bn1', inpl)
    block(keys, 'layer1.0.', inpl, inpl)
    block(keys, 'layer2.0.', 2 * inpl, 2 * inpl)
    block(keys, 'layer3.0.', 4 * inpl, nf)
    ch = 2 * nf
    groups = {k: [] for k in ('hg', 'res', 'fc', 'score', 'fc_', 'score_')}
    for i in range(num_stacks):
        for d in range(depth):
            for j in range(4 if d == 0 else 3):
                for b in range(num_blocks):
                    block(groups['hg'], 'hg.%d.hg.%d.%d.%d.' % (i, d, j, b), ch, nf)
        for b in range(num_blocks):
            block(groups['res'], 'res.%d.%d.' % (i, b), ch, nf)
        conv(groups['fc'], 'fc.%d.0' % i, ch, ch, 1)
        bn(groups['fc'], 'fc.%d.1' % i, ch)
        conv(groups['score'], 'score.%d' % i, num_joints, ch, 1)
        if i < num_stacks - 1:
            conv(groups['fc_'], 'fc_.%d' % i, ch, ch, 1)
            conv(groups['score_'], 'score_.%d' % i, ch, num_joints, 1)
    for k in ('hg', 'res', 'fc', 'score', 'fc_', 'score_'):
        keys += groups[k]
    return keys


class Bottleneck(nn.Module):
    """hourglass.py:11-52 (pre-activation, expansion 2): container of bn1, conv1, bn2, conv2, bn3, conv3, relu, downsample."""
    expansion = 2

    def __init__(self, inplanes, planes):
        super().__init__()
        self.bn1 = BatchNorm2d(inplanes)
        self.conv1 = Conv2d(inplanes, planes, 1)
        self.bn2 = BatchNorm2d(planes)
        self.conv2 = Conv2d(planes, planes, 3, padding=1)
        self.bn3 = BatchNorm2d(planes)
        self.conv3 = Conv2d(planes, 2 * planes, 1)
        self.relu = ReLU(inplace=True)
        self.downsample = nn.Sequential(Conv2d(inplanes, 2 * planes, 1)) if inplanes != 2 * planes else None
        self.stride = 1
    forward = _no_eager


def _residual(inplanes, planes, num_blocks):
    return nn.Sequential(*[Bottleneck(inplanes if b == 0 else 2 * planes, planes) for b in range(num_blocks)])


class Hourglass(nn.Module):
    """hourglass.py:55-95: hg[d][j] residual sequences (4 at the innermost level, 3 elsewhere) + the nearest x2 upsample."""

    def __init__(self, num_blocks, planes, depth):
        super().__init__()
        self.depth = depth
        self.upsample = Upsample(scale_factor=2)
        self.hg = nn.ModuleList([nn.ModuleList([_residual(2 * planes, planes, num_blocks) for _ in range(4 if d == 0 else 3)])
                                 for d in range(depth)])
    forward = _no_eager


class HourglassNet(FlatArenaNet):
    """Stacked hourglass (Newell et al.) student/teacher of the FPD path, HIP-backed."""

    def __init__(self, cfg, **kwargs):
        super().__init__()
        extra = cfg.MODEL.EXTRA
        self.cfg_hg = {'F': int(extra.NUM_FEATURES), 'S': int(extra.NUM_STACKS), 'J': int(cfg.MODEL.NUM_JOINTS),
                       'num_blocks': int(extra.NUM_BLOCKS)}
        self.num_stacks = self.cfg_hg['S']
        self.fpd_dtype = self._dtype_from(cfg, kwargs)
        keys = hourglass_keys(self.cfg_hg['F'], self.cfg_hg['S'], self.cfg_hg['J'], self.cfg_hg['num_blocks'])
        self._init_flat(G.ParamTable(keys, bucket_of=G.hourglass_bucket_of(self.cfg_hg['S'])))
        self._build_tree()
        self._bind_tree()
        self.reset_parameters()

    def _build_tree(self):
        """The reference's module tree (hourglass.py:100-168) in its registration order, then every parameter / buffer
        re-pointed at its view of the flat arenas."""
        F_, S, J, nb = self.cfg_hg['F'], self.cfg_hg['S'], self.cfg_hg['J'], self.cfg_hg['num_blocks']
        inpl, nf = F_ // 4, F_ // 2
        self.inplanes, self.num_feats = 2 * nf, nf
        self.conv1 = Conv2d(3, inpl, 7, stride=2, padding=3)
        self.bn1 = BatchNorm2d(inpl)
        self.relu = ReLU(inplace=True)
        self.layer1 = _residual(inpl, inpl, 1)
        self.layer2 = _residual(2 * inpl, 2 * inpl, 1)     # hourglass.py:121: planes = the running self.inplanes
        self.layer3 = _residual(4 * inpl, nf, 1)
        self.maxpool = MaxPool2d(2, stride=2)
        ch = 2 * nf
        self.hg = nn.ModuleList([Hourglass(nb, nf, 4) for _ in range(S)])
        self.res = nn.ModuleList([_residual(ch, nf, nb) for _ in range(S)])
        self.fc = nn.ModuleList([nn.Sequential(Conv2d(ch, ch, 1), BatchNorm2d(ch), self.relu) for _ in range(S)])
        self.score = nn.ModuleList([Conv2d(ch, J, 1) for _ in range(S)])
        self.fc_ = nn.ModuleList([Conv2d(ch, ch, 1) for _ in range(S - 1)])
        self.score_ = nn.ModuleList([Conv2d(J, ch, 1) for _ in range(S - 1)])

    def reset_parameters(self):
        """torch defaults, as the reference relies on (no custom init, hourglass.py:195-197): conv weight and
        bias U(+-1/sqrt(fan_in)) (= kaiming_uniform(a=sqrt 5)), BN weight 1 / bias 0 / mean 0 / var 1."""
        with torch.no_grad():
            for key, shp in self.table.keys:
                v = self._view(key)
                if key.endswith('num_batches_tracked'):
                    v.zero_()
                elif key.endswith('running_mean'):
                    v.zero_()
                elif key.endswith('running_var'):
                    v.fill_(1.0)
                elif len(shp) == 4:
                    bound = 1.0 / math.sqrt(shp[1] * shp[2] * shp[3])
                    v.uniform_(-bound, bound)
                else:
                    base = key.rsplit('.', 1)[0]
                    if (base + '.running_mean') in self.table.entries:
                        v.fill_(1.0) if key.endswith('.weight') else v.zero_()
                    else:
                        ws = self.table.logical[base + '.weight']
                        bound = 1.0 / math.sqrt(ws[1] * ws[2] * ws[3])
                        v.uniform_(-bound, bound)

    # ---- shape-only walk in the reference's execution order (hourglass.py:32-52,80-92,170-192) ----
    def shape_forward(self, input_shape):
        """Fire every module's forward hooks, in the order the reference's forward would, with shape-only (meta-device)
        tensors; returns the list of output shapes.  This is what `utils.get_model_summary` and other hook-based
        inspection tools get instead of the reference's CPU dry-run (the compute itself is one fused plan)."""
        def fire(m, ishape, oshape):
            if m._forward_hooks:
                ti = torch.empty(tuple(ishape), device='meta')
                to = [torch.empty(tuple(o), device='meta') for o in oshape] if isinstance(oshape, list) else \
                    torch.empty(tuple(oshape), device='meta')
                for h in list(m._forward_hooks.values()):
                    h(m, (ti,), to)
            return oshape

        def conv(m, s):
            n, c, h, w = s
            assert c == m.in_channels, (c, m.in_channels)
            k, st, p = m.kernel_size[0], m.stride[0], m.padding[0]
            return fire(m, s, (n, m.out_channels, (h + 2 * p - k) // st + 1, (w + 2 * p - k) // st + 1))

        def same(m, s):
            return fire(m, s, s)

        def block(b, s):
            o = same(b.relu, same(b.bn1, s))
            o = conv(b.conv1, o)
            o = conv(b.conv2, same(b.relu, same(b.bn2, o)))
            o = conv(b.conv3, same(b.relu, same(b.bn3, o)))
            if b.downsample is not None:
                fire(b.downsample, s, conv(b.downsample[0], s))
            return fire(b, s, o)

        def seq(q, s):
            o = s
            for b in q:
                o = block(b, o)
            return fire(q, s, o)

        def hour(hgm, n, s):
            up1 = seq(hgm.hg[n - 1][0], s)
            low = (s[0], s[1], s[2] // 2, s[3] // 2)                 # F.max_pool2d: functional, no module, no hook
            low = seq(hgm.hg[n - 1][1], low)
            low = hour(hgm, n - 1, low) if n > 1 else seq(hgm.hg[n - 1][3], low)
            low = seq(hgm.hg[n - 1][2], low)
            fire(hgm.upsample, low, up1)
            return up1

        x = same(self.relu, same(self.bn1, conv(self.conv1, tuple(input_shape))))
        x = seq(self.layer1, x)
        x = fire(self.maxpool, x, (x[0], x[1], x[2] // 2, x[3] // 2))
        x = seq(self.layer3, seq(self.layer2, x))
        outs = []
        for i in range(self.num_stacks):
            y = fire(self.hg[i], x, hour(self.hg[i], self.hg[i].depth, x))
            y = seq(self.res[i], y)
            f = self.fc[i]
            y = fire(f, y, same(f[2], same(f[1], conv(f[0], y))))
            sc = conv(self.score[i], y)
            outs.append(sc)
            if i < self.num_stacks - 1:
                conv(self.fc_[i], y)
                conv(self.score_[i], sc)
        return outs


def get_pose_net(cfg, is_train, **kwargs):
    """Same factory signature as the reference (is_train is ignored there too, hourglass.py:195-197)."""
    return HourglassNet(cfg, **kwargs)
