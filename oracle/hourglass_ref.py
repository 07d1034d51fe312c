"""CPU restatement of the reference stacked-hourglass forward (test infrastructure).

Functional form over a flat ``state_dict`` (the reference's key names), NCHW fp32,
plain ``torch.nn.functional`` ops.  Follows, without copying,
  /root/reference/lib/models/hourglass.py
    :32-52   Bottleneck.forward      (pre-activation, expansion 2, bias=True)
    :80-92   Hourglass._hour_glass_forward (depth-4 recursion, 2x2 max-pool, nearest x2 up-add)
    :170-192 HourglassNet.forward    (stem, 3 residual layers, per-stack hg/res/fc/score, inter-stack sum)
BatchNorm semantics are torch defaults with momentum 0.1 (:10,:18): eps 1e-5,
biased variance for normalisation, unbiased for the running estimate.

Autograd through these functions gives the reference gradients: pass a
state_dict whose parameter tensors have ``requires_grad=True``.
"""
import torch
import torch.nn.functional as F

BN_MOMENTUM = 0.1  # hourglass.py:10
BN_EPS = 1e-5      # torch default used by every nn.BatchNorm2d in the file


def _bn(sd, name, x, train):
    """nn.BatchNorm2d(momentum=0.1) -- hourglass.py:18,21,25,118,162."""
    rm, rv = sd[name + '.running_mean'], sd[name + '.running_var']
    y = F.batch_norm(x, rm, rv, sd[name + '.weight'], sd[name + '.bias'],
                     training=train, momentum=BN_MOMENTUM, eps=BN_EPS)
    if train and (name + '.num_batches_tracked') in sd:
        sd[name + '.num_batches_tracked'] += 1
    return y


def _conv(sd, name, x, stride=1, padding=0):
    return F.conv2d(x, sd[name + '.weight'], sd[name + '.bias'], stride=stride, padding=padding)


def bottleneck(sd, p, x, train):
    """hourglass.py:32-52.  `p` is the key prefix, e.g. 'layer1.0.'."""
    t = _conv(sd, p + 'conv1', F.relu(_bn(sd, p + 'bn1', x, train)))
    t = _conv(sd, p + 'conv2', F.relu(_bn(sd, p + 'bn2', t, train)), padding=1)
    t = _conv(sd, p + 'conv3', F.relu(_bn(sd, p + 'bn3', t, train)))
    skip = x
    if (p + 'downsample.0.weight') in sd:      # 1x1 conv on the skip, hourglass.py:47-48,142-146
        skip = _conv(sd, p + 'downsample.0', x)
    return t + skip


def _residual_seq(sd, p, x, train, num_blocks):
    for b in range(num_blocks):
        x = bottleneck(sd, '%s%d.' % (p, b), x, train)
    return x


def hour_glass(sd, p, n, x, train, num_blocks):
    """hourglass.py:80-92.  `p` = 'hg.<stack>.hg.'; level n uses hg[n-1]."""
    q = '%s%d.' % (p, n - 1)
    up1 = _residual_seq(sd, q + '0.', x, train, num_blocks)
    low = F.max_pool2d(x, 2, stride=2)
    low = _residual_seq(sd, q + '1.', low, train, num_blocks)
    if n > 1:
        low = hour_glass(sd, p, n - 1, low, train, num_blocks)
    else:
        low = _residual_seq(sd, q + '3.', low, train, num_blocks)
    low = _residual_seq(sd, q + '2.', low, train, num_blocks)
    return up1 + F.interpolate(low, scale_factor=2, mode='nearest')


def hourglass_forward(sd, x, num_stacks, train, num_blocks=1, depth=4):
    """hourglass.py:170-192.  Returns the list of per-stack heat-maps [N,J,H/4,W/4]."""
    x = _conv(sd, 'conv1', x, stride=2, padding=3)
    x = F.relu(_bn(sd, 'bn1', x, train))
    x = _residual_seq(sd, 'layer1.', x, train, 1)
    x = F.max_pool2d(x, 2, stride=2)
    x = _residual_seq(sd, 'layer2.', x, train, 1)
    x = _residual_seq(sd, 'layer3.', x, train, 1)
    out = []
    for i in range(num_stacks):
        y = hour_glass(sd, 'hg.%d.hg.' % i, depth, x, train, num_blocks)
        y = _residual_seq(sd, 'res.%d.' % i, y, train, num_blocks)
        y = F.relu(_bn(sd, 'fc.%d.1' % i, _conv(sd, 'fc.%d.0' % i, y), train))
        score = _conv(sd, 'score.%d' % i, y)
        out.append(score)
        if i < num_stacks - 1:
            x = x + _conv(sd, 'fc_.%d' % i, y) + _conv(sd, 'score_.%d' % i, score)
    return out


def hourglass_keys(num_feats, num_stacks, num_joints, num_blocks=1, depth=4):
    """(key, shape) list of the reference HourglassNet state_dict, in module
    registration order (hourglass.py:100-168: stem, layer1-3, then the ModuleLists
    hg, res, fc, score, fc_, score_).  Lets tests build a synthetic checkpoint
    without importing the reference."""

    def conv(dst, name, co, ci, k):
        dst.append((name + '.weight', (co, ci, k, k)))
        dst.append((name + '.bias', (co,)))

    def bn(dst, name, c):
        dst.append((name + '.weight', (c,)))
        dst.append((name + '.bias', (c,)))
        dst.append((name + '.running_mean', (c,)))
        dst.append((name + '.running_var', (c,)))
        dst.append((name + '.num_batches_tracked', ()))

    def block(dst, p, cin, planes, down):
        bn(dst, p + 'bn1', cin)
        conv(dst, p + 'conv1', planes, cin, 1)
        bn(dst, p + 'bn2', planes)
        conv(dst, p + 'conv2', planes, planes, 3)
        bn(dst, p + 'bn3', planes)
        conv(dst, p + 'conv3', planes * 2, planes, 1)
        if down:
            conv(dst, p + 'downsample.0', planes * 2, cin, 1)

    inpl, nf = num_feats // 4, num_feats // 2
    keys = []
    conv(keys, 'conv1', inpl, 3, 7)
    bn(keys, 'bn1', inpl)
    block(keys, 'layer1.0.', inpl, inpl, True)        # F/4 -> F/2, 1x1 downsample on the skip
    block(keys, 'layer2.0.', 2 * inpl, 2 * inpl, True)  # F/2 -> F (planes = the UPDATED self.inplanes, :122)
    block(keys, 'layer3.0.', 4 * inpl, nf, False)     # F -> F
    ch = nf * 2
    names = ('hg', 'res', 'fc', 'score', 'fc_', 'score_')
    per = {k: [] for k in names}
    for i in range(num_stacks):
        for d in range(depth):
            for j in range(4 if d == 0 else 3):
                for b in range(num_blocks):
                    block(per['hg'], 'hg.%d.hg.%d.%d.%d.' % (i, d, j, b), ch, nf, False)
        for b in range(num_blocks):
            block(per['res'], 'res.%d.%d.' % (i, b), ch, nf, False)
        conv(per['fc'], 'fc.%d.0' % i, ch, ch, 1)
        bn(per['fc'], 'fc.%d.1' % i, ch)
        conv(per['score'], 'score.%d' % i, num_joints, ch, 1)
        if i < num_stacks - 1:
            conv(per['fc_'], 'fc_.%d' % i, ch, ch, 1)
            conv(per['score_'], 'score_.%d' % i, ch, num_joints, 1)
    for k in names:
        keys.extend(per[k])
    return keys
