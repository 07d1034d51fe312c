"""CPU restatement of the reference HRNet pose network forward (test infrastructure; groundwork for the HRNet rows of
SURVEY.md section 8(a) -- no product path exists for this model yet).

Functional form over a flat ``state_dict`` with the reference's key names, NCHW fp32, plain ``torch.nn.functional`` ops.
Follows, without copying,
  /root/reference/lib/models/pose_hrnet.py
    :28-57    BasicBlock.forward            post-activation, two 3x3 convs, bias-free, expansion 1
    :60-98    Bottleneck.forward            1x1 -> 3x3 -> 1x1, expansion 4 (stem layer1 only)
    :139-175  _make_one_branch              downsample = 1x1 conv + BN when channels change
    :187-242  _make_fuse_layers             up: 1x1 conv + BN + nearest x2^(j-i); down: (i-j) x [3x3/2 conv + BN (+ReLU except last)]
    :247-265  HighResolutionModule.forward  branches, then relu(sum_j fuse_ij(x_j))
    :333-372  _make_transition_layer        3x3 conv+BN+ReLU on a channel change; new branch = 3x3/2 conv+BN+ReLU from the last one
    :425-460  PoseHighResolutionNet.forward stem (two 3x3/2 convs), layer1, three stages, final 1x1 conv (the only conv with bias)
BatchNorm: momentum 0.1 for the blocks and the stem (:19 BN_MOMENTUM), torch default 0.1 for the fuse / transition BNs
(which do not pass a momentum) -- the same value; eps 1e-5 everywhere.

`cfg` is the plain dict of MODEL.EXTRA: {'STAGE2': {...}, 'STAGE3': {...}, 'STAGE4': {...}, 'FINAL_CONV_KERNEL': 1} with the
reference's field names (NUM_MODULES, NUM_BRANCHES, BLOCK, NUM_BLOCKS, NUM_CHANNELS, FUSE_METHOD).
Autograd through these functions gives the reference gradients (pass parameters with requires_grad=True).
"""
import torch
import torch.nn.functional as F

BN_MOMENTUM = 0.1
BN_EPS = 1e-5
EXPANSION = {'BASIC': 1, 'BOTTLENECK': 4}


def _bn(sd, name, x, train):
    y = F.batch_norm(x, sd[name + '.running_mean'], sd[name + '.running_var'], sd[name + '.weight'], sd[name + '.bias'],
                     training=train, momentum=BN_MOMENTUM, eps=BN_EPS)
    if train and (name + '.num_batches_tracked') in sd:
        sd[name + '.num_batches_tracked'] += 1
    return y


def _conv(sd, name, x, stride=1, padding=0):
    return F.conv2d(x, sd[name + '.weight'], sd.get(name + '.bias'), stride=stride, padding=padding)


def basic_block(sd, p, x, train, stride=1):
    """pose_hrnet.py:41-57."""
    out = F.relu(_bn(sd, p + 'bn1', _conv(sd, p + 'conv1', x, stride=stride, padding=1), train))
    out = _bn(sd, p + 'bn2', _conv(sd, p + 'conv2', out, padding=1), train)
    res = x
    if (p + 'downsample.0.weight') in sd:
        res = _bn(sd, p + 'downsample.1', _conv(sd, p + 'downsample.0', x, stride=stride), train)
    return F.relu(out + res)


def bottleneck(sd, p, x, train, stride=1):
    """pose_hrnet.py:78-98."""
    out = F.relu(_bn(sd, p + 'bn1', _conv(sd, p + 'conv1', x), train))
    out = F.relu(_bn(sd, p + 'bn2', _conv(sd, p + 'conv2', out, stride=stride, padding=1), train))
    out = _bn(sd, p + 'bn3', _conv(sd, p + 'conv3', out), train)
    res = x
    if (p + 'downsample.0.weight') in sd:
        res = _bn(sd, p + 'downsample.1', _conv(sd, p + 'downsample.0', x, stride=stride), train)
    return F.relu(out + res)


_BLOCK = {'BASIC': basic_block, 'BOTTLENECK': bottleneck}


def hr_module(sd, p, xs, train, block, num_blocks, multi_scale_output=True):
    """pose_hrnet.py:247-265.  xs: list of branch inputs; p = 'stageK.M.'."""
    nb = len(xs)
    ys = []
    for i in range(nb):
        y = xs[i]
        for b in range(num_blocks[i]):
            y = _BLOCK[block](sd, '%sbranches.%d.%d.' % (p, i, b), y, train)
        ys.append(y)
    if nb == 1:
        return ys
    outs = []
    for i in range(nb if multi_scale_output else 1):
        acc = None
        for j in range(nb):
            q = '%sfuse_layers.%d.%d.' % (p, i, j)
            if j == i:
                t = ys[j]
            elif j > i:       # low resolution -> high: 1x1 conv, BN, nearest up-sampling by 2^(j-i)
                t = _bn(sd, q + '1', _conv(sd, q + '0', ys[j]), train)
                t = F.interpolate(t, scale_factor=2 ** (j - i), mode='nearest')
            else:             # high -> low: (i-j) stride-2 3x3 convs, ReLU between them but not after the last
                t = ys[j]
                for k in range(i - j):
                    t = _bn(sd, '%s%d.1' % (q, k), _conv(sd, '%s%d.0' % (q, k), t, stride=2, padding=1), train)
                    if k != i - j - 1:
                        t = F.relu(t)
            acc = t if acc is None else acc + t
        outs.append(F.relu(acc))
    return outs


def _transition(sd, p, ys, pre_channels, cur_channels, train):
    """pose_hrnet.py:333-372 applied as in forward() (:436-457): a branch without a transition layer passes through; every
    transition layer -- the 3x3 conv+BN+ReLU of a width change as well as the stride-2 chain of a new branch -- is fed the
    LAST branch of the previous stage (forward() passes y_list[-1] whenever transition[i] is not None)."""
    xs = []
    for i, c in enumerate(cur_channels):
        q = '%s%d.' % (p, i)
        if i < len(pre_channels):
            if c != pre_channels[i]:
                xs.append(F.relu(_bn(sd, q + '1', _conv(sd, q + '0', ys[-1], padding=1), train)))
            else:
                xs.append(ys[i])
        else:
            t = ys[-1]
            for j in range(i + 1 - len(pre_channels)):
                t = F.relu(_bn(sd, '%s%d.1' % (q, j), _conv(sd, '%s%d.0' % (q, j), t, stride=2, padding=1), train))
            xs.append(t)
    return xs


def stage_channels(cfg):
    """[[channels per branch] for STAGE2..4], multiplied by the block expansion (pose_hrnet.py:291-296)."""
    return [[c * EXPANSION[cfg[s]['BLOCK']] for c in cfg[s]['NUM_CHANNELS']] for s in ('STAGE2', 'STAGE3', 'STAGE4')]


def hrnet_forward(sd, cfg, x, train):
    """pose_hrnet.py:425-460 -> heat-maps [N, J, H/4, W/4]."""
    x = F.relu(_bn(sd, 'bn1', _conv(sd, 'conv1', x, stride=2, padding=1), train))
    x = F.relu(_bn(sd, 'bn2', _conv(sd, 'conv2', x, stride=2, padding=1), train))
    for b in range(4):                                   # layer1 = 4 Bottlenecks, 64 -> 256 (:288)
        x = bottleneck(sd, 'layer1.%d.' % b, x, train)
    chans = stage_channels(cfg)
    ys, pre = [x], [256]
    for si, sname in enumerate(('STAGE2', 'STAGE3', 'STAGE4')):
        sc = cfg[sname]
        xs = _transition(sd, 'transition%d.' % (si + 1), ys, pre, chans[si], train)
        for m in range(sc['NUM_MODULES']):
            last = (sname == 'STAGE4') and m == sc['NUM_MODULES'] - 1          # multi_scale_output=False (:320-321,400-403)
            xs = hr_module(sd, 'stage%d.%d.' % (si + 2, m), xs, train, sc['BLOCK'], sc['NUM_BLOCKS'], multi_scale_output=not last)
        ys, pre = xs, chans[si]
    k = cfg.get('FINAL_CONV_KERNEL', 1)
    return _conv(sd, 'final_layer', ys[0], padding=1 if k == 3 else 0)


def hrnet_keys(cfg, num_joints):
    """[(key, shape)] in the reference's state_dict order (module registration order of pose_hrnet.py:276-331)."""
    keys = []

    def conv(name, cout, cin, k, bias=False):
        keys.append((name + '.weight', (cout, cin, k, k)))
        if bias:
            keys.append((name + '.bias', (cout,)))

    def bn(name, c):
        keys.extend([(name + '.weight', (c,)), (name + '.bias', (c,)), (name + '.running_mean', (c,)),
                     (name + '.running_var', (c,)), (name + '.num_batches_tracked', ())])

    def block(kind, p, cin, planes, downsample):
        if kind == 'BASIC':
            conv(p + 'conv1', planes, cin, 3); bn(p + 'bn1', planes)
            conv(p + 'conv2', planes, planes, 3); bn(p + 'bn2', planes)
        else:
            conv(p + 'conv1', planes, cin, 1); bn(p + 'bn1', planes)
            conv(p + 'conv2', planes, planes, 3); bn(p + 'bn2', planes)
            conv(p + 'conv3', planes * 4, planes, 1); bn(p + 'bn3', planes * 4)
        if downsample:
            conv(p + 'downsample.0', planes * EXPANSION[kind], cin, 1); bn(p + 'downsample.1', planes * EXPANSION[kind])

    conv('conv1', 64, 3, 3); bn('bn1', 64)
    conv('conv2', 64, 64, 3); bn('bn2', 64)
    for b in range(4):
        block('BOTTLENECK', 'layer1.%d.' % b, 64 if b == 0 else 256, 64, downsample=(b == 0))
    chans = stage_channels(cfg)
    pre = [256]
    for si, sname in enumerate(('STAGE2', 'STAGE3', 'STAGE4')):
        sc, cur = cfg[sname], chans[si]
        # transition (registered BEFORE the stage, pose_hrnet.py:297,309,321)
        for i, c in enumerate(cur):
            q = 'transition%d.%d.' % (si + 1, i)
            if i < len(pre):
                if c != pre[i]:
                    conv(q + '0', c, pre[i], 3); bn(q + '1', c)
            else:
                for j in range(i + 1 - len(pre)):
                    cout = c if j == i - len(pre) else pre[-1]
                    conv('%s%d.0' % (q, j), cout, pre[-1], 3); bn('%s%d.1' % (q, j), cout)
        inch = list(cur)
        for m in range(sc['NUM_MODULES']):
            p = 'stage%d.%d.' % (si + 2, m)
            exp = EXPANSION[sc['BLOCK']]
            for i in range(sc['NUM_BRANCHES']):
                planes = sc['NUM_CHANNELS'][i]
                for b in range(sc['NUM_BLOCKS'][i]):
                    cin = inch[i] if b == 0 else planes * exp
                    block(sc['BLOCK'], '%sbranches.%d.%d.' % (p, i, b), cin, planes, downsample=(b == 0 and cin != planes * exp))
                inch[i] = planes * exp
            last = (sname == 'STAGE4') and m == sc['NUM_MODULES'] - 1
            nb = sc['NUM_BRANCHES']
            if nb > 1:
                for i in range(1 if last else nb):
                    for j in range(nb):
                        q = '%sfuse_layers.%d.%d.' % (p, i, j)
                        if j > i:
                            conv(q + '0', inch[i], inch[j], 1); bn(q + '1', inch[i])
                        elif j < i:
                            for k in range(i - j):
                                cout = inch[i] if k == i - j - 1 else inch[j]
                                conv('%s%d.0' % (q, k), cout, inch[j], 3); bn('%s%d.1' % (q, k), cout)
        pre = cur
    conv('final_layer', num_joints, chans[2][0], cfg.get('FINAL_CONV_KERNEL', 1), bias=True)
    return keys
