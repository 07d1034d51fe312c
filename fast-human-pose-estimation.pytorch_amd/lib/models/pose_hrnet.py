"""`models.pose_hrnet` with the reference's module API, executed by the MI355X HIP path.

API mirrored from /root/reference/lib/models/pose_hrnet.py:
  * get_pose_net(cfg, is_train, **kw) -> nn.Module; `is_train and cfg.MODEL.INIT_WEIGHTS` -> init_weights(cfg.MODEL.PRETRAINED)
    (:495-501)
  * PoseHighResolutionNet.forward(x[N,3,H,W] fp32) -> ONE tensor [N,J,H/4,W/4] (:425-460; lib/core/function.py branches on
    isinstance(outputs, list))
  * state_dict keys / shapes / order identical to the reference (conv1, bn1, conv2, bn2, layer1, transition1, stage2,
    transition2, stage3, transition3, stage4, final_layer; convolutions bias-free except final_layer), so reference HRNet
    checkpoints load with strict=True and `init_weights`' PRETRAINED_LAYERS filter works on the same names
  * cfg access exactly as the reference does it: cfg.MODEL.EXTRA attributes and cfg['MODEL']['EXTRA']['STAGEk'] items with
    NUM_MODULES / NUM_BRANCHES / NUM_BLOCKS / NUM_CHANNELS / BLOCK / FUSE_METHOD, FINAL_CONV_KERNEL, PRETRAINED_LAYERS
The compute is graph.HRNetGraph (one recorded plan of gfx950 kernels); leaf modules are torch-typed (nn.Conv2d /
nn.BatchNorm2d subclasses with the reference's kernel size / stride / padding) but storage-less and never called; the
containers carry the reference's names (the reference's container CLASSES -- BasicBlock, HighResolutionModule -- are plain
nn.Module nodes here: state_dict / named_parameters / isinstance(m, nn.Conv2d) loops behave identically, class names differ).
"""
import logging
import os

import torch
import torch.nn as nn

from ... import graph as G
from ... import runtime as R
from ._flat import BatchNorm2d, Conv2d, FlatArenaNet, ReLU

logger = logging.getLogger(__name__)
EXPANSION = G.HRNET_EXPANSION


def _plain(d):
    """cfg node (yacs-like attr dict or plain dict) -> plain nested dict / lists."""
    if isinstance(d, dict):
        return {k: _plain(v) for k, v in d.items()}
    if isinstance(d, (list, tuple)):
        return [_plain(v) for v in d]
    return d


def hrnet_keys(extra, num_joints):
    """[(key, shape)] in the reference's state_dict order (module registration order, pose_hrnet.py:276-331)."""
    keys = []

    def conv(name, cout, cin, k, bias=False):
        keys.append((name + '.weight', (cout, cin, k, k)))
        if bias:
            keys.append((name + '.bias', (cout,)))

    def bn(name, c):
        keys.extend([(name + '.weight', (c,)), (name + '.bias', (c,)), (name + '.running_mean', (c,)),
                     (name + '.running_var', (c,)), (name + '.num_batches_tracked', ())])

    def block(kind, p, cin, planes, down):
        if kind == 'BASIC':
            conv(p + 'conv1', planes, cin, 3); bn(p + 'bn1', planes)
            conv(p + 'conv2', planes, planes, 3); bn(p + 'bn2', planes)
        else:
            conv(p + 'conv1', planes, cin, 1); bn(p + 'bn1', planes)
            conv(p + 'conv2', planes, planes, 3); bn(p + 'bn2', planes)
            conv(p + 'conv3', 4 * planes, planes, 1); bn(p + 'bn3', 4 * planes)
        if down:
            conv(p + 'downsample.0', planes * EXPANSION[kind], cin, 1); bn(p + 'downsample.1', planes * EXPANSION[kind])

    conv('conv1', 64, 3, 3); bn('bn1', 64)
    conv('conv2', 64, 64, 3); bn('bn2', 64)
    for b in range(4):
        block('BOTTLENECK', 'layer1.%d.' % b, 64 if b == 0 else 256, 64, b == 0)
    pre = [256]
    for si, sname in enumerate(('STAGE2', 'STAGE3', 'STAGE4')):
        sc = extra[sname]
        exp = EXPANSION[sc['BLOCK']]
        cur = [c * exp for c in sc['NUM_CHANNELS']]
        for i, c in enumerate(cur):                      # transition layers are registered before their stage (:297,309,321)
            q = 'transition%d.%d.' % (si + 1, i)
            if i < len(pre):
                if c != pre[i]:
                    conv(q + '0', c, pre[i], 3); bn(q + '1', c)
            else:
                for j in range(i + 1 - len(pre)):
                    cout = c if j == i - len(pre) else pre[-1]
                    conv('%s%d.0' % (q, j), cout, pre[-1], 3); bn('%s%d.1' % (q, j), cout)
        inch = list(cur)
        nb = sc['NUM_BRANCHES']
        for m in range(sc['NUM_MODULES']):
            p = 'stage%d.%d.' % (si + 2, m)
            for i in range(nb):
                planes = sc['NUM_CHANNELS'][i]
                for b in range(sc['NUM_BLOCKS'][i]):
                    cin = inch[i] if b == 0 else planes * exp
                    block(sc['BLOCK'], '%sbranches.%d.%d.' % (p, i, b), cin, planes, b == 0 and cin != planes * exp)
                inch[i] = planes * exp
            last = sname == 'STAGE4' and m == sc['NUM_MODULES'] - 1           # multi_scale_output=False (:320-321)
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
    conv('final_layer', num_joints, pre[0], extra.get('FINAL_CONV_KERNEL', 1), bias=True)
    return keys


class _Node(nn.Module):
    """A container of the reference's module tree (name only)."""


class PoseHighResolutionNet(FlatArenaNet):
    RETURNS_LIST = False

    def __init__(self, cfg, **kwargs):
        super().__init__()
        extra = _plain(cfg['MODEL']['EXTRA'])
        self.extra = {k: extra[k] for k in ('STAGE2', 'STAGE3', 'STAGE4')}
        self.extra['FINAL_CONV_KERNEL'] = int(extra.get('FINAL_CONV_KERNEL', 1))
        for s in ('STAGE2', 'STAGE3', 'STAGE4'):
            if self.extra[s].get('FUSE_METHOD', 'SUM') != 'SUM':
                raise ValueError('only FUSE_METHOD SUM is built (every experiments/*.yaml of the reference uses it)')
        self.pretrained_layers = list(extra.get('PRETRAINED_LAYERS', ['*']))
        J = int(cfg.MODEL.NUM_JOINTS)
        self.cfg_hg = {'arch': 'hrnet', 'extra': self.extra, 'J': J}
        self.fpd_dtype = self._dtype_from(cfg, kwargs)
        # optional MODEL.WEIGHT_DTYPE 'fp8' (BASELINE configs[4]): forward convolutions of the bf16 build run with e4m3
        # weights (per-output-channel scales) and e4m3 activations on the CDNA4 fp8 matrix pipe (csrc/conv_tile_f8.hip);
        # master weights, gradients and Adam stay fp32 / bf16
        try:
            wd = cfg.MODEL['WEIGHT_DTYPE'] if 'WEIGHT_DTYPE' in cfg.MODEL else ''
        except TypeError:
            wd = getattr(cfg.MODEL, 'WEIGHT_DTYPE', '')
        wd = str(kwargs.get('weight_dtype', wd)).lower()
        if wd not in ('', 'none', 'bf16', 'fp32', 'fp8', 'e4m3'):
            raise ValueError('MODEL.WEIGHT_DTYPE %r: expected fp8 or nothing' % wd)
        self.cfg_hg['fp8'] = wd in ('fp8', 'e4m3')
        if self.cfg_hg['fp8'] and self.fpd_dtype != R.BF16:
            raise ValueError('MODEL.WEIGHT_DTYPE fp8 needs MODEL.DTYPE bf16 (activations are stored as bf16)')
        keys = hrnet_keys(self.extra, J)
        self._init_flat(G.ParamTable(keys, bucket_of=G.hrnet_bucket_of))
        self._build_tree()
        self._bind_tree()
        self.reset_parameters()

    def _build_tree(self):
        """Containers by name; typed leaves with the reference's conv geometry (taken from the op graph of a dummy shape)."""
        g = G.HRNetGraph(self.table, self.extra, self.cfg_hg['J'], 1, 64, 64, train=False)
        geom = {o.wkey[:-7]: (o.dims[7], o.dims[8]) for o in g.fwd if o.kind == 'conv'}        # name -> (stride, pad)
        self.relu = None
        for key, shp in self.table.keys:
            parts = key.split('.')
            node = self
            for i, p in enumerate(parts[:-2]):
                if p not in node._modules:
                    node.add_module(p, _Node())
                    if p == 'layer1' and node is self and self.relu is None:      # registration order: conv1 bn1 conv2 bn2 relu layer1
                        pass
                node = node._modules[p]
            leaf_name, field = parts[-2], parts[-1]
            if leaf_name in node._modules:
                continue
            full = '.'.join(parts[:-1])
            if field == 'weight' and len(shp) == 4:
                stride, pad = geom[full]
                node.add_module(leaf_name, Conv2d(shp[1], shp[0], shp[2], stride=stride, padding=pad,
                                                  bias=(full + '.bias') in self.table.entries))
            else:
                node.add_module(leaf_name, BatchNorm2d(shp[0]))
        # the reference registers one shared ReLU after bn2 (parameter-free: no effect on state_dict order)
        self.relu = ReLU(inplace=True)

    def reset_parameters(self):
        """torch defaults (what PoseHighResolutionNet.__init__ leaves behind when INIT_WEIGHTS is off): conv weight/bias
        U(+-1/sqrt(fan_in)), BN weight 1 / bias 0 / running mean 0 / var 1."""
        import math
        with torch.no_grad():
            for key, shp in self.table.keys:
                v = self._view(key)
                if key.endswith('num_batches_tracked') or key.endswith('running_mean'):
                    v.zero_()
                elif key.endswith('running_var'):
                    v.fill_(1.0)
                elif len(shp) == 4:
                    v.uniform_(-1.0 / math.sqrt(shp[1] * shp[2] * shp[3]), 1.0 / math.sqrt(shp[1] * shp[2] * shp[3]))
                else:
                    base = key.rsplit('.', 1)[0]
                    if (base + '.running_mean') in self.table.entries:
                        v.fill_(1.0) if key.endswith('.weight') else v.zero_()
                    else:
                        ws = self.table.logical[base + '.weight']
                        bound = 1.0 / math.sqrt(ws[1] * ws[2] * ws[3])
                        v.uniform_(-bound, bound)

    def init_weights(self, pretrained=''):
        """pose_hrnet.py:462-492: N(0, 0.001) conv weights, zero conv bias, BN (1, 0); then the PRETRAINED_LAYERS-filtered,
        non-strict load of `pretrained` if that file exists (an unknown non-empty path is an error, as in the reference)."""
        logger.info('=> init weights from normal distribution')
        with torch.no_grad():
            for m in self.modules():
                if isinstance(m, nn.Conv2d):
                    nn.init.normal_(m.weight, std=0.001)
                    if m.bias is not None:
                        nn.init.constant_(m.bias, 0)
                elif isinstance(m, nn.BatchNorm2d):
                    nn.init.constant_(m.weight, 1)
                    nn.init.constant_(m.bias, 0)
        if os.path.isfile(pretrained):
            sd = torch.load(pretrained, map_location='cpu')
            logger.info('=> loading pretrained model {}'.format(pretrained))
            need = {k: v for k, v in sd.items() if k.split('.')[0] in self.pretrained_layers or self.pretrained_layers[0] == '*'}
            self.load_state_dict(need, strict=False)
        elif pretrained:
            logger.error('=> please download pre-trained models first!')
            raise ValueError('{} is not exist!'.format(pretrained))


def get_pose_net(cfg, is_train, **kwargs):
    """pose_hrnet.py:495-501."""
    model = PoseHighResolutionNet(cfg, **kwargs)
    if is_train and cfg.MODEL.INIT_WEIGHTS:
        model.init_weights(cfg.MODEL.PRETRAINED)
    return model
