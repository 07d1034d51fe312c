"""`models.hourglass` with the reference's module API, executed by the MI355X HIP path.

API mirrored from /root/reference/lib/models/hourglass.py:
  * get_pose_net(cfg, is_train, **kw) -> nn.Module                      (:195-197)
  * forward(x[N,3,H,W] fp32) -> list of NUM_STACKS tensors [N,J,H/4,W/4] (:170-192)
  * state_dict keys / shapes identical (conv OIHW, BN weight/bias/running_*/num_batches_tracked), so
    reference checkpoints load with strict=True (tools/fpd_train.py:139-141, lib/utils/utils.py:250-255)
  * .train()/.eval() switch BatchNorm between batch and running statistics (lib/core/function.py:110-111)
What differs by design: parameters are views into one flat HBM arena (conv weights channels-last), the
forward/backward run as a recorded plan of hand-written gfx950 kernels, and a CPU input is an error --
there is no eager/CPU fallback (the reference's CPU dry-runs for add_graph/get_model_summary at
tools/fpd_train.py:162-167 must be given a CUDA tensor).
"""
import math

import torch
import torch.nn as nn

from ... import executor as E
from ... import graph as G
from ... import runtime as R

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


class _Node(nn.Module):
    """Container reproducing one level of the reference's module tree (names only; no compute)."""


class _HourglassFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, model, inst, x, *params):
        inst.image().copy_(x)
        inst.run('prep')
        inst.run('fwd')
        ctx.model, ctx.inst = model, inst
        outs = []
        l, st = R.lib(), R.current_stream()
        for i, o in enumerate(inst.g.outputs):
            n, h, w, c = o.shape
            t = torch.empty((n, c, h, w), dtype=torch.float32, device=x.device)
            R.check(l.fpd_nhwc_to_nchw(inst.A.ptr(o.buf), t.data_ptr(), n, c, h, w, inst.dtype, st), 'nhwc_to_nchw')
            outs.append(t)
        return tuple(outs)

    @staticmethod
    def backward(ctx, *gouts):
        inst, model = ctx.inst, ctx.model
        l, st = R.lib(), R.current_stream()
        for i, (o, g) in enumerate(zip(inst.g.out_grads, gouts)):
            n, h, w, c = o.shape
            if g is None:
                inst.A.view(o.buf).zero_()
                continue
            g = g.contiguous().float()
            R.check(l.fpd_nchw_to_nhwc(g.data_ptr(), inst.A.ptr(o.buf), n, c, h, w, inst.dtype, st), 'nchw_to_nhwc')
        inst.run('bwd')
        model._attach_grads()
        return (None, None, None) + (None,) * (len(ctx.needs_input_grad) - 3)


class HourglassNet(nn.Module):
    """Stacked hourglass (Newell et al.) student/teacher of the FPD path, HIP-backed."""

    def __init__(self, cfg, **kwargs):
        super().__init__()
        extra = cfg.MODEL.EXTRA
        self.cfg_hg = {'F': int(extra.NUM_FEATURES), 'S': int(extra.NUM_STACKS), 'J': int(cfg.MODEL.NUM_JOINTS),
                       'num_blocks': int(extra.NUM_BLOCKS)}
        self.num_stacks = self.cfg_hg['S']
        try:                                  # optional MODEL.DTYPE: 'fp32' (parity build) | 'bf16' (throughput build)
            cfg_dt = cfg.MODEL['DTYPE'] if 'DTYPE' in cfg.MODEL else 'fp32'
        except TypeError:
            cfg_dt = getattr(cfg.MODEL, 'DTYPE', 'fp32')
        dt = str(kwargs.get('dtype', cfg_dt))
        self.fpd_dtype = R.BF16 if dt in ('bf16', 'bfloat16') else R.F32
        keys = hourglass_keys(self.cfg_hg['F'], self.cfg_hg['S'], self.cfg_hg['J'], self.cfg_hg['num_blocks'])
        self.table = G.ParamTable(keys, bucket_of=G.hourglass_bucket_of(self.cfg_hg['S']))
        self._flat = {n: torch.zeros(max(self.table.sizes[n], 4), dtype=torch.int64 if n == 'nbt' else torch.float32)
                      for n in ('param', 'rstat', 'nbt')}
        self._flat_grad = None
        self._state = None
        self._instances = {}
        self._build_tree()
        self.reset_parameters()

    # ---- module tree with the reference's key names, tensors = views of the flat arenas ----
    def _view(self, key, grad=False):
        b = self.table[key]
        flat = self._flat_grad if grad else self._flat[b.arena]
        v = flat[b.off:b.off + b.numel].view(b.shape)
        if len(b.shape) == 4:
            v = v.permute(0, 3, 1, 2)          # K,R,S,C storage presented as the reference's OIHW tensor
        return v

    def _build_tree(self):
        for key, _ in self.table.keys:
            parts = key.split('.')
            node = self
            for p in parts[:-1]:
                if p not in node._modules:
                    node.add_module(p, _Node())
                node = node._modules[p]
            leaf = parts[-1]
            if self.table[key].arena == 'param':
                node.register_parameter(leaf, nn.Parameter(self._view(key)))
            else:
                node.register_buffer(leaf, self._view(key))

    def _relink(self):
        for key, _ in self.table.keys:
            parts = key.split('.')
            node = self
            for p in parts[:-1]:
                node = node._modules[p]
            if self.table[key].arena == 'param':
                node._parameters[parts[-1]].data = self._view(key)
            else:
                node._buffers[parts[-1]] = self._view(key)

    def _apply(self, fn, recurse=True):
        # move / cast the flat arenas as a whole, then re-point every parameter and buffer at them
        for n in self._flat:
            t = fn(self._flat[n])
            self._flat[n] = t if n == 'nbt' else t.float()
        self._flat_grad = None
        self._state = None
        self._instances = {}
        self._relink()
        return self

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

    # ---- device state / plans ----
    def device_state(self):
        if self._state is None:
            dev = self._flat['param'].device
            if dev.type != 'cuda':
                raise R.FpdError('HourglassNet must be on a CUDA (ROCm) device: call .cuda() first; no CPU fallback')
            R.lib()
            st = E.ModelState.__new__(E.ModelState)
            st.table, st.device, st.dtype = self.table, dev, self.fpd_dtype
            st.A = E.Arenas(dev, self.fpd_dtype)
            for n in ('param', 'rstat', 'nbt'):
                st.A.t[n] = self._flat[n]
            self._flat_grad = torch.zeros_like(self._flat['param'])
            st.A.t['grad'] = self._flat_grad
            self._state = st
        return self._state

    def _attach_grads(self):
        for key in self.table.trainable_keys():
            parts = key.split('.')
            node = self
            for p in parts[:-1]:
                node = node._modules[p]
            node._parameters[parts[-1]].grad = self._view(key, grad=True)

    def instance(self, shape, train):
        key = (tuple(shape), bool(train))
        if key not in self._instances:
            st = self.device_state()
            n, c, h, w = shape
            assert c == 3, 'expected an RGB image batch [N,3,H,W]'
            self._instances[key] = E.GraphInstance(st, self.cfg_hg, n, h, w, train=train).finalize()
        return self._instances[key]

    def forward(self, x):
        if not x.is_cuda:
            raise R.FpdError('fpd_amd HourglassNet.forward needs a CUDA (ROCm) tensor; there is no CPU path')
        x = x.float().contiguous()
        inst = self.instance(x.shape, self.training)
        if self.training and torch.is_grad_enabled():
            params = [p for p in self.parameters()]
            return list(_HourglassFn.apply(self, inst, x, *params))
        inst.image().copy_(x)
        inst.run('prep')
        inst.run('fwd')
        outs = []
        l, st = R.lib(), R.current_stream()
        for o in inst.g.outputs:
            n, h, w, c = o.shape
            t = torch.empty((n, c, h, w), dtype=torch.float32, device=x.device)
            R.check(l.fpd_nhwc_to_nchw(inst.A.ptr(o.buf), t.data_ptr(), n, c, h, w, inst.dtype, st), 'nhwc_to_nchw')
            outs.append(t)
        return outs


def get_pose_net(cfg, is_train, **kwargs):
    """Same factory signature as the reference (is_train is ignored there too, hourglass.py:195-197)."""
    return HourglassNet(cfg, **kwargs)
