"""What the HIP-backed models share: parameters and BN buffers as views of flat HBM arenas (graph.ParamTable layout, conv
weights stored K,R,S,C = the reference's OIHW tensor in channels-last memory order), storage-less leaf modules with the
torch classes' metadata, the autograd bridge and the plan cache.  The model classes (hourglass.py, pose_hrnet.py) add the
reference's module tree and initialisation."""
import torch
import torch.nn as nn

from ... import executor as E
from ... import runtime as R

BN_MOMENTUM = 0.1


def _no_eager(self, *a, **kw):
    raise R.FpdError('%s is executed inside the fused HIP plan of its network (model(x) on a CUDA tensor); the '
                     'leaf modules hold parameters/metadata only -- there is no eager torch path' % type(self).__name__)


class Conv2d(nn.Conv2d):
    """nn.Conv2d metadata + parameters (views of the model's flat arena); storage-less construction."""

    def __init__(self, cin, cout, k, stride=1, padding=0, bias=True):
        super().__init__(cin, cout, k, stride=stride, padding=padding, bias=bias, device='meta')
    forward = _no_eager


class BatchNorm2d(nn.BatchNorm2d):
    def __init__(self, c):
        super().__init__(c, momentum=BN_MOMENTUM, device='meta')
    forward = _no_eager


class ReLU(nn.ReLU):
    forward = _no_eager


class MaxPool2d(nn.MaxPool2d):
    forward = _no_eager


class Upsample(nn.Upsample):
    forward = _no_eager


class _NetFn(torch.autograd.Function):
    """Module-API bridge: forward = the recorded plan, backward = its reverse-mode plan (gradients land in the flat arena)."""

    @staticmethod
    def forward(ctx, model, inst, x, *params):
        inst.image().copy_(x)
        inst.run('prep')
        inst.run('fwd')
        ctx.model, ctx.inst = model, inst
        return tuple(model._outputs_nchw(inst))

    @staticmethod
    def backward(ctx, *gouts):
        inst, model = ctx.inst, ctx.model
        l, st = R.lib(), R.current_stream()
        for o, g in zip(inst.g.out_grads, gouts):
            n, h, w, c = o.shape
            if g is None:
                inst.A.view(o.buf).zero_()
                continue
            g = g.contiguous().float()
            R.check(l.fpd_nchw_to_nhwc(g.data_ptr(), inst.A.ptr(o.buf), n, c, h, w, inst.dtype, st), 'nchw_to_nhwc')
        inst.run('bwd')
        model._attach_grads()
        return (None, None, None) + (None,) * (len(ctx.needs_input_grad) - 3)


class FlatArenaNet(nn.Module):
    """Base of the HIP-backed networks.  Subclasses set self.cfg_hg (the dict executor.GraphInstance builds the op graph
    from), self.fpd_dtype, call _init_flat(table), build their module tree and call _bind_tree()."""

    def _init_flat(self, table):
        self.table = table
        self._flat = {n: torch.zeros(max(table.sizes[n], 4), dtype=torch.int64 if n == 'nbt' else torch.float32)
                      for n in ('param', 'rstat', 'nbt')}
        self._flat_grad = None
        self._state = None
        self._instances = {}

    @staticmethod
    def _dtype_from(cfg, kwargs):
        try:                                  # optional MODEL.DTYPE: 'fp32' (parity build) | 'bf16' (throughput build)
            cfg_dt = cfg.MODEL['DTYPE'] if 'DTYPE' in cfg.MODEL else 'fp32'
        except TypeError:
            cfg_dt = getattr(cfg.MODEL, 'DTYPE', 'fp32')
        dt = str(kwargs.get('dtype', cfg_dt))
        return R.BF16 if dt in ('bf16', 'bfloat16') else R.F32

    # ---- tensors = views of the flat arenas ----
    def _view(self, key, grad=False):
        b = self.table[key]
        flat = self._flat_grad if grad else self._flat[b.arena]
        v = flat[b.off:b.off + b.numel].view(b.shape)
        if len(b.shape) == 4:
            v = v.permute(0, 3, 1, 2)          # K,R,S,C storage presented as the reference's OIHW tensor
        return v

    def _owner(self, key):
        parts = key.split('.')
        node = self
        for p in parts[:-1]:
            node = node._modules[p]
        return node, parts[-1]

    def _bind_tree(self):
        """Point every parameter / buffer of the (storage-less) module tree at its view of the flat arenas."""
        for key, _ in self.table.keys:
            node, leaf = self._owner(key)
            if self.table[key].arena == 'param':
                node._parameters[leaf] = nn.Parameter(self._view(key))
            else:
                node._buffers[leaf] = self._view(key)
        meta = [k for k, v in list(self.named_parameters()) + list(self.named_buffers()) if v.is_meta]
        assert not meta, 'module tree and key table disagree: %r' % meta[:4]
        assert [k for k in self.state_dict()] == [k for k, _ in self.table.keys], 'state_dict order != key table'

    def _relink(self):
        for key, _ in self.table.keys:
            node, leaf = self._owner(key)
            if self.table[key].arena == 'param':
                node._parameters[leaf].data = self._view(key)
            else:
                node._buffers[leaf] = self._view(key)

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

    # ---- device state / plans ----
    def device_state(self):
        if self._state is None:
            dev = self._flat['param'].device
            if dev.type != 'cuda':
                raise R.FpdError('%s must be on a CUDA (ROCm) device: call .cuda() first; no CPU fallback' % type(self).__name__)
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
            node, leaf = self._owner(key)
            node._parameters[leaf].grad = self._view(key, grad=True)

    def instance(self, shape, train):
        key = (tuple(shape), bool(train))
        if key not in self._instances:
            st = self.device_state()
            n, c, h, w = shape
            assert c == 3, 'expected an RGB image batch [N,3,H,W]'
            self._instances[key] = E.GraphInstance(st, self.cfg_hg, n, h, w, train=train).finalize()
        return self._instances[key]

    def _outputs_nchw(self, inst):
        l, st = R.lib(), R.current_stream()
        outs = []
        for o in inst.g.outputs:
            n, h, w, c = o.shape
            t = torch.empty((n, c, h, w), dtype=torch.float32, device=inst.A.device)
            R.check(l.fpd_nhwc_to_nchw(inst.A.ptr(o.buf), t.data_ptr(), n, c, h, w, inst.dtype, st), 'nhwc_to_nchw')
            outs.append(t)
        return outs

    RETURNS_LIST = True        # hourglass: list of per-stack maps; HRNet: one tensor (callers branch on isinstance(list))

    def forward(self, x):
        if not x.is_cuda:
            raise R.FpdError('%s.forward needs a CUDA (ROCm) tensor; there is no CPU path' % type(self).__name__)
        x = x.float().contiguous()
        inst = self.instance(x.shape, self.training)
        if self.training and torch.is_grad_enabled():
            outs = list(_NetFn.apply(self, inst, x, *list(self.parameters())))
        else:
            inst.image().copy_(x)
            inst.run('prep')
            inst.run('fwd')
            outs = self._outputs_nchw(inst)
        return outs if self.RETURNS_LIST else outs[0]
