"""Optimizer factory and checkpoint I/O with the reference's interface (/root/reference/lib/utils/utils.py:59-83,204-258)."""
import os

import torch

from ... import runtime as R


class FusedAdam(torch.optim.Optimizer):
    """torch.optim.Adam(lr) semantics (utils.py:69-73: betas (0.9, 0.999), eps 1e-8, no weight decay, no amsgrad) as ONE
    HIP launch over the model's flat parameter / gradient arenas instead of 752 per-tensor launches.  lr is read from
    param_groups[0]['lr'] at every step, so torch LR schedulers (MultiStepLR, tools/fpd_train.py:236-239) work."""

    def __init__(self, model, lr=1e-3, betas=(0.9, 0.999), eps=1e-8):
        self.model = model.module if hasattr(model, 'module') else model
        super().__init__(list(self.model.parameters()), dict(lr=lr, betas=betas, eps=eps))
        flat = self.model._flat['param']         # the flat fp32 arena every parameter is a view of (create the optimizer
        self.m = torch.zeros_like(flat)          # AFTER .to(device), like any torch optimizer)
        self.v = torch.zeros_like(flat)
        self.lr_dev = torch.full((1,), lr, dtype=torch.float32, device=flat.device)
        self.step_dev = torch.zeros(1, dtype=torch.int64, device=flat.device)
        self._lr_host = lr

    def sync_lr(self):
        """param_groups[0]['lr'] (what LR schedules write) -> the device scalar the Adam kernel reads."""
        lr = float(self.param_groups[0]['lr'])
        if lr != self._lr_host:
            self.lr_dev.fill_(lr)
            self._lr_host = lr

    def adam_args(self):
        st = self.model.device_state()
        g = self.param_groups[0]
        a = R.AdamT()
        a.n = st.table.sizes['param']
        a.param, a.grad = st.A.tensor('param').data_ptr(), st.A.tensor('grad').data_ptr()
        a.m, a.v, a.param_lp = self.m.data_ptr(), self.v.data_ptr(), None
        a.lr, a.beta1, a.beta2, a.eps = g['lr'], g['betas'][0], g['betas'][1], g['eps']
        a.bias_corr1 = a.bias_corr2 = 1.0
        a.grad_scale = 1.0
        a.lr_dev, a.step_dev = self.lr_dev.data_ptr(), self.step_dev.data_ptr()
        return a

    @torch.no_grad()
    def step(self, closure=None):
        self.sync_lr()
        R.check(R.lib().fpd_adam(self.adam_args(), R.current_stream()), 'fpd_adam')

    def zero_grad(self, set_to_none=True):
        st = self.model.device_state()
        st.A.tensor('grad').zero_()

    # ---- checkpoint interop: the state dict has torch.optim.Adam's layout (per-parameter exp_avg / exp_avg_sq in the
    #      reference's OIHW shapes), so `checkpoint['optimizer']` written here loads into the reference's Adam and a
    #      reference checkpoint resumes here (tools/fpd_train.py:224-234, AUTO_RESUME) ----
    def _views(self, flat):
        m = self.model
        out = []
        for key in m.table.trainable_keys():
            b = m.table[key]
            v = flat[b.off:b.off + b.numel].view(b.shape)
            out.append(v.permute(0, 3, 1, 2) if len(b.shape) == 4 else v)
        return out

    def state_dict(self):
        step = self.step_dev.to(torch.float32).reshape(())
        state = {}
        if int(self.step_dev.item()) > 0:
            for i, (m, v) in enumerate(zip(self._views(self.m), self._views(self.v))):
                state[i] = {'step': step.clone(), 'exp_avg': m.clone(), 'exp_avg_sq': v.clone()}
        g = {k: v for k, v in self.param_groups[0].items() if k != 'params'}
        g.setdefault('weight_decay', 0)
        g.setdefault('amsgrad', False)
        g['params'] = list(range(len(self.param_groups[0]['params'])))
        return {'state': state, 'param_groups': [g]}

    def load_state_dict(self, sd):
        if 'm' in sd and 'v' in sd:
            # the round-1 checkpoints held the two moment arenas as flat tensors in the parameter order of THAT round; the
            # table has been re-ordered since (backward-completion buckets), so copying them would hand every parameter
            # another parameter's moments.  Refuse instead of mis-assigning.
            raise ValueError('optimizer state in the retired flat {m, v, step} layout cannot be mapped onto the current '
                             'parameter order; resume from a torch.optim.Adam-style state_dict (state / param_groups)')
        st = sd.get('state', {})
        mv, vv = self._views(self.m), self._views(self.v)
        if len(st) not in (0, len(mv)):
            raise ValueError('optimizer state has %d parameter entries, the model has %d' % (len(st), len(mv)))
        self.m.zero_(); self.v.zero_(); self.step_dev.zero_()
        ids = sd['param_groups'][0]['params']
        for dst_m, dst_v, pid in zip(mv, vv, ids):
            e = st.get(pid)
            if e is None:
                continue
            dst_m.copy_(e['exp_avg']); dst_v.copy_(e['exp_avg_sq'])
            self.step_dev.fill_(int(float(e['step'])))
        g = sd['param_groups'][0]
        for k in ('lr', 'betas', 'eps', 'initial_lr'):   # initial_lr: what torch's LR schedulers resume from
            if k in g:
                self.param_groups[0][k] = tuple(g[k]) if k == 'betas' else g[k]
        self.param_groups[0].setdefault('initial_lr', self.param_groups[0]['lr'])
        self._lr_host = None
        self.sync_lr()


def multistep_lr(base_lr, milestones, gamma, epoch):
    """Learning rate the reference trains epoch `epoch` with (tools/fpd_train.py:236-239,253: MultiStepLR built with
    last_epoch = -1 and stepped at the START of every epoch, so epoch e runs at the scheduler's value for e + 1, i.e.
    a milestone m takes effect in epoch m - 1).  Closed form, so a resumed run continues exactly where an
    uninterrupted one would be (torch's chainable scheduler form re-applies a milestone that falls on the resume epoch
    when it is handed an already-decayed lr; the reference under torch 1.0 additionally skips one epoch on resume --
    neither artefact is reproduced)."""
    return base_lr * gamma ** sum(1 for m in milestones if m <= epoch + 1)


def get_optimizer(cfg, model):
    """utils.py:59-75."""
    if cfg.TRAIN.OPTIMIZER == 'adam':
        return FusedAdam(model, lr=cfg.TRAIN.LR)
    if cfg.TRAIN.OPTIMIZER == 'sgd':
        # usable through the module API (autograd + .grad views); core.function.fpd_train refuses it (fused Adam only)
        return torch.optim.SGD(model.parameters(), lr=cfg.TRAIN.LR, momentum=cfg.TRAIN.MOMENTUM,
                               weight_decay=cfg.TRAIN.WD, nesterov=cfg.TRAIN.NESTEROV)
    raise ValueError('unknown optimizer %r' % cfg.TRAIN.OPTIMIZER)


def get_model_summary(model, *input_tensors, item_length=26, verbose=False):
    """utils.py:86-202: per-layer table (name, input size, output size, parameters, multiply-adds) collected by forward
    hooks on every non-container module.  The reference obtains it from a CPU dry-run; the HIP-backed models execute as
    one fused plan, so the hooks are fired by `model.shape_forward(input_shape)` instead -- the same modules in the same
    order with shape-only (meta) tensors, which is everything the hooks read.  Same text as the reference's."""
    import os as _os
    import torch.nn as nn
    summary, hooks, instances = [], [], {}

    def hook(module, inp, output):
        cls = type(module).__name__
        instances[cls] = instances.get(cls, 0) + 1
        params = 0
        if 'Conv' in cls or 'BatchNorm' in cls or 'Linear' in cls:
            params = sum(p.numel() for p in module.parameters())
        flops = 'Not Available'
        if 'Conv' in cls and hasattr(module, 'weight'):
            flops = int(module.weight.numel())
            for d in list(output.size())[2:]:
                flops *= int(d)
        elif isinstance(module, nn.Linear):
            flops = int(output.numel()) * int(inp[0].size(1))
        out0 = output[0] if isinstance(output, list) else output
        summary.append(('%s_%d' % (cls, instances[cls]), list(inp[0].size()), list(out0.size()), params, flops))

    target = model.module if hasattr(model, 'module') and hasattr(model.module, 'shape_forward') else model
    def add_hooks(m):                       # utils.py:147-152 via nn.Module.apply: a module shared by several parents
        if not isinstance(m, (nn.ModuleList, nn.Sequential)) and m is not target:      # (the model's ReLU) is visited, and
            hooks.append(m.register_forward_hook(hook))                                # hooked, once per parent
    target.apply(add_hooks)
    target.eval()
    try:
        target.shape_forward(tuple(input_tensors[0].shape))
    finally:
        for h in hooks:
            h.remove()
    sp, nl = item_length, _os.linesep
    rule = '-' * sp * 5 + nl
    details = ''
    if verbose:
        heads = ('Name', 'Input Size', 'Output Size', 'Parameters', 'Multiply Adds (Flops)')
        details = 'Model Summary' + nl + ''.join(h + ' ' * (sp - len(h)) for h in heads) + nl + rule
    params_sum = flops_sum = 0
    for row in summary:
        params_sum += row[3]
        if row[4] != 'Not Available':
            flops_sum += row[4]
        if verbose:
            details += ''.join(str(c) + ' ' * (sp - len(str(c))) for c in row) + nl + rule
    details += nl + 'Total Parameters: {:,}'.format(params_sum) + nl + rule
    details += 'Total Multiply Adds (For Convolution and Linear Layers only): {:,} GFLOPs'.format(flops_sum / (1024 ** 3)) + nl + rule
    details += 'Number of Layers' + nl
    for cls in instances:
        details += '{} : {} layers   '.format(cls, instances[cls])
    return details


def save_checkpoint(states, is_best, output_dir, filename='checkpoint.pth'):
    """utils.py:78-83."""
    torch.save(states, os.path.join(output_dir, filename))
    if is_best and 'state_dict' in states:
        torch.save(states['best_state_dict'], os.path.join(output_dir, 'model_best.pth'))


def load_checkpoint(checkpoint, model, strict=True, model_info=''):
    """utils.py:204-258: accepts a raw state_dict, a DataParallel state_dict ('module.' prefix) or a wrapped dict."""
    sd = torch.load(checkpoint, map_location='cpu') if isinstance(checkpoint, str) else checkpoint
    for key in ('best_state_dict', 'state_dict'):
        if isinstance(sd, dict) and key in sd and isinstance(sd[key], dict):
            sd = sd[key]
            break
    sd = {(k[7:] if k.startswith('module.') else k): v for k, v in sd.items()}
    target = model.module if hasattr(model, 'module') else model
    return target.load_state_dict(sd, strict=strict)
