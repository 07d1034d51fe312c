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
        st = self.model.device_state()
        flat = st.A.tensor('param')
        self.m = torch.zeros_like(flat)
        self.v = torch.zeros_like(flat)
        self.lr_dev = torch.full((1,), lr, dtype=torch.float32, device=flat.device)
        self.step_dev = torch.zeros(1, dtype=torch.int64, device=flat.device)
        self._lr_host = lr

    def sync_lr(self):
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

    def state_dict(self):
        return {'m': self.m, 'v': self.v, 'step': self.step_dev, 'param_groups': self.param_groups}

    def load_state_dict(self, sd):
        self.m.copy_(sd['m']); self.v.copy_(sd['v']); self.step_dev.copy_(sd['step'])
        self.param_groups[0]['lr'] = sd['param_groups'][0]['lr']


def get_optimizer(cfg, model):
    """utils.py:59-75."""
    if cfg.TRAIN.OPTIMIZER == 'adam':
        return FusedAdam(model, lr=cfg.TRAIN.LR)
    if cfg.TRAIN.OPTIMIZER == 'sgd':
        return torch.optim.SGD(model.parameters(), lr=cfg.TRAIN.LR, momentum=cfg.TRAIN.MOMENTUM,
                               weight_decay=cfg.TRAIN.WD, nesterov=cfg.TRAIN.NESTEROV)
    raise ValueError('unknown optimizer %r' % cfg.TRAIN.OPTIMIZER)


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
