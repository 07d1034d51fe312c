"""`core.loss.JointsMSELoss` with the reference's interface (/root/reference/lib/core/loss.py:15-39),
computed by the fused HIP loss kernel (csrc/loss_adam.hip).

forward(output[B,J,h,w], target[B,J,h,w], target_weight[B,J,1]) -> 0-dim tensor that supports +=, scalar
multiplication, .item() and .backward(), as lib/core/function.py:128-152 needs.  The closed form the kernel
evaluates, 0.5/(B*J*h*w) * sum w^2 (p-g)^2, equals the reference's per-joint loop.  In the fused training
step (core.function.fpd_train -> executor.FusedFPDStep) the same kernel evaluates the pose and the
distillation term of every stack in one pass; this class is the stand-alone (compatibility) entry."""
import torch
import torch.nn as nn

from ... import runtime as R


class _LossFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, output, target, weight, use_w):
        if not output.is_cuda:
            raise R.FpdError('JointsMSELoss needs CUDA (ROCm) tensors; there is no CPU path')
        l, st = R.lib(), R.current_stream()
        b, j, h, w = output.shape
        if j > 32:
            raise R.FpdError('JointsMSELoss: at most 32 joints supported, got %d' % j)
        dev = output.device
        p = torch.empty((b, h, w, j), dtype=torch.float32, device=dev)
        R.check(l.fpd_nchw_to_nhwc(output.detach().float().contiguous().data_ptr(), p.data_ptr(), b, j, h, w, R.F32, st))
        tgt = target.detach().float().contiguous()
        wt = (weight.detach().float().reshape(b, j).contiguous() if use_w
              else torch.ones((b, j), dtype=torch.float32, device=dev))
        losses = torch.zeros(2, dtype=torch.float64, device=dev)
        dp = torch.empty_like(p)
        a = R.LossT()
        a.B, a.J, a.H, a.W, a.S, a.dtype, a.target_nchw, a.alpha = b, j, h, w, 1, R.F32, 1, 0.0
        a.out[0], a.dout[0] = p.data_ptr(), dp.data_ptr()
        a.teacher, a.target, a.weight, a.losses = p.data_ptr(), tgt.data_ptr(), wt.data_ptr(), losses.data_ptr()
        a.grad_scale = 1.0
        R.check(l.fpd_loss(a, st), 'fpd_loss')
        g = torch.empty((b, j, h, w), dtype=torch.float32, device=dev)
        R.check(l.fpd_nhwc_to_nchw(dp.data_ptr(), g.data_ptr(), b, j, h, w, R.F32, st))
        ctx.save_for_backward(g)
        return losses[0].float()

    @staticmethod
    def backward(ctx, gl):
        (g,) = ctx.saved_tensors
        return g * gl, None, None, None


class JointsMSELoss(nn.Module):
    def __init__(self, use_target_weight):
        super().__init__()
        self.use_target_weight = use_target_weight

    def forward(self, output, target, target_weight):
        return _LossFn.apply(output, target, target_weight, bool(self.use_target_weight))
