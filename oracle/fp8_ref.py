"""TEST INFRASTRUCTURE (oracle) -- specification of the fp8 forward convolution (csrc/conv_tile_f8.hip, BASELINE
configs[4] "HRNet-W32 student fp8 weights (CDNA4 fp8 MFMA)") as fake quantisation in torch on the CPU.
Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.

The reference has no fp8 path (its arithmetic is torch fp32, /root/reference/lib/models/pose_hrnet.py): this file DEFINES
the numerics the kernel must reproduce -- OCP e4m3fn (torch.float8_e4m3fn: round to nearest even, here saturating at
+-448), one fp32 scale per output channel for the weights (amax / 448), unit scale for the activations, exact products
accumulated in fp32 -- and the accuracy tests (tests/test_fp8_gpu.py) bound its distance from the reference's fp32
convolution.  PARITY: pinned to the reference only through that bound (there is nothing in the reference to pin an fp8
result to)."""
import torch
import torch.nn.functional as F

E4M3_MAX = 448.0


def quant_e4m3(x):
    """fp32 -> e4m3fn (saturating) -> fp32."""
    return x.float().clamp(-E4M3_MAX, E4M3_MAX).to(torch.float8_e4m3fn).float()


def quant_weights(w):
    """w [K, ...] fp32 master weights -> (q as fp32 values on the e4m3 grid, scale [K]); w ~= q * scale."""
    k = w.shape[0]
    amax = w.reshape(k, -1).abs().amax(dim=1)
    scale = torch.where(amax > 0, amax / E4M3_MAX, torch.ones_like(amax))
    q = quant_e4m3(w / scale.view(-1, *([1] * (w.dim() - 1))))
    return q, scale


def conv_f8(x_nhwc, w_krsc, stride=1, pad=0):
    """x [N,H,W,C] fp32 (the operand AFTER the BN+ReLU prologue, not yet rounded), w [K,R,S,C] fp32 master weights ->
    y [N,P,Q,K] fp32 (before bias / residual)."""
    q, scale = quant_weights(w_krsc)
    xq = quant_e4m3(x_nhwc)
    y = F.conv2d(xq.permute(0, 3, 1, 2).double(), q.permute(0, 3, 1, 2).double(), None, stride=stride, padding=pad)
    return (y.permute(0, 2, 3, 1) * scale.double()).float()
