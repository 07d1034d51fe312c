#!/usr/bin/env python
"""CPU study (test infrastructure, imports oracle/): where does the bf16 build's distance from the fp32 reference come
from, and what would an fp32 residual stream buy?  Evaluates the frozen teacher forward of the oracle with rounding hooks:

  ref32      fp32 everywhere (the reference's arithmetic)            truth = fp64
  bf16_all   every stored tensor rounded to bf16 (today's build: stream + bottleneck internals; bf16 weights)
  bf16_int   fp32 residual stream: only bottleneck-internal tensors (and weights) bf16
  bf16_w     only the weights bf16
  autocast   the same oracle graph under torch.autocast(bfloat16) on the CPU (what the REFERENCE gives at this precision)
on a random-init calibrated net and on a net whose weights were trained (plain fp32 Adam on synthetic targets).

    python tools/probes/bf16_study.py [--feats 64 --stacks 4 --size 128 --train-steps 0]
"""
import argparse
import os
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import fpd_ref, hourglass_ref as H  # noqa: E402


def bf(x):
    return x.to(torch.bfloat16).to(x.dtype)


def ident(x):
    return x


def forward(sd, x, S, r_int, r_str, r_w, train=False, depth=4):
    def conv(name, t, stride=1, padding=0):
        return F.conv2d(t, r_w(sd[name + '.weight']), sd[name + '.bias'], stride=stride, padding=padding)

    def bnrelu(name, t):
        return F.relu(H._bn(sd, name, t, train))

    def block(p, t):
        a = conv(p + 'conv1', r_int(bnrelu(p + 'bn1', t)))
        a = conv(p + 'conv2', r_int(bnrelu(p + 'bn2', r_int(a))), padding=1)
        a = conv(p + 'conv3', r_int(bnrelu(p + 'bn3', r_int(a))))
        skip = conv(p + 'downsample.0', t) if (p + 'downsample.0.weight') in sd else t
        return r_str(a + skip)

    def hg(p, n, t):
        q = '%s%d.' % (p, n - 1)
        up1 = block(q + '0.0.', t)
        low = block(q + '1.0.', r_str(F.max_pool2d(t, 2, 2)))
        low = hg(p, n - 1, low) if n > 1 else block(q + '3.0.', low)
        low = block(q + '2.0.', low)
        return r_str(up1 + F.interpolate(low, scale_factor=2, mode='nearest'))
    x = r_str(conv('conv1', x, 2, 3))
    x = r_str(bnrelu('bn1', x))
    x = block('layer1.0.', x)
    x = r_str(F.max_pool2d(x, 2, 2))
    x = block('layer3.0.', block('layer2.0.', x))
    outs = []
    for i in range(S):
        y = block('res.%d.0.' % i, hg('hg.%d.hg.' % i, depth, x))
        y = r_int(bnrelu('fc.%d.1' % i, r_int(conv('fc.%d.0' % i, y))))
        sc = r_str(conv('score.%d' % i, y))
        outs.append(sc)
        if i < S - 1:
            x = r_str(x + conv('fc_.%d' % i, y) + conv('score_.%d' % i, sc))
    return outs


def rel(a, b):
    return float((a.double() - b.double()).norm() / b.double().norm())


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--feats', type=int, default=64)
    ap.add_argument('--stacks', type=int, default=4)
    ap.add_argument('--size', type=int, default=128)
    ap.add_argument('--batch', type=int, default=2)
    ap.add_argument('--train-steps', type=int, default=0)
    args = ap.parse_args()
    torch.manual_seed(0)
    S = args.stacks
    keys = H.hourglass_keys(args.feats, S, 16)
    sd = fpd_ref.synth_state_dict(keys, 2)
    hs = args.size // 4
    xs = [fpd_ref.synth_batch(50 + i, args.batch, 16, (args.size, args.size), (hs, hs)) for i in range(3)]
    if args.train_steps:                       # a trained net: plain fp32 Adam on the synthetic targets
        names = fpd_ref.param_names(sd)
        adam = {}
        for it in range(args.train_steps):
            x, tg, tw = fpd_ref.synth_batch(1000 + it, args.batch, 16, (args.size, args.size), (hs, hs))
            for k in names:
                sd[k].requires_grad_(True); sd[k].grad = None
            outs = H.hourglass_forward(sd, x, S, train=True)
            loss = sum(fpd_ref.joints_mse_loss(o, tg, tw) for o in outs)
            loss.backward()
            g = {k: sd[k].grad.detach().clone() for k in names}
            for k in names:
                sd[k].requires_grad_(False)
            fpd_ref.adam_update(sd, g, adam, 1e-3)
            if it % 20 == 0:
                print('train step %d loss %.5f' % (it, float(loss)), flush=True)
    else:
        fpd_ref.calibrate_bn(sd, S, [b[0] for b in xs[:2]])
    x = xs[2][0]
    sd64 = {k: (v.double() if v.is_floating_point() else v) for k, v in sd.items()}
    with torch.no_grad():
        truth = forward(sd64, x.double(), S, ident, ident, ident)
        variants = {
            'ref32': forward(sd, x, S, ident, ident, ident),
            'bf16_all': forward(sd, x, S, bf, bf, bf),
            'bf16_int': forward(sd, x, S, bf, ident, bf),
            'bf16_w': forward(sd, x, S, ident, ident, bf),
        }
        with torch.autocast('cpu', dtype=torch.bfloat16):
            variants['autocast'] = [o.float() for o in H.hourglass_forward(sd, x, S, train=False)]
    print('net hg S=%d F=%d, %dx%d, batch %d, %s; |map| rms per stack: %s' % (
        S, args.feats, args.size, args.size, args.batch, 'trained %d steps' % args.train_steps if args.train_steps else 'random init, calibrated BN',
        ['%.3f' % float(t.pow(2).mean().sqrt()) for t in truth]))
    for name, outs in variants.items():
        print('%-9s rel-L2 vs fp64 per stack: %s' % (name, ' '.join('%.2e' % rel(o, t) for o, t in zip(outs, truth))))


if __name__ == '__main__':
    main()
