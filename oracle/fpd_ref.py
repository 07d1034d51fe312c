"""CPU restatement of the FPD loss and train-step (test infrastructure).

Follows, without copying:
  /root/reference/lib/core/loss.py:21-39      JointsMSELoss.forward
  /root/reference/lib/core/function.py:107-147 fpd_train loop body
  /root/reference/lib/utils/utils.py:69-73     Adam(lr) only (no weight decay)
  /root/reference/lib/dataset/JointsDataset.py:233-289 generate_target (gaussian)
"""
import zlib

import numpy as np
import torch

from . import hourglass_ref


def joints_mse_loss(output, target, target_weight, use_target_weight=True):
    """loss.py:21-39: sum_j 0.5*mean_{b,hw}((p_j*w_j - g_j*w_j)^2) / J.

    Written per joint like the reference (not the closed form) so that the
    fp32 summation order of nn.MSELoss(mean) per joint is retained."""
    b, j = output.shape[0], output.shape[1]
    p = output.reshape(b, j, -1)
    g = target.reshape(b, j, -1)
    total = 0
    for k in range(j):
        pk, gk = p[:, k, :], g[:, k, :]
        if use_target_weight:
            w = target_weight[:, k]            # [B,1]
            pk, gk = pk * w, gk * w
        total = total + 0.5 * torch.mean((pk - gk) ** 2)
    return total / j


def joints_mse_closed_form(output, target, target_weight):
    """0.5/(B*J*HW) * sum w^2 (p-g)^2 in fp64 -- the quantity the HIP loss kernel computes."""
    b, j = output.shape[0], output.shape[1]
    w = target_weight.reshape(b, j, 1).double()
    d = (output.reshape(b, j, -1).double() - target.reshape(b, j, -1).double()) * w
    return 0.5 * (d * d).sum() / d.numel()


def fpd_losses(outputs, toutput, target, target_weight, alpha):
    """function.py:121-140.  outputs: list (hourglass) or tensor (HRNet); toutput: last teacher map."""
    if isinstance(toutput, (list, tuple)):
        toutput = toutput[-1]
    if not isinstance(outputs, (list, tuple)):
        outputs = [outputs]
    pose = joints_mse_loss(outputs[0], target, target_weight)
    kd = joints_mse_loss(outputs[0], toutput, target_weight)
    for o in outputs[1:]:
        pose = pose + joints_mse_loss(o, target, target_weight)
        kd = kd + joints_mse_loss(o, toutput, target_weight)
    loss = (1 - alpha) * pose + alpha * kd
    return pose, kd, loss


def param_names(sd):
    return [k for k in sd if not (k.endswith('running_mean') or k.endswith('running_var')
                                  or k.endswith('num_batches_tracked'))]


def fpd_step(student_sd, teacher_sd, s_stacks, t_stacks, inp, target, target_weight, alpha,
             adam_state=None, lr=2.5e-4, teacher_no_grad=True, student_forward=None, teacher_forward=None):
    """One iteration of function.py:114-147 on CPU.  Mutates student_sd (Adam update and BN
    running stats) in place.  Returns dict(outputs, toutput, pose, kd, loss, grads).

    teacher_no_grad=False reproduces the reference faithfully (the teacher forward is NOT under
    no_grad, function.py:120, so autograd also back-propagates through the teacher); the student
    gradients are identical either way.

    student_forward / teacher_forward: optional callables (sd, x, train) -> [maps] replacing the hourglass forward (the
    same loop drives any pose net of the reference: bench.py passes the HRNet restatement for its configs[3] line)."""
    fwd_s = student_forward or (lambda sd, x, train: hourglass_ref.hourglass_forward(sd, x, s_stacks, train=train))
    fwd_t = teacher_forward or (lambda sd, x, train: hourglass_ref.hourglass_forward(sd, x, t_stacks, train=train))
    names = param_names(student_sd)
    for k in names:
        student_sd[k].requires_grad_(True)
        student_sd[k].grad = None
    outputs = fwd_s(student_sd, inp, True)
    if teacher_no_grad:
        with torch.no_grad():
            toutput = fwd_t(teacher_sd, inp, False)[-1]
    else:
        tn = param_names(teacher_sd)
        for k in tn:
            teacher_sd[k].requires_grad_(True)
        toutput = fwd_t(teacher_sd, inp, False)[-1]
    pose, kd, loss = fpd_losses(outputs, toutput, target, target_weight, alpha)
    loss.backward()
    grads = {k: student_sd[k].grad.detach().clone() for k in names}
    if adam_state is not None:
        adam_update(student_sd, grads, adam_state, lr)
    for k in names:
        student_sd[k].requires_grad_(False)
    if not teacher_no_grad:
        for k in param_names(teacher_sd):
            teacher_sd[k].requires_grad_(False)
            teacher_sd[k].grad = None
    return dict(outputs=[o.detach() for o in outputs], toutput=toutput.detach(),
                pose=pose.detach(), kd=kd.detach(), loss=loss.detach(), grads=grads)


def adam_update(sd, grads, state, lr, b1=0.9, b2=0.999, eps=1e-8):
    """torch.optim.Adam defaults (utils.py:69-73): no weight decay, no amsgrad."""
    state['step'] = state.get('step', 0) + 1
    t = state['step']
    with torch.no_grad():
        for k, g in grads.items():
            m = state.setdefault('m.' + k, torch.zeros_like(g))
            v = state.setdefault('v.' + k, torch.zeros_like(g))
            m.mul_(b1).add_(g, alpha=1 - b1)
            v.mul_(b2).addcmul_(g, g, value=1 - b2)
            bc1, bc2 = 1 - b1 ** t, 1 - b2 ** t
            denom = (v.sqrt() / (bc2 ** 0.5)).add_(eps)
            sd[k].addcdiv_(m, denom, value=-lr / bc1)


# ----------------------------------------------------------------------------------------------
# synthetic data; numpy RNG so it is identical on every machine
# ----------------------------------------------------------------------------------------------
def generate_target(joints_xy, vis, image_size, heatmap_size, sigma):
    """JointsDataset.py:233-289 for one sample.  joints_xy [J,2] in image pixels, vis [J] in {0,1}.
    image_size/heatmap_size are (W,H).  Returns target [J,h,w] f32, target_weight [J,1] f32."""
    nj = joints_xy.shape[0]
    wh, hh = heatmap_size
    tw = np.ones((nj, 1), np.float32)
    tw[:, 0] = vis
    target = np.zeros((nj, hh, wh), np.float32)
    rad = sigma * 3
    size = 2 * rad + 1
    ax = np.arange(0, size, 1, np.float32)
    gauss = np.exp(-((ax[None, :] - size // 2) ** 2 + (ax[:, None] - size // 2) ** 2) / (2 * sigma ** 2))
    sx, sy = image_size[0] / wh, image_size[1] / hh
    for j in range(nj):
        mx = int(joints_xy[j, 0] / sx + 0.5)
        my = int(joints_xy[j, 1] / sy + 0.5)
        x0, y0 = int(mx - rad), int(my - rad)
        x1, y1 = int(mx + rad + 1), int(my + rad + 1)
        if x0 >= wh or y0 >= hh or x1 < 0 or y1 < 0:
            tw[j] = 0
            continue
        if tw[j] > 0.5:
            gx0, gx1 = max(0, -x0), min(x1, wh) - x0
            gy0, gy1 = max(0, -y0), min(y1, hh) - y0
            ix0, ix1 = max(0, x0), min(x1, wh)
            iy0, iy1 = max(0, y0), min(y1, hh)
            target[j, iy0:iy1, ix0:ix1] = gauss[gy0:gy1, gx0:gx1]
    return target, tw


def synth_batch(seed, batch, num_joints, image_size=(256, 256), heatmap_size=(64, 64), sigma=2,
                p_vis=0.85):
    """Seeded synthetic MPII-shaped batch: (input [B,3,H,W], target [B,J,h,w], target_weight [B,J,1])."""
    rng = np.random.RandomState(seed)
    w, h = image_size
    inp = rng.standard_normal((batch, 3, h, w)).astype(np.float32)
    tg = np.zeros((batch, num_joints, heatmap_size[1], heatmap_size[0]), np.float32)
    tw = np.zeros((batch, num_joints, 1), np.float32)
    for b in range(batch):
        xy = np.stack([rng.uniform(0, w, num_joints), rng.uniform(0, h, num_joints)], 1)
        vis = (rng.uniform(0, 1, num_joints) < p_vis).astype(np.float32)
        tg[b], tw[b] = generate_target(xy, vis, image_size, heatmap_size, sigma)
    return torch.from_numpy(inp), torch.from_numpy(tg), torch.from_numpy(tw)


def blob_batch(seed, batch, num_joints, image_size=(256, 256), heatmap_size=(64, 64), sigma=2, p_vis=0.85, noise=0.3):
    """Seeded synthetic batch with LEARNABLE structure (the random crops of synth_batch carry no information about their
    joints, so a network trained on them can only learn the mean map): every visible joint is drawn into the image as a
    Gaussian blob of 4*sigma pixels in a joint-specific colour (16 colours: 8 hues x 2 signs of the third channel), on
    weak noise.  Targets / weights exactly as synth_batch (JointsDataset.generate_target).  Used by the trained-pair
    fixture (tests/golden/make_golden_trained.py) and the convergence test."""
    rng = np.random.RandomState(seed)
    w, h = image_size
    inp = (noise * rng.standard_normal((batch, 3, h, w))).astype(np.float32)
    tg = np.zeros((batch, num_joints, heatmap_size[1], heatmap_size[0]), np.float32)
    tw = np.zeros((batch, num_joints, 1), np.float32)
    ys, xs = np.mgrid[0:h, 0:w].astype(np.float32)
    s_img = 4.0 * sigma
    for b in range(batch):
        xy = np.stack([rng.uniform(0.1 * w, 0.9 * w, num_joints), rng.uniform(0.1 * h, 0.9 * h, num_joints)], 1)
        vis = (rng.uniform(0, 1, num_joints) < p_vis).astype(np.float32)
        tg[b], tw[b] = generate_target(xy, vis, image_size, heatmap_size, sigma)
        for j in range(num_joints):
            if tw[b, j, 0] < 0.5:
                continue
            th = 2.0 * np.pi * (j % 8) / 8.0
            col = np.array([np.cos(th), np.sin(th), 1.0 if j < 8 else -1.0], np.float32) * 2.0
            g = np.exp(-((xs - xy[j, 0]) ** 2 + (ys - xy[j, 1]) ** 2) / (2.0 * s_img ** 2))
            inp[b] += col[:, None, None] * g[None]
    return torch.from_numpy(inp), torch.from_numpy(tg), torch.from_numpy(tw)


def synth_state_dict(keys, seed):
    """Deterministic synthetic checkpoint for a list of (key, shape): conv weights/biases
    uniform(+-1/sqrt(fan_in)) (the scale of torch's default init), BN affine near (1, 0) with
    spread so gamma/beta matter, running stats near (0, 1).  One numpy stream per key so the
    values do not depend on key order."""
    sd = {}
    shapes = dict(keys)
    for k, shp in keys:
        rng = np.random.RandomState((zlib.crc32(k.encode()) ^ (seed * 2654435761)) & 0x7FFFFFFF)
        if k.endswith('num_batches_tracked'):
            sd[k] = torch.zeros((), dtype=torch.long)
        elif k.endswith('running_mean'):
            sd[k] = torch.from_numpy((0.05 * rng.standard_normal(shp)).astype(np.float32))
        elif k.endswith('running_var'):
            sd[k] = torch.from_numpy((1.0 + 0.2 * rng.uniform(-1, 1, shp)).astype(np.float32))
        elif len(shp) == 4:
            bound = 1.0 / np.sqrt(shp[1] * shp[2] * shp[3])
            sd[k] = torch.from_numpy(rng.uniform(-bound, bound, shp).astype(np.float32))
        else:
            base = k[:-len('.weight')] if k.endswith('.weight') else k[:-len('.bias')]
            if (base + '.running_mean') in shapes:          # BN affine
                if k.endswith('.weight'):
                    sd[k] = torch.from_numpy((1.0 + 0.1 * rng.standard_normal(shp)).astype(np.float32))
                else:
                    sd[k] = torch.from_numpy((0.1 * rng.standard_normal(shp)).astype(np.float32))
            else:                                           # conv bias
                wshape = shapes[base + '.weight']
                bound = 1.0 / np.sqrt(wshape[1] * wshape[2] * wshape[3])
                sd[k] = torch.from_numpy(rng.uniform(-bound, bound, shp).astype(np.float32))
    return sd


def calibrate_bn(sd, num_stacks, batches):
    """Give a random-init net sane running stats: cumulative-average BN statistics over
    `batches` (SURVEY.md section 8(d) 'teacher BN calibration').  Uses momentum=None semantics
    (cumulative moving average) implemented by running train-mode forwards on a copy whose
    running stats we average ourselves."""
    acc_m, acc_v = {}, {}
    for x in batches:
        tmp = {k: v.clone() for k, v in sd.items()}
        for k in tmp:
            if k.endswith('running_mean'):
                tmp[k].zero_()
            if k.endswith('running_var'):
                tmp[k].fill_(0.0)
        # momentum 0.1 from zero state => running = 0.1 * batch_stat ; recover batch_stat
        with torch.no_grad():
            hourglass_ref.hourglass_forward(tmp, x, num_stacks, train=True)
        for k in tmp:
            if k.endswith('running_mean'):
                acc_m[k] = acc_m.get(k, 0) + tmp[k] / hourglass_ref.BN_MOMENTUM
            if k.endswith('running_var'):
                acc_v[k] = acc_v.get(k, 0) + tmp[k] / hourglass_ref.BN_MOMENTUM
    n = len(batches)
    for k in acc_m:
        sd[k] = (acc_m[k] / n).contiguous()
    for k in acc_v:
        sd[k] = (acc_v[k] / n).contiguous()
    return sd
