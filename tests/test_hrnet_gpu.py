"""HRNet product path on the MI355X (SURVEY.md section 8(a) rows PoseHighResolutionNet.forward, BasicBlock, Bottleneck,
HighResolutionModule + fuse layers, transition layers; /root/reference/lib/models/pose_hrnet.py:28-98,187-265,333-372,425-460):
new ops against their CPU specification (oracle/plan_interp.py), strided convolutions and their data / weight gradients,
and one whole FPD iteration of the scaled-down W-net pair against the golden vectors the reference's own pose_hrnet.py +
loss.py produced (tests/golden/hrnet_tiny.npz), through models.pose_hrnet.get_pose_net and executor.FusedFPDStep."""
import os

import numpy as np
import pytest
import torch

from oracle import fpd_ref, hrnet_ref, plan_interp as PI
from tests import _cases
from tests._cases_hrnet import CONFIG, extra_cfg
from tests.test_kernels_gpu import RS, TOL, Bench, make_bn, rnd, tensor_stats

pytestmark = pytest.mark.gpu
G = None
GOLD = os.path.join(os.path.dirname(__file__), 'golden', 'hrnet_tiny.npz')


def setup_module(module):
    from tests import test_kernels_gpu as T
    T.setup_module(T)
    global G
    G = T.G


def ew(name, dims, **kw):
    f = dict(x=None, x2=None, dy=None, add=None, y=None, out_stats=None, bstats=None, dgamma=None, dbeta=None, bn=None)
    f.update(kw)
    return G.Op('ew', op=name, dims=tuple(dims), **f)


@pytest.mark.parametrize('dtype', [0, 1])
@pytest.mark.parametrize('case', [(2, 16, 12, 32), (3, 8, 24, 64), (1, 64, 48, 8)])
def test_affsum_relu_mask_dilate_and_stats_only_sums(case, dtype):
    """affsum (identity + train-BN x2-up + eval-BN x4-up + plain term, ReLU), relu_mask, dilate2, BN sums without a store."""
    N, H, W, C = case
    gen = torch.Generator().manual_seed(11 + sum(case))
    bt = Bench(dtype)
    x0 = bt.act((N, H, W, C), rnd(gen, N, H, W, C), 'x0')
    x1v = rnd(gen, N, H // 2, W // 2, C) + 0.2
    x1 = bt.act((N, H // 2, W // 2, C), x1v, 'x1')
    x2 = bt.act((N, H // 4, W // 4, C), rnd(gen, N, H // 4, W // 4, C), 'x2')
    x3 = bt.act((N, H, W, C), rnd(gen, N, H, W, C), 'x3')
    bn1 = make_bn(bt, gen, C, 'train', 'b1', relu=False)
    bn1.count = N * (H // 2) * (W // 2)
    bn1.stats = bt.buf('stats', (RS, 2, C), tensor_stats(x1v.to(torch.bfloat16).float() if dtype == 1 else x1v))
    bn2 = make_bn(bt, gen, C, 'eval', 'b2', relu=False)
    bn3 = make_bn(bt, gen, C, 'eval', 'b3', relu=True)
    y = bt.act((N, H, W, C), None, 'y')
    dy = bt.act((N, H, W, C), rnd(gen, N, H, W, C), 'dy')
    g = bt.act((N, H, W, C), None, 'g')
    dil = bt.act((N, 2 * H, 2 * W, C), None, 'dil')
    bst = bt.buf('stats', (RS, 2, C), torch.zeros(RS, 2, C, dtype=torch.float64))
    # block-tail backward in one pass: g = dy * (mask source > 0) written AND the BN-backward sums of term x1 (x2 = mask source)
    mk = bt.act((N, H // 2, W // 2, C), rnd(gen, N, H // 2, W // 2, C), 'mk')
    dy2 = bt.act((N, H // 2, W // 2, C), rnd(gen, N, H // 2, W // 2, C), 'dy2')
    g2 = bt.act((N, H // 2, W // 2, C), None, 'g2')
    bst2 = bt.buf('stats', (RS, 2, C), torch.zeros(RS, 2, C, dtype=torch.float64))
    ops = [G.Op('affsum', terms=[(x0, None, 1), (x1, bn1, 2), (x2, bn2, 4), (x3, bn3, 1)], y=y, relu=True, out_stats=None,
                dims=(N, H, W, C), extra_in=[x0, x1, x2, x3]),
           ew('relu_mask', (N, H, W, C), x=y, dy=dy, y=g),
           ew('dilate2', (N, 2 * H, 2 * W, C), x=g, y=dil),
           ew('bnrelu_bwd_r', (N, H // 2, W // 2, C), x=x1, dy=x1, y=None, bstats=bst, bn=bn1),
           ew('bnrelu_bwd_r', (N, H // 2, W // 2, C), x=x1, x2=mk, dy=dy2, y=g2, bstats=bst2, bn=bn1)]
    bt.realise().run(ops, 0)
    bt.compare(y, label='affsum %s' % (case,), **TOL[dtype])
    bt.compare(g, label='relu_mask', **TOL[dtype])
    bt.compare(dil, label='dilate2', **TOL[dtype])
    bt.compare(bst, atol=TOL[dtype]['atol'] * bn1.count, rtol=TOL[dtype]['rtol'], label='bn sums (no store)')
    bt.compare(g2, label='masked gradient (mask source x2)', **TOL[dtype])
    bt.compare(bst2, atol=TOL[dtype]['atol'] * bn1.count, rtol=TOL[dtype]['rtol'], label='bn sums of the masked gradient')
    exp = torch.where(bt.cpu.view(mk.buf).float() > 0, bt.cpu.view(dy2.buf).float(), torch.zeros(1))
    assert torch.equal(bt.gpu.view(g2.buf).float().cpu(), exp.to(bt.cpu.view(g2.buf).dtype).float())      # a mask is exact


S2 = [(2, 16, 12, 32, 64), (2, 32, 24, 16, 16), (1, 64, 48, 64, 128), (3, 8, 6, 8, 8), (2, 64, 64, 3, 64)]


@pytest.mark.parametrize('backend', [0, 1, 2])
@pytest.mark.parametrize('dtype', [0, 1])
@pytest.mark.parametrize('case', S2)
def test_stride2_conv_forward_dgrad_wgrad(case, dtype, backend):
    """3x3 stride-2 pad-1 convolution (pose_hrnet.py:223-242,279-284,349-372): forward with a folded train-mode BN+ReLU and
    output statistics, its data gradient as the stride-1 convolution of the zero-dilated output gradient with the flipped
    weights, and its weight gradient -- each vs the CPU specification; the dilated formulation itself is checked against
    torch autograd of F.conv2d(stride=2)."""
    N, H, W, C, K = case
    gen = torch.Generator().manual_seed(5 + sum(case))
    bt = Bench(dtype)
    P, Q = H // 2, W // 2
    x_val = rnd(gen, N, H, W, C) + 0.3
    x = bt.act((N, H, W, C), x_val, 'x')
    wv = rnd(gen, K, 3, 3, C, scale=1.0 / np.sqrt(9 * C))
    wm = bt.buf('param', (K, 3, 3, C), wv)
    wf = bt.buf('wlp', (K, 3, 3, C))
    wb = bt.buf('wlp', (C, 3, 3, K))
    y = bt.act((N, P, Q, K), None, 'y')
    ost = bt.buf('stats', (RS, 2, K), torch.zeros(RS, 2, K, dtype=torch.float64))
    use_bn = C % 8 == 0
    bn = None
    if use_bn:
        bn = make_bn(bt, gen, C, 'train')
        bn.count = N * H * W
        bn.stats = bt.buf('stats', (RS, 2, C), tensor_stats(x_val.to(torch.bfloat16).float() if dtype == 1 else x_val))
    dyv = rnd(gen, N, P, Q, K)
    dy = bt.act((N, P, Q, K), dyv, 'dy')
    dil = bt.act((N, H, W, K), None, 'dil')
    dx = bt.act((N, H, W, C), None, 'dx')
    dw = bt.buf('grad', (K, 3, 3, C), torch.zeros(K, 3, 3, C))
    fwd = G.Op('conv', x=x, w=wf, wkey='w', bias=None, bkey=None, residual=None, y=y, out_stats=ost, bn=bn, epi='plain',
               epi_x=None, epi_bn=None, epi_stats=None, dims=(N, H, W, C, K, 3, 3, 2, 1, P, Q))
    ops = [G.Op('wprep', entries=[{'w': wm, 'w_fwd': wf, 'w_bwd': wb}]), fwd,
           ew('dilate2', (N, H, W, K), x=dy, y=dil),
           G.Op('conv', x=dil, w=wb, wkey='w', bias=None, bkey=None, residual=None, y=dx, out_stats=None, bn=None, epi='plain',
                epi_x=None, epi_bn=None, epi_stats=None, dims=(N, H, W, K, C, 3, 3, 1, 1, H, W)),
           G.Op('wgrad', x=x, dy=dy, dw=dw, dbias=None, bn=bn, dims=(N, H, W, C, K, 3, 3, 2, 1, P, Q))]
    bt.realise().run(ops, backend)
    bt.compare(y, label='s2 conv y %s' % (case,), **TOL[dtype])
    bt.compare(ost, atol=TOL[dtype]['atol'] * N * P * Q, rtol=TOL[dtype]['rtol'], label='s2 conv stats')
    bt.compare(dx, label='s2 dgrad', **TOL[dtype])
    m = N * P * Q
    tol = dict(atol=2e-4 + 1e-6 * m, rtol=2e-4) if dtype == 0 else dict(atol=2e-2 + 2e-5 * m, rtol=3e-2)
    bt.compare(dw, label='s2 wgrad', **tol)
    if dtype == 0 and backend == 0 and not use_bn:             # the specification itself vs torch autograd (once per shape class)
        xt = x_val.permute(0, 3, 1, 2).clone().requires_grad_(True)
        wt = wv.permute(0, 3, 1, 2).clone().requires_grad_(True)
        out = torch.nn.functional.conv2d(xt, wt, None, stride=2, padding=1)
        out.backward(dyv.permute(0, 3, 1, 2))
        torch.testing.assert_close(bt.cpu.view(dx.buf), xt.grad.permute(0, 2, 3, 1), atol=1e-4, rtol=1e-4)
        torch.testing.assert_close(bt.cpu.view(dw), wt.grad.permute(0, 2, 3, 1), atol=1e-3, rtol=1e-4)


class AD(dict):
    __getattr__ = dict.__getitem__


def hrnet_cfg(which, dtype='fp32'):
    from fpd_amd.lib.config import _wrap
    c = CONFIG
    extra = dict(extra_cfg(c[which]), PRETRAINED_LAYERS=['*'])
    return _wrap({'MODEL': {'NAME': 'pose_hrnet', 'NUM_JOINTS': c['joints'], 'INIT_WEIGHTS': False, 'PRETRAINED': '', 'DTYPE': dtype,
                            'EXTRA': extra}})


def build_pair(dtype='fp32'):
    from fpd_amd.lib import models
    c = CONFIG
    student = eval('models.pose_hrnet.get_pose_net')(hrnet_cfg('s', dtype), is_train=True)
    teacher = eval('models.pose_hrnet.get_pose_net')(hrnet_cfg('t', dtype), is_train=False)
    s_sd = fpd_ref.synth_state_dict(hrnet_ref.hrnet_keys(extra_cfg(c['s']), c['joints']), 1)
    t_sd = fpd_ref.synth_state_dict(hrnet_ref.hrnet_keys(extra_cfg(c['t']), c['joints']), 2)
    student.load_state_dict(s_sd, strict=True)
    teacher.load_state_dict(t_sd, strict=True)
    return student.cuda(), teacher.cuda(), s_sd, t_sd


def test_hrnet_fused_fpd_step_matches_the_reference_goldens():
    """One FPD iteration (function.py:119-147, single-tensor branch) of the small W-net pair, fp32 parity build: teacher map,
    student map, pose / KD / total loss, student gradients and BN running statistics against what the reference's own
    PoseHighResolutionNet + JointsMSELoss produced.  Maps: |ours - ref| <= 1e-4 above the fp32 noise floor (fp64 referee);
    gradients: relative L2 vs the fp64 oracle (one ReLU-kink flip moves them ~1e-3, see tests/test_hrnet_graph_cpu.py)."""
    from fpd_amd import executor as E
    from tests.test_hrnet_graph_cpu import truth64
    c = CONFIG
    gold = np.load(GOLD)
    student, teacher, s_sd, t_sd = build_pair()
    inp, tg, tw = fpd_ref.synth_batch(100, c['batch'], c['joints'], c['image'], c['heat'])
    W, H = c['image']
    step = E.FusedFPDStep(student.device_state(), student.cfg_hg, teacher.device_state(), teacher.cfg_hg, c['batch'], H, W,
                          alpha=c['alpha'])
    step.set_batch(inp, tg, tw)
    step.teacher_async(inp)
    s = step.student
    torch.cuda.current_stream().wait_event(step.ev_t[0])
    s.run('prep'); s.run('fwd'); s.run('mid'); s.run('bwd')
    torch.cuda.synchronize()
    hw, hh = c['heat']
    tmap = step.tmap[0].view(c['batch'], hh, hw, c['joints']).permute(0, 3, 1, 2).float().cpu().numpy()
    out = s.output_view(0).permute(0, 3, 1, 2).float().cpu().numpy()
    t_out, t_grad = truth64()
    assert np.abs(tmap - gold['toutput']).max() < 1e-4
    _cases.assert_parity(out, gold['output'], t_out.numpy(), 'student map', floor=1e-4)
    pose, kd, loss = step.losses()
    assert abs(pose - float(gold['pose'])) < 2e-5 and abs(kd - float(gold['kd'])) < 2e-5 and abs(loss - float(gold['loss'])) < 2e-5
    student._attach_grads()
    flat = torch.cat([p.grad.reshape(-1) for p in student.parameters()]).double().cpu()
    rel = float((flat - t_grad).norm() / t_grad.norm())
    assert rel < 5e-3, rel
    assert abs(float(flat.norm()) - float(gold['grad_norm'])) < 3e-3 * float(gold['grad_norm'])
    sd = student.state_dict()
    for k in gold.files:
        if k.startswith('s_after/'):
            assert np.abs(sd[k[len('s_after/'):]].float().cpu().numpy() - gold[k]).max() < 5e-6, k
    print('hrnet tiny: map err %.2e, loss %.6f (ref %.6f), grad rel-L2 %.2e' % (np.abs(out - gold['output']).max(), loss,
                                                                             float(gold['loss']), rel))


def test_hrnet_module_api_single_tensor_and_autograd():
    """models.pose_hrnet through the reference's calling convention (function.py:119-146): model(x) returns ONE tensor,
    JointsMSELoss objects, loss.backward() fills .grad -- equal to the fused step's numbers."""
    from fpd_amd.lib.core.loss import JointsMSELoss
    c = CONFIG
    gold = np.load(GOLD)
    student, teacher, _, _ = build_pair()
    inp, tg, tw = fpd_ref.synth_batch(100, c['batch'], c['joints'], c['image'], c['heat'])
    x, tg, tw = inp.cuda(), tg.cuda(), tw.cuda()
    student.train(); teacher.eval()
    output = student(x)
    with torch.no_grad():
        toutput = teacher(x)
    assert isinstance(output, torch.Tensor) and isinstance(toutput, torch.Tensor) and tuple(output.shape) == gold['output'].shape
    crit = JointsMSELoss(True).cuda()
    pose, kd = crit(output, tg, tw), crit(output, toutput, tw)
    loss = (1 - c['alpha']) * pose + c['alpha'] * kd
    loss.backward()
    assert abs(loss.item() - float(gold['loss'])) < 2e-5
    flat = torch.cat([p.grad.reshape(-1) for p in student.parameters()]).cpu()
    stride = int(gold['grad_stride'])
    ref = torch.from_numpy(gold['grad_flat'])
    assert float((flat[::stride] - ref).norm() / ref.norm()) < 5e-3


def test_hrnet_w32_w48_bf16_step_runs_at_coco_shape():
    """BASELINE configs[3] shapes (W32 <- W48, 256x192, J=17) in the bf16 build at a small batch: finite losses equal to the
    fp32 build's to 1e-3 on the same weights, the Adam step decreases the loss, and the bf16 student map within the
    random-init amplification band of the fp32 one (measured 0.19 relative L2 -- bf16 weight rounding alone moves such
    untrained networks by 10-30 %, see tests/test_bf16_parity_gpu.py)."""
    from fpd_amd import executor as E
    from fpd_amd.lib import models
    from fpd_amd.lib.config import _wrap
    B, J, H, W = 4, 17, 256, 192

    def cfg(widths, dtype):
        extra = dict(extra_cfg(dict(widths=widths, blocks=4, modules=(1, 4, 3))), PRETRAINED_LAYERS=['*'])
        return _wrap({'MODEL': {'NAME': 'pose_hrnet', 'NUM_JOINTS': J, 'INIT_WEIGHTS': False, 'PRETRAINED': '', 'DTYPE': dtype, 'EXTRA': extra}})
    x, tg, tw = fpd_ref.synth_batch(7, B, J, (W, H), (W // 4, H // 4))
    maps, losses = {}, {}
    for dtype in ('fp32', 'bf16'):
        torch.manual_seed(1)
        s = models.pose_hrnet.get_pose_net(cfg([32, 64, 128, 256], dtype), is_train=True).cuda()
        torch.manual_seed(2)
        t = models.pose_hrnet.get_pose_net(cfg([48, 96, 192, 384], dtype), is_train=False).cuda()
        step = E.FusedFPDStep(s.device_state(), s.cfg_hg, t.device_state(), t.cfg_hg, B, H, W, alpha=0.5, lr=1e-3)
        step.set_batch(x, tg, tw)
        tr = []
        for it in range(4 if dtype == 'bf16' else 1):
            step.step()
            tr.append(step.losses()[2])
            if it == 0:
                maps[dtype] = step.student.output_view(0).float().cpu().clone()
        losses[dtype] = tr
        del step, s, t
        torch.cuda.empty_cache()
    assert all(np.isfinite(v) for v in losses['bf16']) and losses['bf16'][-1] < losses['bf16'][0], losses
    rel = float((maps['bf16'] - maps['fp32']).norm() / maps['fp32'].norm())
    assert rel < 0.4 and abs(losses['bf16'][0] - losses['fp32'][0]) < 1e-3 * losses['fp32'][0], (rel, losses)
    print('hrnet W32<-W48 bf16 vs fp32: map rel-L2 %.3e, loss %.5f vs %.5f' % (rel, losses['bf16'][0], losses['fp32'][0]))


def test_hrnet_w32_w48_full_shape_fp32_vs_oracle_on_cuda():
    """W32 <- W48 at 256x192, J=17, B=8, fp32 parity build against the oracle restatement (oracle/hrnet_ref.py, bit-identical
    to the reference module on CPU) executed on CUDA tensors -- torch fp32 / MIOpen, an implementation independent of this
    package's kernels; referee for the frozen teacher: the oracle in fp64 on two samples.  Criterion as in
    tests/test_fullsize_gpu.py: never less accurate than 1.5x the torch fp32 evaluation where a referee exists, losses to
    1e-5, whole-vector gradient relative L2 between the two fp32 evaluations < 2e-2."""
    from fpd_amd import executor as E
    from fpd_amd.lib import models
    from fpd_amd.lib.config import _wrap
    dev = torch.device('cuda', 0)
    B, J, H, W = 8, 17, 256, 192
    ex_s = extra_cfg(dict(widths=[32, 64, 128, 256], blocks=4, modules=(1, 4, 3)))
    ex_t = extra_cfg(dict(widths=[48, 96, 192, 384], blocks=4, modules=(1, 4, 3)))

    def cfg(ex):
        return _wrap({'MODEL': {'NAME': 'pose_hrnet', 'NUM_JOINTS': J, 'INIT_WEIGHTS': False, 'PRETRAINED': '', 'DTYPE': 'fp32',
                                'EXTRA': dict(ex, PRETRAINED_LAYERS=['*'])}})
    s_sd = fpd_ref.synth_state_dict(hrnet_ref.hrnet_keys(ex_s, J), 1)
    t_sd = fpd_ref.synth_state_dict(hrnet_ref.hrnet_keys(ex_t, J), 2)
    x, tg, tw = fpd_ref.synth_batch(100, B, J, (W, H), (W // 4, H // 4))
    student = models.pose_hrnet.get_pose_net(cfg(ex_s), is_train=True)
    teacher = models.pose_hrnet.get_pose_net(cfg(ex_t), is_train=False)
    student.load_state_dict(s_sd, strict=True)
    teacher.load_state_dict(t_sd, strict=True)
    student, teacher = student.to(dev), teacher.to(dev)
    step = E.FusedFPDStep(student.device_state(), student.cfg_hg, teacher.device_state(), teacher.cfg_hg, B, H, W, alpha=0.5)
    step.set_batch(x, tg, tw)
    step.teacher_async(x)
    s = step.student
    torch.cuda.current_stream().wait_event(step.ev_t[0])
    torch.cuda.synchronize()
    ours_t = step.tmap[0].view(B, H // 4, W // 4, J).permute(0, 3, 1, 2).float().cpu()
    # checker: oracle on CUDA
    t_cu = {k: v.to(dev) for k, v in t_sd.items()}
    with torch.no_grad():
        m_t = hrnet_ref.hrnet_forward(t_cu, ex_t, x.to(dev), train=False)
    step.tmap[0].copy_(m_t.permute(0, 2, 3, 1).reshape(-1))          # both student steps distil from the same map
    s.run('prep'); s.run('fwd'); s.run('mid'); s.run('bwd')
    torch.cuda.synchronize()
    ours_m = s.output_view(0).permute(0, 3, 1, 2).float().cpu()
    pose, kd, loss = step.losses()
    student._attach_grads()
    ours_g = torch.cat([p.grad.reshape(-1) for p in student.parameters()]).double().cpu()
    s_cu = {k: v.to(dev) for k, v in s_sd.items()}
    names = [k for k in s_cu if s_cu[k].is_floating_point() and 'running' not in k]
    for k in names:
        s_cu[k].requires_grad_(True)
    m_m = hrnet_ref.hrnet_forward(s_cu, ex_s, x.to(dev), train=True)
    mp, mk, ml = fpd_ref.fpd_losses([m_m], m_t, tg.to(dev), tw.to(dev), 0.5)
    ml.backward()
    m_g = torch.cat([s_cu[k].grad.reshape(-1) for k in names]).double().cpu()
    t64 = {k: (v.double() if v.is_floating_point() else v) for k, v in t_sd.items()}
    with torch.no_grad():
        t_t = hrnet_ref.hrnet_forward(t64, ex_t, x[:2].double(), train=False)

    def err(a, b):
        return float((a.double().cpu() - b.double().cpu()).abs().max())
    e_o, e_m = err(ours_t[:2], t_t), err(m_t[:2], t_t)
    assert e_o <= max(1e-4, 1.5 * e_m), ('teacher map', e_o, e_m)
    assert err(ours_m, m_m.detach()) < 1e-3 * max(1.0, float(m_m.abs().max())), ('student map', err(ours_m, m_m.detach()))
    assert abs(pose - float(mp)) < 1e-5 * max(1, abs(float(mp))) and abs(kd - float(mk)) < 1e-5 * max(1, abs(float(mk)))
    rel = float((ours_g - m_g).norm() / m_g.norm())
    assert rel < 2e-2, rel
    print('hrnet W32<-W48 fp32 vs MIOpen: teacher |.-fp64| %.2e (MIOpen %.2e), student map %.2e, loss %.6f vs %.6f, grads rel-L2 %.2e'
          % (e_o, e_m, err(ours_m, m_m.detach()), loss, float(ml), rel))
