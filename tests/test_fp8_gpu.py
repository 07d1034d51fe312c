"""fp8 forward convolutions (BASELINE configs[4]: "HRNet-W32 student fp8 weights (CDNA4 fp8 MFMA)") on the MI355X, through
the C ABI, against their specification (oracle/fp8_ref.py: OCP e4m3fn fake quantisation in torch; plan_interp runs it
for ops that carry w8).  The reference has no fp8 path, so accuracy is additionally bounded against the bf16 build.

Tolerances: weight quantisation -- exact (bytes and scales).  Convolution output -- the products of e4m3 values are exact
in fp32, so kernel and specification differ by accumulation order and the final bf16 rounding only: 6e-3 relative +
6e-3 absolute on O(1) outputs (2^-8 bf16 half-ulp + slack for sums landing next to a rounding boundary)."""
import numpy as np
import pytest
import torch

from oracle import fp8_ref, fpd_ref, plan_interp as PI
from tests._cases_hrnet import extra_cfg
from tests.test_kernels_gpu import RS, Bench, make_bn, rnd, tensor_stats

pytestmark = pytest.mark.gpu
G = R = E = None


def setup_module(module):
    from tests import test_kernels_gpu as T
    T.setup_module(T)
    global G, R, E
    G, R, E = T.G, T.R, T.E


def test_weight_quantisation_is_exact():
    gen = torch.Generator().manual_seed(3)
    ws = [rnd(gen, 64, 3, 3, 64, scale=0.05), rnd(gen, 8, 1, 1, 32, scale=2.0), torch.zeros(16, 1, 1, 32),
          rnd(gen, 256, 3, 3, 128, scale=1e-3)]
    ws[1][3] = 0
    ws[1][0, 0, 0, :4] = torch.tensor([1e-9, -3e-8, 5e2, 1.0])
    dev = torch.device('cuda:0')
    ents, keep = [], []
    for w in ws:
        wd = w.to(dev).contiguous()
        q = torch.zeros(w.numel(), dtype=torch.uint8, device=dev)
        sc = torch.zeros(w.shape[0], dtype=torch.float32, device=dev)
        e = R.WquantEntryT()
        e.w, e.w8, e.scale, e.K, e.RSC = wd.data_ptr(), q.data_ptr(), sc.data_ptr(), w.shape[0], w.numel() // w.shape[0]
        ents.append(e)
        keep.append((wd, q, sc))
    arr = (R.WquantEntryT * len(ents))(*ents)
    table = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8).to(dev)
    R.check(R.lib().fpd_weight_quant_f8(table.data_ptr(), len(ents), R.current_stream()), 'wquant')
    torch.cuda.synchronize()
    for w, (wd, q, sc) in zip(ws, keep):
        eq, es = fp8_ref.quant_weights(w)
        assert torch.equal(sc.cpu(), es)
        got = q.cpu().view(torch.float8_e4m3fn).float().reshape(w.shape)
        assert torch.equal(got, eq), (got - eq).abs().max()


F8_CASES = [
    # N, H, W, C, K, R, bn_mode, residual, stats
    (2, 16, 16, 64, 64, 3, 'train', True, True),
    (2, 32, 24, 32, 32, 3, 'train', False, True),        # HRNet-W32 full-resolution branch at a small map
    (1, 64, 48, 32, 64, 1, 'eval', False, False),
    (2, 8, 6, 256, 256, 3, None, True, True),            # lowest-resolution branch, raw (already activated) input
    (3, 12, 16, 128, 8, 1, 'train', False, True),        # narrowest output the vector epilogue takes
    (2, 64, 64, 64, 128, 1, 'eval', True, False),
    (1, 96, 72, 32, 32, 3, 'train', True, True),         # 384x288 map width
]


@pytest.mark.parametrize('case', F8_CASES)
def test_conv_forward_f8(case):
    N, H, W, C, K, Rr, bn_mode, use_res, use_stats = case
    pad = (Rr - 1) // 2
    gen = torch.Generator().manual_seed(29 + sum(v for v in case if isinstance(v, int) and not isinstance(v, bool)))
    bt = Bench(1)
    x_val = rnd(gen, N, H, W, C) + 0.3
    if bn_mode is None:
        x_val = x_val.clamp_min(0)
    x = bt.act((N, H, W, C), x_val, 'x')
    wm = bt.buf('param', (K, Rr, Rr, C), rnd(gen, K, Rr, Rr, C, scale=1.0 / np.sqrt(C * Rr * Rr)))
    w = bt.buf('wlp', (K, Rr, Rr, C))
    w8 = bt.buf('w8', (K, Rr, Rr, C))
    w8s = bt.buf('w8s', (K,))
    bias = bt.buf('param', (K,), 0.1 * rnd(gen, K))
    res = bt.act((N, H, W, K), rnd(gen, N, H, W, K), 'res') if use_res else None
    y = bt.act((N, H, W, K), None, 'y')
    bn = None
    if bn_mode:
        bn = make_bn(bt, gen, C, bn_mode)
        bn.count = N * H * W
        if bn_mode == 'train':
            bn.stats = bt.buf('stats', (RS, 2, C), tensor_stats(x_val.to(torch.bfloat16).float()))
    ostats = bt.buf('stats', (RS, 2, K), torch.zeros(RS, 2, K, dtype=torch.float64)) if use_stats else None
    dims = (N, H, W, C, K, Rr, Rr, 1, pad, H, W)
    assert G.f8_conv_domain(dims)
    op = G.Op('conv', x=x, w=w, wkey='w', bias=bias, bkey='b', residual=res, y=y, out_stats=ostats, bn=bn, epi='plain',
              epi_x=None, epi_bn=None, epi_stats=None, dims=dims, w8=w8, w8s=w8s, w_master=wm)
    bt.sizes['w8'] = (bt.sizes['w8'] + 15) // 16 * 16
    bt.realise()
    # device side: quantise the master weights, then the fp8 convolution; CPU side: the interpreter quantises on the fly
    low = E.Lowering(bt.gpu, 1)
    plan = R.Plan()
    plan.add(*low.wprep([(wm, w, None)]))
    plan.add(*low.wquant([(wm, w8, w8s)]))
    code, st = low.op(op)
    assert code == R.OP_CONV_F8
    plan.add(code, st)
    plan.run(0, len(plan))
    torch.cuda.synchronize()
    PI.run(bt.cpu, [op])
    bt.compare(y, atol=6e-3, rtol=6e-3, label='conv_f8 y %s' % (case,))
    if use_stats:
        bt.compare(ostats, atol=6e-3 * N * H * W, rtol=6e-3, label='conv_f8 out_stats')
    # and it is a different function from the bf16 convolution (the test would not notice a silent fallback otherwise)
    op16 = G.Op('conv', x=x, w=w, wkey='w', bias=bias, bkey='b', residual=res, y=y, out_stats=None, bn=bn, epi='plain',
                epi_x=None, epi_bn=None, epi_stats=None, dims=dims)
    y8 = bt.gpu.view(y.buf).float().cpu().clone()
    p2 = R.Plan()
    p2.add(*low.op(op16))
    p2.run(0, 1)
    torch.cuda.synchronize()
    y16 = bt.gpu.view(y.buf).float().cpu()
    rel = float((y8 - y16).norm() / y16.norm())
    assert 2e-3 < rel < 0.12, rel


def test_f8_outside_domain_falls_back_to_bf16_weights():
    """A stride-2 convolution handed to fpd_conv_forward_f8 runs on the bf16 kernels with c.w (same result as fpd_conv_forward)."""
    gen = torch.Generator().manual_seed(5)
    N, H, W, C, K = 2, 16, 12, 32, 64
    bt = Bench(1)
    x = bt.act((N, H, W, C), rnd(gen, N, H, W, C), 'x')
    w = bt.buf('wlp', (K, 3, 3, C), rnd(gen, K, 3, 3, C, scale=0.06))
    w8 = bt.buf('w8', (K, 3, 3, C))
    w8s = bt.buf('w8s', (K,))
    y = bt.act((N, H // 2, W // 2, K), None, 'y')
    dims = (N, H, W, C, K, 3, 3, 2, 1, H // 2, W // 2)
    assert not G.f8_conv_domain(dims)
    op = G.Op('conv', x=x, w=w, wkey='w', bias=None, bkey=None, residual=None, y=y, out_stats=None, bn=None, epi='plain',
              epi_x=None, epi_bn=None, epi_stats=None, dims=dims)
    bt.realise()
    low = E.Lowering(bt.gpu, 1)
    _, c = low.op(op)
    f = R.ConvF8T()
    f.c, f.w8, f.w8_scale = c, bt.gpu.ptr(w8), bt.gpu.ptr(w8s)
    R.check(R.lib().fpd_conv_forward_f8(f, R.current_stream()), 'conv_f8 fallback')
    torch.cuda.synchronize()
    PI.run(bt.cpu, [op])
    bt.compare(y, atol=3e-2, rtol=3e-2, label='fallback')


def _hr_cfg(widths, J, weight_dtype='', blocks=4, modules=(1, 4, 3)):
    from fpd_amd.lib.config import _wrap
    extra = dict(extra_cfg(dict(widths=widths, blocks=blocks, modules=modules)), PRETRAINED_LAYERS=['*'])
    return _wrap({'MODEL': {'NAME': 'pose_hrnet', 'NUM_JOINTS': J, 'INIT_WEIGHTS': False, 'PRETRAINED': '', 'DTYPE': 'bf16',
                            'WEIGHT_DTYPE': weight_dtype, 'EXTRA': extra}})


def test_hrnet_fp8_student_step():
    """HRNet-W32 student (fp8 forward convolutions) <- bf16 HRNet-W48 teacher at 384x288 (BASELINE configs[4] shapes, small
    batch).  Per-layer accuracy is pinned by the kernel tests above; end to end this random-init train-mode-BN network
    amplifies any operand rounding ~100x (bf16 storage alone moves its map by 10-30 %, tests/test_bf16_parity_gpu.py), so the
    whole-step assertions are the ones that survive that: >=250 convolutions actually on the fp8 pipe, finite losses within
    5 % of the bf16 build's on the same weights, and Adam steps on the fp8 path decrease the loss.  Measured: map rel-L2
    0.62 against the bf16 build and uncorrelated gradients (cosine 0.02) -- the two builds are different realisations of a
    chaotic network; the gradient of the fp8 step is pinned where it can be: against the interpreter running the same fp8
    op list (test_hrnet_fp8_train_step_matches_the_interpreter)."""
    from fpd_amd.lib import models
    B, J, H, W = 2, 17, 384, 288
    x, tg, tw = fpd_ref.synth_batch(7, B, J, (W, H), (W // 4, H // 4), sigma=3)
    maps, losses, grads, ncv = {}, {}, {}, {}
    for wd in ('', 'fp8'):
        torch.manual_seed(1)
        s = models.pose_hrnet.get_pose_net(_hr_cfg([32, 64, 128, 256], J, wd), is_train=True).cuda()
        torch.manual_seed(2)
        t = models.pose_hrnet.get_pose_net(_hr_cfg([48, 96, 192, 384], J), is_train=False).cuda()
        step = E.FusedFPDStep(s.device_state(), s.cfg_hg, t.device_state(), t.cfg_hg, B, H, W, alpha=0.5, lr=1e-3)
        ncv[wd] = sum(1 for o in step.student.g.fwd if o.kind == 'conv' and getattr(o, 'w8', None) is not None)
        step.set_batch(x, tg, tw)
        tr = []
        for it in range(4):
            step.step()
            tr.append(step.losses()[2])
            if it == 0:
                maps[wd] = step.student.output_view(0).float().cpu().clone()
                s._attach_grads()
                grads[wd] = torch.cat([p.grad.reshape(-1) for p in s.parameters()]).double().cpu()
        losses[wd] = tr
        del step, s, t
        torch.cuda.empty_cache()
    assert ncv[''] == 0 and ncv["fp8"] >= 250, ncv
    assert all(np.isfinite(v) for v in losses['fp8']) and losses['fp8'][-1] < losses['fp8'][0], losses
    rel = float((maps['fp8'] - maps['']).norm() / maps[''].norm())
    cos = float((grads['fp8'] * grads['']).sum() / (grads['fp8'].norm() * grads[''].norm()))
    print('hrnet fp8 vs bf16 student: map rel-L2 %.3e, loss %.5f vs %.5f, gradient cosine %.4f' % (rel, losses['fp8'][0], losses[''][0], cos))
    assert 1e-3 < rel < 1.5, rel
    assert abs(losses['fp8'][0] - losses[''][0]) < 5e-2 * losses[''][0], losses


def test_hrnet_fp8_eval_forward_matches_the_interpreter():
    """Whole eval-mode forward of a small HRNet with fp8 convolutions: HIP plan vs the CPU interpreter running the same op
    list with the fp8 specification (both quantise the same fp32 master weights)."""
    from fpd_amd.lib import models
    from oracle import hrnet_ref
    from tests import _interp_util as U
    J, B, H, W = 5, 2, 128, 96
    widths = [32, 64, 128, 256]
    cfg = _hr_cfg(widths, J, 'fp8', blocks=1, modules=(1, 1, 1))
    m = models.pose_hrnet.get_pose_net(cfg, is_train=False)
    ex = extra_cfg(dict(widths=widths, blocks=1, modules=(1, 1, 1)))
    keys = hrnet_ref.hrnet_keys(ex, J)
    sd = fpd_ref.synth_state_dict(keys, 4)
    m.load_state_dict(sd, strict=True)
    m = m.cuda().eval()
    x, _, _ = fpd_ref.synth_batch(11, B, J, (W, H), (W // 4, H // 4))
    with torch.no_grad():
        out = m(x.cuda()).cpu()
    inst = m.instance(x.shape, False)
    assert sum(1 for o in inst.g.fwd if o.kind == 'conv' and o.w8 is not None) > 20
    table = G.ParamTable(keys, bucket_of=G.hrnet_bucket_of)
    g = G.HRNetGraph(table, ex, J, B, H, W, False, wlp_is_master=False, fp8=True)
    A = U.make_arenas(g, table, G.plan_memory(g.fwd), act_dtype=torch.bfloat16)
    U.load_params(A, table, sd)
    A.t['image'].copy_(x.reshape(-1))
    PI.run(A, [U.wprep_op(g, table)] + g.fwd)
    ref = A.view(g.outputs[0].buf).float().permute(0, 3, 1, 2)
    rel = float((out - ref).norm() / ref.norm())
    # identical specification, different accumulation order: values next to a bf16 / e4m3 rounding boundary land on
    # either side and the network amplifies those flips (same band as the bf16 build's interpreter comparison)
    print('hrnet fp8 eval forward vs interpreter: rel-L2 %.3e' % rel)
    assert rel < 0.12, rel


def test_hrnet_fp8_train_step_matches_the_interpreter():
    """One plain training step (forward with train-mode BN, loss, backward) of a small HRNet with fp8 forward convolutions:
    HIP plan vs the CPU interpreter executing the SAME op list with the fp8 specification in the forward and the bf16 ops in
    the backward (straight-through: data / weight gradients use the bf16 weights and the stored bf16 activations; the
    backward op list is the bf16 build's, tests/test_fp8_graph_cpu.py).  With train-mode BN this random-init network turns
    the e4m3 rounding-boundary flips between two evaluation orders (each moves an operand by 6-12 %) into a different
    realisation: measured map rel-L2 0.25 and uncorrelated gradient vectors (cosine 0.08) between HIP and interpreter, where
    the eval-mode forward of the same network agrees to 1e-2.  What is asserted is therefore what a wiring error would
    still break: the loss within 5 %, the map within a factor-two band of that measurement, and the gradient norm within
    a factor of two (a dropped / doubled term or a wrong scale shows as a larger norm error or a non-finite value)."""
    from fpd_amd.lib import models
    from oracle import hrnet_ref
    from tests import _interp_util as U
    J, B, H, W = 5, 2, 128, 96
    widths = [32, 64, 128, 256]
    m = models.pose_hrnet.get_pose_net(_hr_cfg(widths, J, 'fp8', blocks=1, modules=(1, 1, 1)), is_train=True)
    ex = extra_cfg(dict(widths=widths, blocks=1, modules=(1, 1, 1)))
    keys = hrnet_ref.hrnet_keys(ex, J)
    sd = fpd_ref.synth_state_dict(keys, 4)
    m.load_state_dict(sd, strict=True)
    m = m.cuda().train()
    x, tg, tw = fpd_ref.synth_batch(11, B, J, (W, H), (W // 4, H // 4))
    step = E.FusedFPDStep(m.device_state(), m.cfg_hg, None, None, B, H, W, alpha=0.0, lr=1e-3)
    assert sum(1 for o in step.student.g.fwd if o.kind == 'conv' and o.w8 is not None) > 20
    step.set_batch(x, tg, tw)
    s = step.student
    s.run('prep'); s.run('fwd'); s.run('mid'); s.run('bwd')
    torch.cuda.synchronize()
    ours_map = s.output_view(0).float().cpu()
    pose = step.losses()[0]
    m._attach_grads()
    table = G.ParamTable(keys, bucket_of=G.hrnet_bucket_of)
    ours_g = torch.cat([m._view(k, grad=True).reshape(-1) for k in table.trainable_keys()]).double().cpu()
    # interpreter: same graph class / flags, bf16 storage
    g = G.HRNetGraph(table, ex, J, B, H, W, True, wlp_is_master=False, fp8=True)
    A = U.make_arenas(g, table, G.plan_memory(g.fwd + g.bwd, reuse_delay=4), act_dtype=torch.bfloat16)
    U.load_params(A, table, sd)
    A.t['image'].copy_(x.reshape(-1))
    PI.run(A, [U.wprep_op(g, table)] + g.fwd)
    out = A.view(g.outputs[0].buf).float()
    w2 = (tw.reshape(B, J) ** 2)[:, None, None, :]
    diff = out - tg.permute(0, 2, 3, 1)
    ref_pose = float(0.5 * (w2 * diff * diff).sum() / tg.numel())
    A.view(g.out_grads[0].buf).copy_((w2 * diff / tg.numel()).to(torch.bfloat16))
    PI.run(A, g.bwd)
    ref_g = torch.cat([A.view(table.grad(k)).reshape(-1) for k in table.trainable_keys()]).double()
    rel_map = float((ours_map - out).norm() / out.norm())
    rel_g = float((ours_g - ref_g).norm() / ref_g.norm())
    cos = float((ours_g * ref_g).sum() / (ours_g.norm() * ref_g.norm()))
    print('hrnet fp8 train step vs interpreter: map rel-L2 %.3e, loss %.6f vs %.6f, gradient rel-L2 %.3e cosine %.4f'
          % (rel_map, pose, ref_pose, rel_g, cos))
    assert rel_map < 0.5 and abs(pose - ref_pose) < 5e-2 * ref_pose, (rel_map, pose, ref_pose)
    assert torch.isfinite(ours_g).all() and 0.5 < float(ours_g.norm() / ref_g.norm()) < 2.0, float(ours_g.norm() / ref_g.norm())
