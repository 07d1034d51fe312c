"""Host logic of the fp8 forward path (BASELINE configs[4]) without a GPU: which convolutions of HRNet-W32 go to the fp8
kernel (graph.f8_conv_domain must agree with the library's own fpd_conv_f8_in_domain), arena layout of the e4m3 copies,
and the op list executed by the CPU interpreter with the fp8 specification (oracle/fp8_ref.py) against the bf16 one."""
import numpy as np
import torch

from fpd_amd import graph as G
from oracle import fp8_ref, fpd_ref, hrnet_ref, plan_interp as PI
from tests import _interp_util as U
from tests._cases_hrnet import CONFIG, extra_cfg


def test_fp8_domain_rule_matches_the_library_and_layout_is_aligned():
    from fpd_amd import runtime as R
    ex = extra_cfg(dict(widths=[32, 64, 128, 256], blocks=4, modules=(1, 4, 3)))
    keys = hrnet_ref.hrnet_keys(ex, 17)
    table = G.ParamTable(keys, bucket_of=G.hrnet_bucket_of)
    g = G.HRNetGraph(table, ex, 17, 2, 384, 288, True, wlp_is_master=False, fp8=True)
    convs = [o for o in g.fwd if o.kind == 'conv']
    n8 = 0
    for o in convs:
        c = R.ConvT()
        (c.N, c.H, c.W, c.C, c.K, c.R, c.S, c.stride, c.pad, c.P, c.Q) = o.dims
        c.dtype, c.epi = R.BF16, R.EPI_PLAIN
        assert bool(R.lib().fpd_conv_f8_in_domain(c)) == G.f8_conv_domain(o.dims), o.dims
        assert (o.w8 is not None) == G.f8_conv_domain(o.dims)
        n8 += o.w8 is not None
    # everything but the two stride-2 stem convolutions, the stride-2 fuse / transition convolutions and final_layer (K=17)
    s2 = sum(1 for o in convs if o.dims[7] == 2)
    assert n8 == len(convs) - s2 - 1, (n8, len(convs), s2)
    mac8 = sum(np.prod([o.dims[0], o.dims[9], o.dims[10], o.dims[3], o.dims[4], o.dims[5], o.dims[6]]) for o in convs if o.w8 is not None)
    mac = sum(np.prod([o.dims[0], o.dims[9], o.dims[10], o.dims[3], o.dims[4], o.dims[5], o.dims[6]]) for o in convs)
    assert mac8 / mac > 0.85                       # stride-1 3x3 + 1x1 convolutions are ~90 % of the MACs
    offs = sorted((q.off, q.numel) for q, _ in g.w8.values())
    for (o1, n1), (o2, _) in zip(offs, offs[1:]):
        assert o1 % 16 == 0 and o1 + n1 <= o2
    assert g.w8_size >= sum(n for _, n in offs) and g.w8s_size >= sum(sc.numel for _, sc in g.w8.values())
    # backward operands are untouched: data / weight gradients read the bf16 copies
    assert all(getattr(o, 'w8', None) is None for o in g.bwd if o.kind == 'conv')
    # the fp32 parity build never uses fp8
    g32 = G.HRNetGraph(table, ex, 17, 1, 256, 192, False, wlp_is_master=True, fp8=True)
    assert not getattr(g32, 'w8', None)


def test_fp8_forward_in_the_interpreter_stays_close_to_bf16():
    c = CONFIG
    ex = extra_cfg(c['t'])
    keys = hrnet_ref.hrnet_keys(ex, c['joints'])
    table = G.ParamTable(keys, bucket_of=G.hrnet_bucket_of)
    inp, _, _ = fpd_ref.synth_batch(100, c['batch'], c['joints'], c['image'], c['heat'])
    sd = fpd_ref.synth_state_dict(keys, 2)
    outs = {}
    for fp8 in (False, True):
        g = G.HRNetGraph(table, ex, c['joints'], c['batch'], c['image'][1], c['image'][0], False, wlp_is_master=False, fp8=fp8)
        act = G.plan_memory(g.fwd)
        A = U.make_arenas(g, table, act, act_dtype=torch.bfloat16)
        U.load_params(A, table, sd)
        A.t['image'].copy_(inp.reshape(-1))
        PI.run(A, [U.wprep_op(g, table)] + g.fwd)
        outs[fp8] = A.view(g.outputs[0].buf).float().clone()
        assert (sum(1 for o in g.fwd if o.kind == 'conv' and o.w8 is not None) > 10) == fp8
    rel = float((outs[True] - outs[False]).norm() / outs[False].norm())
    assert 1e-4 < rel < 0.35, rel                  # e4m3 operands: a few per cent per convolution, ~25 layers deep


def test_quantisation_spec():
    w = torch.tensor([[0.0, 0.0], [1.0, -3.0], [1e-3, 7e-4]])
    q, s = fp8_ref.quant_weights(w)
    assert s[0] == 1.0 and torch.allclose(s[1:], torch.tensor([3.0 / 448, 1e-3 / 448]))
    assert torch.equal(q[1], torch.tensor([144.0, -448.0]))          # 149.33 -> 144 (e4m3 grid step 16 above 128)
    x = torch.tensor([1e4, -1e4, 0.3, 1.0 / 512, 1.0 / 1024 * 0.49])
    assert torch.equal(fp8_ref.quant_e4m3(x), torch.tensor([448.0, -448.0, 0.3125, 1.0 / 512, 0.0]))
