"""Helpers to run the product's graph through the CPU interpreter (oracle/plan_interp.py)."""
import torch

from fpd_amd import graph as G
from oracle import plan_interp as PI


def load_params(A, table, sd):
    """state_dict (OIHW) -> arenas (conv weights K,R,S,C)."""
    for k, _ in table.keys:
        v = sd[k]
        dst = A.view(table[k])
        if v.dim() == 4:
            dst.copy_(v.permute(0, 2, 3, 1))
        else:
            dst.copy_(v.reshape(dst.shape))


def make_arenas(g, table, act_size, act_dtype=torch.float32, extra=None):
    sizes = {'param': table.sizes['param'], 'grad': table.sizes['param'], 'rstat': table.sizes['rstat'],
             'nbt': table.sizes['nbt'], 'wlp': g.wlp_size, 'act': act_size, 'stats': g.stats_size,
             'image': g.N * 3 * g.H * g.W, 'losses': 2}
    sizes.update(extra or {})
    return PI.Arenas(sizes, act_dtype)


def wprep_op(g, table):
    entries = []
    for k in table.conv_keys():
        if k in getattr(g, 'MASTER_ONLY', ('conv1.weight',)):
            continue
        entries.append({'w': table[k], 'w_fwd': g.wfwd.get(k), 'w_bwd': g.wbwd.get(k)})
    return G.Op('wprep', entries=entries)


def flat_grads_oihw(A, table):
    """gradient arena -> flat vector in nn.Module.parameters() order with OIHW conv weights."""
    out = []
    for k in table.trainable_keys():
        g = A.view(table.grad(k))
        if g.dim() == 4:
            g = g.permute(0, 3, 1, 2)
        out.append(g.reshape(-1))
    return torch.cat(out)
