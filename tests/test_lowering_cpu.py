"""Host logic of the lowering (graph IR -> C structs) for the benchmark student, without a GPU: the library's dispatch
predicates are pure host code and the structs only carry addresses, so the whole backward of hg4x128 at batch 32 can be
lowered here against fake arena addresses.  Checks the two lowering-time fusions of round 3 -- 1x1 weight gradients formed
by their data-gradient launch (fpd_conv_t.wg_partial) and BN-backward applies evaluated by the data gradient that consumes
them (fpd_conv_t.fold_x) -- structurally: how many, wired to the right tensors, and ordered so that every reader of a
materialised operand comes after the launch that writes it."""
import pytest

from oracle import hourglass_ref


class FakeArenas:
    """Arena name -> a distinct fake base address; nothing is ever dereferenced on the CPU."""
    device, dtype, parent = 'cpu', 1, None

    def __init__(self):
        self.bases = {}

    def ptr(self, buf):
        if buf is None:
            return None
        base = self.bases.setdefault(buf.arena, (len(self.bases) + 1) << 40)
        return base + buf.off * 8


def _lower(monkeypatch, fold):
    from fpd_amd import executor as E, graph as G, runtime as R
    R.lib()
    monkeypatch.setenv('FPD_FOLD_APPLY', '1' if fold else '0')
    g = G.HourglassGraph(G.ParamTable(hourglass_ref.hourglass_keys(128, 4, 16)), 128, 4, 16, 32, 256, 256, train=True)
    G.plan_memory(g.fwd + g.bwd, reuse_delay=400)
    low = E.Lowering(FakeArenas(), 1)
    low.use_partials = True
    bwd = [o for o in g.bwd if o.kind != 'seed']
    low.plan_folds(bwd)
    lowered = [low.op(o) for o in bwd]
    return R, low, bwd, lowered


def _members(op):
    return [m for m in ((op.a, op.b) if op.kind in ('conv2', 'ew2') else (op,)) if m is not None]


def test_backward_lowering_of_the_benchmark_student_folds_and_fuses(monkeypatch):
    R, low, bwd, lowered = _lower(monkeypatch, fold=True)
    flat = [(i, m) for i, o in enumerate(bwd) for m in _members(o)]
    cands = [m for _, m in flat if m.kind == 'conv' and getattr(m, 'fold_apply', None) is not None]
    active = [m for m in cands if getattr(m, 'fold_active', False)]
    # every Bottleneck contributes two foldable applies (bn2 -> conv1's data gradient, bn3 -> conv2's); at batch 32 every one
    # of those launches is served by the persistent kernel (>= 256 tiles) or by a FOLD variant of the halo-tile kernel
    assert len(cands) == len(active) == 118
    assert sum(1 for _, m in flat if getattr(m, 'folded', False)) == 118
    # 1x1 weight gradients formed by their data-gradient launch: 57 while conv_pp served the >= 32-high levels only (round 3); 105
    # since conv_c1 (round 6, from 2048 pixels) also takes the 16x16 and 8x8 levels' 128 <-> 64 data gradients (48 more); 102 since
    # it serves the three 128 -> 128 data gradients (fc_) as plain data gradients (their dW tiles do not fit its LDS)
    assert len(low.fused) == 102
    A = low.A
    pos = {id(m): i for i, m in flat}
    for i, op in enumerate(bwd):
        code, st = lowered[i]
        structs = [st.a, st.b] if op.kind == 'conv2' else [st]
        for m, s in zip(_members(op), structs):
            if not (m.kind == 'conv' and getattr(m, 'fold_active', False)):
                continue
            ap = m.fold_apply
            assert ap.add is None and s.x == A.ptr(ap.dy.buf) and s.fold_x == A.ptr(ap.x.buf) and s.fold_stats == A.ptr(ap.bstats)
            assert s.fold_dgamma == A.ptr(ap.dgamma) and s.fold_dbeta == A.ptr(ap.dbeta)
            fused = getattr(m, 'fused_active', False)
            assert (s.fold_out is None) == fused          # materialised only for a SEPARATE weight-gradient launch
            assert pos[id(ap)] < pos[id(m)]               # the (no-op) apply still precedes: the IR order is unchanged
            readers = [j for j, r in flat if r is not m and r is not ap and any(t is ap.y for t in r.acts_in())]
            assert all(j > pos[id(m)] for j in readers), 'a reader of the evaluated operand precedes the launch that writes it'
            assert fused or readers                        # ... and if nothing else read it, it would not be written
    # folded applies are lowered as no-ops; a pair that lost one member goes out as a single launch
    for i, op in enumerate(bwd):
        folded = [getattr(m, 'folded', False) for m in _members(op)]
        if op.kind == 'ew' and folded[0]:
            assert lowered[i][0] == R.OP_NOP
        if op.kind == 'ew2' and any(folded):
            assert lowered[i][0] == (R.OP_NOP if all(folded) else R.OP_EW)
    launches = sum(1 for c, _ in lowered if c != R.OP_NOP)
    R2, low2, bwd2, lowered2 = _lower(monkeypatch, fold=False)
    assert not any(getattr(m, 'folded', False) for o in bwd2 for m in _members(o))
    assert sum(1 for c, _ in lowered2 if c != R2.OP_NOP) - launches == 86      # launches that leave the backward (808 -> 722 per step)
