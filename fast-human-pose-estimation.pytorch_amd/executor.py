"""Device-side realisation of graph.py: arenas in HBM, lowering of the op list to the C ABI, native
execution plans (one C call or one hipGraph replay per phase), and the fused FPD train step.

PyTorch is used here only as the HBM allocator, stream provider and RCCL front end
(torch.distributed); every kernel that touches the data is in csrc/ (libfpd_amd.so).
"""
import ctypes as C
import os

import torch

from . import graph as G
from . import runtime as R
from .schedule import PhaseSchedule

_EW = {'bnrelu_fwd': R.EW_BNRELU_FWD, 'bnrelu_bwd_r': R.EW_BNRELU_BWD_R, 'bn_bwd_apply': R.EW_BN_BWD_APPLY,
       'maxpool_fwd': R.EW_MAXPOOL_FWD, 'maxpool_bwd': R.EW_MAXPOOL_BWD, 'upadd_fwd': R.EW_UPADD_FWD,
       'sumpool': R.EW_SUMPOOL, 'add': R.EW_ADD, 'relu_mask': R.EW_RELU_MASK, 'dilate2': R.EW_DILATE2}
_ARENA_DTYPE = {'param': torch.float32, 'grad': torch.float32, 'rstat': torch.float32, 'nbt': torch.int64,
                'stats': torch.int64, 'losses': torch.float64, 'image': torch.float32, 'target': torch.float32,
                'weight': torch.float32, 'adam_m': torch.float32, 'adam_v': torch.float32, 'fold': torch.float32,
                'w8': torch.uint8, 'w8s': torch.float32}


def act_torch_dtype(dtype):
    return torch.bfloat16 if dtype == R.BF16 else torch.float32


_STAT_HI, _STAT_LO = float(2 ** 20), float(2 ** 60)      # include/fpd_amd.h fpd_stat_t: value = hi * 2^-20 + lo * 2^-60


class Arenas:
    """name -> flat device tensor.  Lookups fall through to `parent` (model-level arenas).

    The 'stats' arena holds the exact statistics of include/fpd_amd.h (fpd_stat_t): the IR sizes a buffer over C channels
    as [R][2][C] logical values (what the CPU interpreter keeps as doubles); on the device every logical value is TWO
    64-bit integer limbs, laid out [R][2 sums][2 limbs][C], so the arena has two words per IR element."""

    def __init__(self, device, dtype, parent=None):
        self.device, self.dtype, self.parent = device, dtype, parent
        self.t = {}

    def alloc(self, name, n):
        dt = _ARENA_DTYPE.get(name, act_torch_dtype(self.dtype))
        n = max(int(n), 4) * (2 if name == 'stats' else 1)
        self.t[name] = torch.zeros(n, dtype=dt, device=self.device)
        return self.t[name]

    def stats_read(self, buf):
        """[R][2][C] float64: the value of every (replica, sum, channel) of a statistics buffer (a copy)."""
        raw = self.view(buf)
        return raw[:, :, 0, :].double() / _STAT_HI + raw[:, :, 1, :].double() / _STAT_LO

    def stats_write(self, buf, values):
        """Store [R][2][C] float64 values into a statistics buffer (the split a kernel's contribution goes through)."""
        v = values.to(self.device, torch.float64).reshape(buf.shape)
        hi = torch.round(v * _STAT_HI)
        lo = torch.round((v - hi / _STAT_HI) * _STAT_LO)
        self.view(buf).copy_(torch.stack([hi.to(torch.int64), lo.to(torch.int64)], 2))

    def tensor(self, name):
        if name in self.t:
            return self.t[name]
        if self.parent is not None:
            return self.parent.tensor(name)
        raise KeyError(name)

    def ptr(self, buf):
        if buf is None:
            return None
        t = self.tensor(buf.arena)
        return t.data_ptr() + buf.off * t.element_size() * (2 if buf.arena == 'stats' else 1)

    def view(self, buf):
        if buf.arena == 'stats':                   # raw limbs [R][2 sums][2 limbs][C]
            r, two, c = buf.shape
            return self.tensor('stats')[2 * buf.off:2 * (buf.off + buf.numel)].view(r, two, 2, c)
        return self.tensor(buf.arena)[buf.off:buf.off + buf.numel].view(buf.shape)


def _abuf(a):
    return None if a is None else (a.buf if isinstance(a, G.Act) else a)


# FPD_WHATIF=token[,token...]: TIMING-ONLY experiments (results are wrong when set): the named class of ops is lowered to
# no-ops, which shows what that class costs inside the pipelined step (experiments/r03/whatif.sh).  Student graph:
# nowgrad / nowgrad_small / nowgrad_big (weight gradients, by map height <= 16 / >= 64), noapply (BN-backward applies),
# nobigconv (convolutions on >= 64x64 maps), nobig / nomid / nosmall (every op on >= 64 / 32 / <= 16 high maps, weight
# gradients excepted), noew.  Teacher graph: t_all, t_big, t_mid, t_small; t_bneck_big / t_head / t_plain_big (the >= 64-high ops by kind).
_WHATIF = frozenset(t for t in os.environ.get('FPD_WHATIF', '').split(',') if t)
if _WHATIF:
    import sys
    sys.stderr.write('[fpd_amd] WARNING: FPD_WHATIF=%s is set: op classes are dropped from the plans -- TIMING ONLY, every result '
                     '(maps, losses, gradients, parameters) of this process is WRONG\n' % ','.join(sorted(_WHATIF)))


def _whatif_drop(op, train):
    k = op.kind
    if k in ('conv2', 'bneck2', 'ew2'):
        sub = op.a
        k = sub.kind
    else:
        sub = op
    dims = getattr(sub, 'dims', None)
    h = dims[1] if dims else 0
    if not train:
        if k not in ('conv', 'bneck', 'head', 'ew', 'stem_fwd'):
            return False
        # round 5: the teacher's >= 64-high ops by kind -- fused Bottlenecks (persistent, capped grid), fused heads, and the rest
        # (stem, layer1 / layer2 convolutions, max-pool: plain full-grid launches)
        if h >= 64 and (('t_bneck_big' in _WHATIF and k == 'bneck') or ('t_head' in _WHATIF and k == 'head') or
                        ('t_plain_big' in _WHATIF and k in ('conv', 'ew', 'stem_fwd'))):
            return True
        return ('t_all' in _WHATIF or ('t_big' in _WHATIF and h >= 64) or ('t_mid' in _WHATIF and h == 32) or
                ('t_small' in _WHATIF and h <= 16))
    if k in ('wgrad', 'stem_wgrad', 'wreduce'):
        return ('nowgrad' in _WHATIF or ('nowgrad_small' in _WHATIF and k == 'wgrad' and h <= 16) or
                ('nowgrad_big' in _WHATIF and k == 'wgrad' and h >= 64))
    if k not in ('conv', 'ew', 'stem_fwd'):
        return False
    if k == 'ew' and ('noew' in _WHATIF or ('noapply' in _WHATIF and sub.op == 'bn_bwd_apply')):
        return True
    if k == 'conv' and 'nobigconv' in _WHATIF and h >= 64:
        return True
    if k == 'conv' and h >= 32 and (('no3x3big' in _WHATIF and dims[5] == 3) or ('no1x1big' in _WHATIF and dims[5] == 1)):
        return True                                        # round 6: the >= 32-high convolutions by filter size
    return ('nobig' in _WHATIF and h >= 64) or ('nomid' in _WHATIF and h == 32) or ('nosmall' in _WHATIF and h <= 16)


class Lowering:
    """IR op -> (native op code, ctypes struct)."""

    def __init__(self, arenas, dtype):
        self.A, self.dtype = arenas, dtype
        self.keep = []          # device tables that must outlive the plan
        self.use_partials = False
        self.partials, self.partial_elems = {}, 0      # id(IR wgrad op) -> slab record
        self.fused = set()                             # id(IR wgrad op) computed inside its data-gradient launch (conv_pp)
        self.reduces = []                              # (IR wreduce op, its TableT) filled in by finish_partials()

    def bn(self, bn):
        s = R.BnT()
        if bn is None:
            s.mode = R.BN_NONE
            return s
        s.mode = R.BN_TRAIN if bn.mode == 'train' else R.BN_EVAL
        s.relu = 1 if bn.relu else 0
        s.eps = G.BN_EPS
        s.stats = self.A.ptr(bn.stats) if bn.mode == 'train' else None
        s.gamma, s.beta = self.A.ptr(bn.gamma), self.A.ptr(bn.beta)
        s.running_mean, s.running_var = self.A.ptr(bn.rmean), self.A.ptr(bn.rvar)
        return s

    def conv(self, op, plain=False):
        s = R.ConvT()
        (s.N, s.H, s.W, s.C, s.K, s.R, s.S, s.stride, s.pad, s.P, s.Q) = op.dims
        s.dtype = self.dtype
        s.epi = R.EPI_BNRELU_BWD if op.epi == 'bnrelu_bwd' else R.EPI_PLAIN
        p = self.A.ptr
        s.x, s.w, s.bias, s.residual, s.y = p(_abuf(op.x)), p(op.w), p(op.bias), p(_abuf(op.residual)), p(_abuf(op.y))
        s.out_stats = p(op.out_stats)
        s.bn = self.bn(op.bn)
        s.epi_x, s.epi_bn, s.epi_stats = p(_abuf(op.epi_x)), self.bn(op.epi_bn), p(op.epi_stats)
        s.wg_partial, s.wg_stride, s.wg_bias, s.wg_count = None, 0, 0, 0
        if not plain and getattr(op, 'fused_wgrad', None) is not None and self.use_partials:
            n = R.lib().fpd_conv_fused_wgrad_partials(C.byref(s))
            if n > 0:
                self._fuse_wgrad(op, s, n)
        if not plain:
            self._fill_fold(op, s)
        if getattr(op, 'w8', None) is not None and not plain:
            f = R.ConvF8T()
            f.c, f.w8, f.w8_scale = s, p(op.w8), p(op.w8s)
            return R.OP_CONV_F8, f
        return R.OP_CONV, s

    def plan_folds(self, ops):
        """Decide, BEFORE the list is lowered, which BN-backward applies are evaluated by the data gradient that consumes
        them (graph: conv.fold_apply; include/fpd_amd.h: fpd_conv_t.fold_x): where the library serves the launch (pair)
        that way the apply is marked `folded` (lowered as a no-op) and the convolution `fold_active`."""
        if os.environ.get('FPD_FOLD_APPLY', '1') == '0':
            return
        l = R.lib()
        members_of = lambda top: [m for m in ((top.a, top.b) if top.kind in ('conv2', 'ew2', 'bneck2') else (top,)) if m is not None]
        for i, op in enumerate(ops):
            if op is None or op.kind not in ('conv', 'conv2'):
                continue
            members = members_of(op)
            cands = [m for m in members if getattr(m, 'fold_apply', None) is not None and not getattr(m.fold_apply, 'folded', False)]
            if not cands or any(getattr(m, 'w8', None) is not None for m in members):
                continue
            # Who else reads an apply's result?  Its own convolution's weight gradient is served from the launch's operand image
            # when that is fused too; ANY other reader (a tensor aliased into a residual-path gradient, an op on another lane)
            # needs the evaluated operand in memory: the launch then writes it out (fold_out).  A reader that comes BEFORE the
            # convolution in the list would read it before it exists: no fold.
            early = False
            for m in cands:
                y = _abuf(m.fold_apply.y)
                own = (id(m), id(m.fold_apply), id(getattr(m, 'fold_wgrad', None)))
                m.fold_other_readers = False
                for j, top in enumerate(ops):
                    if top is None:
                        continue
                    for o in members_of(top):
                        if id(o) in own or not any(_abuf(t) is y for t in o.acts_in()):
                            continue
                        if j < i:
                            early = True
                        m.fold_other_readers = True
            if early:
                continue
            if op.kind == 'conv2':
                ps = R.ConvPairT()
                ps.a, ps.b = self.conv(op.a, plain=True)[1], self.conv(op.b, plain=True)[1]
                ok = l.fpd_conv_pair_fold_supported(C.byref(ps)) == 1
            else:
                ok = l.fpd_conv_fold_supported(C.byref(self.conv(op, plain=True)[1])) == 1
            if ok:
                for m in cands:
                    m.fold_active = True
                    m.fold_apply.folded = True

    def _fill_fold(self, op, s):
        """Fold fields of a data gradient that evaluates its BN-backward apply itself (plan_folds decided)."""
        if not getattr(op, 'fold_active', False):
            return
        ap, p = op.fold_apply, self.A.ptr
        assert ap.add is None and _abuf(ap.y) is _abuf(op.x), 'fold: the apply must produce exactly this operand'
        s.x = p(_abuf(ap.dy))                               # the masked gradient; the operand proper is evaluated in-kernel
        s.fold_x, s.fold_bn, s.fold_stats = p(_abuf(ap.x)), self.bn(ap.bn), p(ap.bstats)
        s.fold_dgamma, s.fold_dbeta = p(ap.dgamma), p(ap.dbeta)
        # its other consumer -- the convolution's weight gradient -- reads the evaluated operand unless it is formed right here
        skip = getattr(op, 'fused_active', False) and not getattr(op, 'fold_other_readers', False)
        s.fold_out = None if skip else p(_abuf(ap.y))

    def _fuse_wgrad(self, op, s, n):
        """The data-gradient launch `op` (struct `s`, a ConvT -- possibly a field of a pair struct) also forms the weight
        gradient of `op.fused_wgrad` into `n` slabs: register them like a weight-gradient kernel's (finish_partials patches
        the struct, the bucket's 'wreduce' sums them) and turn the separate 'wgrad' op into a no-op."""
        fw = op.fused_wgrad
        numel = fw.dw.numel                                 # [Kf][1][1][Cf] = [s.C][s.K]
        assert numel == s.C * s.K, (numel, s.C, s.K)
        stride = (numel + s.C + 63) // 64 * 64              # weight slab + bias partials
        self.partials[id(fw)] = [s, fw.dw, numel, stride, n, self.partial_elems, fw.dbias, s.C]
        self.partial_elems += n * stride
        s.wg_bias = 1 if fw.dbias is not None else 0
        s.wg_count = n                                      # the launch refuses a geometry that writes another number of slabs
        self.fused.add(id(fw))
        op.fused_active = True

    def wquant(self, entries):
        """[(master weight Buf, e4m3 Buf, scale Buf)] -> one table-driven launch (fpd_weight_quant_f8)."""
        ents = []
        for w, q, sc in entries:
            e = R.WquantEntryT()
            e.w, e.w8, e.scale = self.A.ptr(w), self.A.ptr(q), self.A.ptr(sc)
            e.K, e.RSC = w.shape[0], w.numel // w.shape[0]
            ents.append(e)
        s = R.TableT()
        s.table, s.n, s.dtype, s.max_elems = self._table(ents, R.WquantEntryT), len(ents), self.dtype, 0
        return R.OP_WQUANT, s

    def bneck(self, op):
        s = R.BneckT()
        (s.N, s.H, s.W, s.C, s.P) = op.dims
        s.dtype = self.dtype
        p = self.A.ptr
        s.x, s.y = p(_abuf(op.x)), p(_abuf(op.y))
        s.w1, s.b1, s.w2, s.b2, s.w3, s.b3 = p(op.w1), p(op.b1), p(op.w2), p(op.b2), p(op.w3), p(op.b3)
        s.bn1, s.bn2, s.bn3 = self.bn(op.bn1), self.bn(op.bn2), self.bn(op.bn3)
        s.folded = p(getattr(op, 'folded', None))
        return R.OP_BNECK, s

    def head(self, op):
        s = R.HeadT()
        (s.N, s.H, s.W, s.C, s.J) = op.dims
        s.dtype = self.dtype
        p = self.A.ptr
        s.y0, s.x, s.score, s.next = p(_abuf(op.y0)), p(_abuf(op.x)), p(_abuf(op.score)), p(_abuf(op.next))
        s.w_fc, s.b_fc, s.w_score, s.b_score = p(op.w_fc), p(op.b_fc), p(op.w_score), p(op.b_score)
        s.w_fc2, s.b_fc2, s.w_score2, s.b_score2 = p(op.w_fc2), p(op.b_fc2), p(op.w_score2), p(op.b_score2)
        s.bn = self.bn(op.bn)
        s.folded = p(getattr(op, 'folded', None))
        return R.OP_HEAD, s

    def wgrad(self, op):
        if id(op) in self.fused:                           # formed by the data-gradient launch (fpd_conv_t.wg_partial)
            return R.OP_NOP, R.MemsetT()
        s = R.WgradT()
        (s.N, s.H, s.W, s.C, s.K, s.R, s.S, s.stride, s.pad, s.P, s.Q) = op.dims
        s.dtype = self.dtype
        p = self.A.ptr
        s.x, s.dy, s.dw, s.dbias = p(_abuf(op.x)), p(_abuf(op.dy)), p(op.dw), p(op.dbias)
        s.bn = self.bn(op.bn)
        s.partial, s.partial_stride = None, 0
        if self.use_partials:
            # two-stage reduction instead of device-scope atomics: ask the library how many slabs this launch writes
            n = R.lib().fpd_wgrad_num_partials(C.byref(s))
            if n > 0:
                numel = op.dw.numel
                stride = (numel + s.K + 63) // 64 * 64          # weight slab + bias partials
                self.partials[id(op)] = [s, op.dw, numel, stride, n, self.partial_elems, op.dbias, s.K]
                self.partial_elems += n * stride
        return R.OP_WGRAD, s

    def wreduce(self, op):
        """Second stage of the weight-gradient reduction for the convolutions of ONE gradient bucket (op.wgrads, all
        lowered before this op): dw += sum of slabs, one launch.  The table is filled in by finish_partials()."""
        if not any(id(w) in self.partials for w in op.wgrads):
            return R.OP_NOP, R.MemsetT()
        t = R.TableT()
        self.reduces.append((op, t))
        return R.OP_WREDUCE, t

    def finish_partials(self):
        """Allocate the slab workspace, patch the recorded wgrad structs and fill the reduce tables (call after every
        backward op has been lowered and BEFORE the structs are copied into the plan)."""
        if not self.partials:
            return
        ws = torch.empty(self.partial_elems, dtype=torch.float32, device=self.A.device)
        self.keep.append(ws)
        for s, dw, numel, stride, n, off, dbias, nbias in self.partials.values():
            if isinstance(s, R.ConvT):                    # fused into a data-gradient launch
                s.wg_partial, s.wg_stride = ws.data_ptr() + 4 * off, stride
            else:
                s.partial, s.partial_stride = ws.data_ptr() + 4 * off, stride
        for op, t in self.reduces:
            ents, mx = [], 0
            for w in op.wgrads:
                rec = self.partials.get(id(w))
                if rec is None:
                    continue
                s, dw, numel, stride, n, off, dbias, nbias = rec
                base = ws.data_ptr() + 4 * off
                e = R.WreduceEntryT()
                e.partial, e.dw, e.n, e.stride, e.count = base, self.A.ptr(dw), numel, stride, n
                ents.append(e)
                if dbias is not None:
                    e = R.WreduceEntryT()
                    e.partial, e.dw, e.n, e.stride, e.count = base + 4 * numel, self.A.ptr(dbias), nbias, stride, n
                    ents.append(e)
                mx = max(mx, numel)
            t.table, t.n, t.dtype, t.max_elems = self._table(ents, R.WreduceEntryT), len(ents), self.dtype, mx

    def stem(self, op):
        s = R.StemT()
        (s.N, s.H, s.W, s.K, s.P, s.Q) = op.dims
        s.dtype = self.dtype
        p = self.A.ptr
        s.x = p(op.image)
        if op.kind == 'stem_fwd':
            s.w, s.bias, s.y, s.out_stats = p(op.w), p(op.bias), p(_abuf(op.y)), p(op.out_stats)
            return R.OP_STEM_FWD, s
        s.dy, s.dw, s.dbias = p(_abuf(op.dy)), p(op.dw), p(op.dbias)
        s.partial, s.partial_stride = None, 0
        if self.use_partials:                  # same two-stage reduction as the other weight gradients (no atomics)
            n = R.lib().fpd_stem_wgrad_num_partials(C.byref(s))
            if n > 0:
                numel = op.dw.numel
                stride = (numel + s.K + 63) // 64 * 64
                self.partials[id(op)] = [s, op.dw, numel, stride, n, self.partial_elems, op.dbias, s.K]
                self.partial_elems += n * stride
        return R.OP_STEM_WGRAD, s

    def ew(self, op):
        if getattr(op, 'folded', False):                   # evaluated by the data gradient that consumes it (plan_folds)
            return R.OP_NOP, R.MemsetT()
        s = R.EwT()
        s.op, s.dtype = _EW[op.op], self.dtype
        (s.N, s.H, s.W, s.C) = op.dims
        p = self.A.ptr
        s.x, s.x2, s.dy, s.add, s.y = p(_abuf(op.x)), p(_abuf(op.x2)), p(_abuf(op.dy)), p(_abuf(op.add)), p(_abuf(op.y))
        s.out_stats, s.bstats, s.dgamma, s.dbeta = p(op.out_stats), p(op.bstats), p(op.dgamma), p(op.dbeta)
        s.bn = self.bn(op.bn)
        return R.OP_EW, s

    def _table(self, entries, cls):
        arr = (cls * len(entries))(*entries)
        raw = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8).to(self.A.device)
        self.keep.append(raw)
        return raw.data_ptr()

    def bnupd(self, op):
        ents = []
        for bn in op.bns:
            e = R.BnupdEntryT()
            e.stats, e.running_mean, e.running_var = self.A.ptr(bn.stats), self.A.ptr(bn.rmean), self.A.ptr(bn.rvar)
            e.num_batches_tracked = self.A.ptr(bn.nbt)
            e.count, e.momentum, e.C = float(bn.count), G.BN_MOMENTUM, bn.C
            ents.append(e)
        s = R.TableT()
        s.table, s.n, s.dtype, s.max_elems = self._table(ents, R.BnupdEntryT), len(ents), self.dtype, 0
        return R.OP_BNUPD, s

    def wprep(self, entries):
        ents, mx = [], 0
        for w, wf, wb in entries:
            e = R.WprepEntryT()
            e.w, e.w_fwd, e.w_bwd = self.A.ptr(w), self.A.ptr(wf), self.A.ptr(wb)
            (e.K, e.R, e.S, e.C) = w.shape
            mx = max(mx, w.numel)
            ents.append(e)
        s = R.TableT()
        s.table, s.n, s.dtype, s.max_elems = self._table(ents, R.WprepEntryT), len(ents), self.dtype, mx
        return R.OP_WPREP, s

    def memset(self, arena_name, off=0, n=None):
        t = self.A.tensor(arena_name)
        n = t.numel() - off if n is None else n
        s = R.MemsetT()
        s.ptr, s.bytes = t.data_ptr() + off * t.element_size(), n * t.element_size()
        return R.OP_MEMSET, s

    def op(self, op):
        if _WHATIF and _whatif_drop(op, getattr(self, 'train', True)):
            return R.OP_NOP, R.MemsetT()
        if op.kind == 'conv2':
            s = R.ConvPairT()
            s.a, s.b = self.conv(op.a, plain=True)[1], self.conv(op.b, plain=True)[1]
            if (self.use_partials and getattr(op.a, 'fused_wgrad', None) is not None and
                    getattr(op.b, 'fused_wgrad', None) is not None):
                na, nb = C.c_int32(0), C.c_int32(0)
                R.check(R.lib().fpd_conv_pair_fused_wgrad_partials(C.byref(s), C.byref(na), C.byref(nb)), 'fpd_conv_pair_fused_wgrad_partials')
                if na.value > 0 and nb.value > 0:          # s.a / s.b are views INTO the pair struct: patched in place later
                    self._fuse_wgrad(op.a, s.a, na.value)
                    self._fuse_wgrad(op.b, s.b, nb.value)
            self._fill_fold(op.a, s.a)
            self._fill_fold(op.b, s.b)
            return R.OP_CONV_PAIR, s
        if op.kind == 'ew2':
            fa, fb = getattr(op.a, 'folded', False), getattr(op.b, 'folded', False)
            if fa or fb:                                   # folded members leave the pair
                return (R.OP_NOP, R.MemsetT()) if (fa and fb) else self.ew(op.b if fa else op.a)
            s = R.EwPairT()
            s.a, s.b = self.ew(op.a)[1], self.ew(op.b)[1]
            return R.OP_EW_PAIR, s
        if op.kind == 'bneck2':
            s = R.BneckPairT()
            s.a, s.b = self.bneck(op.a)[1], self.bneck(op.b)[1]
            return R.OP_BNECK_PAIR, s
        if op.kind == 'head':
            return self.head(op)
        if op.kind == 'head_fold':
            return R.OP_HEAD_FOLD, self.head(op.target)[1]
        if op.kind == 'bneck_fold':
            return R.OP_BNECK_FOLD, self.bneck(op.target)[1]
        if op.kind == 'affsum':
            s = R.AffsumT()
            (s.N, s.H, s.W, s.C) = op.dims
            s.dtype, s.relu, s.nterms = self.dtype, 1 if op.relu else 0, len(op.terms)
            assert len(op.terms) <= R.AFFSUM_MAX and op.out_stats is None
            for j, (t, bn, up) in enumerate(op.terms):
                s.t[j].x, s.t[j].bn, s.t[j].up = self.A.ptr(_abuf(t)), self.bn(bn), up
            s.y = self.A.ptr(_abuf(op.y))
            return R.OP_AFFSUM, s
        if op.kind == 'nchw2nhwc':
            s = R.LayoutT()
            (s.N, s.C, s.H, s.W) = op.dims
            s.src, s.dst, s.dtype = self.A.ptr(op.image), self.A.ptr(_abuf(op.y)), self.dtype
            return R.OP_NCHW2NHWC, s
        if op.kind == 'wreduce':
            return self.wreduce(op)
        if op.kind == 'grad_ready':
            return R.OP_NOP, R.MemsetT()
        return {'conv': self.conv, 'bneck': self.bneck, 'wgrad': self.wgrad, 'stem_fwd': self.stem, 'stem_wgrad': self.stem, 'ew': self.ew,
                'bnupd': self.bnupd}[op.kind](op)


class ModelState:
    """Flat HBM arenas of one model's state_dict (see graph.ParamTable)."""

    def __init__(self, table, device, dtype):
        self.table, self.device, self.dtype = table, device, dtype
        self.A = Arenas(device, dtype)
        for name in ('param', 'rstat', 'nbt'):
            self.A.alloc(name, table.sizes[name])
        self.A.alloc('grad', table.sizes['param'])


class GraphInstance:
    """One HourglassGraph realised on the device: activation/statistics/working-weight arenas + native plan.

    Plan ranges (self.rng): 'prep' (zero statistics, refresh working weight copies), 'fwd', and for training
    graphs 'bwd' (zero parameter gradients, backward ops).  `extra_ops` (e.g. the fused loss) can be spliced
    between fwd and bwd by the trainer before finalize()."""

    def __init__(self, state, cfg, batch, height, width, train, image=None, share_weights_with=None):
        self.state, self.train = state, train
        self._wlp_owner = share_weights_with       # another instance of the same model whose working weights we read
        self.dtype = state.dtype
        env = os.environ.get
        if cfg.get('arch') == 'hrnet':
            self.g = G.HRNetGraph(state.table, cfg['extra'], cfg['J'], batch, height, width, train,
                                  wlp_is_master=(self.dtype == R.F32),
                                  wgrad_batch=int(env('FPD_WGRAD_BATCH')) if env('FPD_WGRAD_BATCH') else None,
                                  fp8=bool(cfg.get('fp8', False)))
        else:
            self.g = G.HourglassGraph(state.table, cfg['F'], cfg['S'], cfg['J'], batch, height, width, train,
                                      num_blocks=cfg.get('num_blocks', 1), wlp_is_master=(self.dtype == R.F32),
                                      fuse_bneck=(self.dtype == R.BF16 and not train and env('FPD_FUSE_BNECK', '1') != '0'),
                                      pair_branches=env('FPD_PAIR', '1') != '0', fuse_head=env('FPD_FUSE_HEAD', '1') != '0',
                                      wgrad_batch=int(env('FPD_WGRAD_BATCH')) if env('FPD_WGRAD_BATCH') else None)
        self.A = Arenas(state.device, self.dtype, parent=state.A)
        self.low = Lowering(self.A, self.dtype)
        self.low.train = train
        self.plan = R.Plan()
        self.rng = {}
        self.graphs = {}
        self._image_ext = image
        self._finalized = False
        self.mid_ops = []          # IR ops between fwd and bwd (loss)
        self.mid_native = []       # callables adding native ops for the mid section

    def _schedule(self, name, ir_ops):
        """Multi-lane schedule of plan range `name`; ir_ops[k] is the IR op behind plan op begin+k (None = barrier)."""
        b, e = self.rng[name]
        assert e - b == len(ir_ops), (name, b, e, len(ir_ops))
        if not self.lanes_enabled or not any(o is not None and o.lane for o in ir_ops):
            return
        sch = PhaseSchedule([((o.lane or 0), o.accesses()) if o is not None else (0, None) for o in ir_ops], self.g.n_lanes)
        for k in range(len(ir_ops)):
            if sch.lanes[k] or sch.waits[k]:
                self.plan.set_schedule(b + k, sch.lanes[k], [b + w for w in sch.waits[k]])
        self.schedules[name] = sch

    def finalize(self):
        g = self.g
        ops = g.fwd + self.mid_ops + g.bwd
        self.lanes_enabled = os.environ.get('FPD_LANES', '1') != '0'
        self.schedules = {}
        # several lanes in flight: do not hand a released block to the very next tensor (false WAR dependencies)
        # (a frozen graph has a single lane: immediate reuse keeps its working set small)
        delay = int(os.environ.get('FPD_REUSE_DELAY', '400')) if self.lanes_enabled else 0
        if not self.train:
            delay = 0
        act = G.plan_memory(ops, reuse_delay=delay)
        self.act_elems = act
        self.A.alloc('act', act)
        self.A.alloc('stats', g.stats_size)
        if self._wlp_owner is not None:            # same ParamTable / same graph -> same layout of the working-weight arena
            o = self._wlp_owner                    # and of the folded BN tables, both written by the owner's 'prep' phase
            assert o.g.wlp_size == g.wlp_size and o.g.fold_size == g.fold_size and not self.train
            self.A.t['wlp'] = o.A.t['wlp']
            self.A.t['fold'] = o.A.t['fold']
        else:
            self.A.alloc('wlp', g.wlp_size)
            self.A.alloc('fold', g.fold_size)
        if self._image_ext is not None:
            self.A.t['image'] = self._image_ext
        else:
            self.A.alloc('image', g.N * 3 * g.H * g.W)
        p = self.plan
        b = len(p)
        p.add(*self.low.memset('stats'))
        entries = []
        for k in self.state.table.conv_keys():
            if k in g.MASTER_ONLY:
                continue
            wf, wb = g.wfwd.get(k), g.wbwd.get(k)
            if wf is not None or wb is not None:
                entries.append((self.state.table[k], wf, wb))
        if entries:
            p.add(*self.low.wprep(entries))
        w8 = getattr(g, 'w8', None)
        if w8:                                     # e4m3 copies + scales of the convolutions that run on the fp8 matrix pipe
            self.A.alloc('w8', g.w8_size)
            self.A.alloc('w8s', g.w8s_size)
            p.add(*self.low.wquant([(self.state.table[k], q, sc) for k, (q, sc) in w8.items()]))
        for op in g.fwd:                           # frozen fused Bottlenecks: fold BN + biases into tables once
            for sub in ((op.a, op.b) if op.kind == 'bneck2' else (op,)):
                if sub.kind == 'bneck' and getattr(sub, 'folded', None) is not None:
                    p.add(*self.low.op(G.Op('bneck_fold', target=sub)))
                if sub.kind == 'head' and getattr(sub, 'folded', None) is not None:
                    p.add(*self.low.op(G.Op('head_fold', target=sub)))
        self.rng['prep'] = (b, len(p))
        b = len(p)
        self.op_index = {}                         # id(IR op) -> plan op (bench.py times single recorded ops through it)
        for op in g.fwd:
            self.op_index[id(op)] = p.add(*self.low.op(op))
        self.rng['fwd'] = (b, len(p))
        self._schedule('fwd', list(g.fwd))
        b = len(p)
        for fn in self.mid_native:
            fn(p)
        self.rng['mid'] = (b, len(p))
        if self.train:
            b = len(p)
            p.add(*self.low.memset('grad'))
            self.low.use_partials = True
            bwd_ir = [op for op in g.bwd if op.kind != 'seed']
            self.low.plan_folds(bwd_ir)
            lowered = [self.low.op(op) for op in bwd_ir]
            self.low.finish_partials()                # patches the wgrad structs / fills the per-bucket reduce tables
            self.bucket_ops = {}                      # gradient bucket -> plan op after which its grad slice is final
            for ir, (code, st) in zip(bwd_ir, lowered):
                k = self.op_index[id(ir)] = p.add(code, st)
                if ir.kind == 'grad_ready':
                    self.bucket_ops[ir.bucket] = k
                    p.mark_event(k)
            self.rng['bwd'] = (b, len(p))
            self._schedule('bwd', [None] + bwd_ir)
        self._finalized = True
        return self

    def run(self, name, stream=None):
        gid = self.graphs.get(name)
        if gid is not None:
            self.plan.replay(gid, stream)
            return
        b, e = self.rng[name]
        self.plan.run(b, e, stream)

    def capture(self, names, cap_stream):
        """Capture plan ranges into hipGraphs (one per name) on `cap_stream` (a non-default stream; the ranges must
        already have run eagerly once so one-time kernel attributes are set).  run(name) then replays the graph:
        the ~1 700 launches of a step stop costing host time one by one."""
        torch.cuda.synchronize()
        with torch.cuda.stream(cap_stream):
            for n in names:
                b, e = self.rng[n]
                self.graphs[n] = self.plan.capture(b, e, C.c_void_p(cap_stream.cuda_stream))
        torch.cuda.synchronize()

    def image(self):
        return self.A.tensor('image').view(self.g.N, 3, self.g.H, self.g.W)

    def calibrate_running_stats(self):
        """After one train-mode forward: overwrite every BN's running estimates with this batch's statistics
        (unbiased variance).  Used to give random-init synthetic teachers sane eval-mode statistics
        (SURVEY.md section 8(d)); not part of the train step."""
        assert self.train
        torch.cuda.synchronize()
        for bn in self.g.bns:
            st = self.A.stats_read(bn.stats).sum(0)
            mean = st[0] / bn.count
            var = (st[1] / bn.count - mean * mean).clamp_min(0) * (bn.count / max(bn.count - 1, 1))
            self.A.view(bn.rmean).copy_(mean.float())
            self.A.view(bn.rvar).copy_(var.float())

    def output_view(self, i):
        return self.A.view(self.g.outputs[i].buf)          # NHWC in act dtype

    def out_grad_view(self, i):
        return self.A.view(self.g.out_grads[i].buf)


class FusedFPDStep:
    """The fused FPD iteration (lib/core/function.py:119-147 of the reference) on one GPU:
         teacher forward (eval BN) -> student forward (train BN) -> fused pose+KD loss fwd/bwd
         -> student backward -> [RCCL all-reduce of the flat gradient] -> flat Adam.
    No host synchronisation inside; losses are read back only when asked for."""

    def __init__(self, student_state, student_cfg, teacher_state, teacher_cfg, batch, height, width, alpha,
                 lr=2.5e-4, betas=(0.9, 0.999), eps=1e-8, world_size=1, adam=None, teacher_chunks=None,
                 use_target_weight=(True, True)):
        dev = student_state.device
        # JointsMSELoss(use_target_weight) of the pose / distillation criterion (tools/fpd_train.py:145-147,177-179):
        # False = that term ignores the loader's target_weight (loss.py:30-37), i.e. a weight buffer of ones
        self.use_w = (bool(use_target_weight[0]), bool(use_target_weight[1]))
        self.dtype = student_state.dtype
        self.alpha, self.world_size = alpha, world_size
        self.B, self.J = batch, student_cfg['J']
        # teacher first: it owns the image buffer; its last-stack map is read in place by the loss kernel
        self.teacher = None
        self.teachers = []
        self.tmap = [None, None]
        if teacher_state is not None:
            assert teacher_state.dtype == self.dtype
            # The frozen teacher normalises with running statistics, so its samples are independent: the batch CAN be cut
            # into chunks that run as separate op chains on separate streams (constructor argument teacher_chunks).
            # Measured on MI355X (r01): 1 chunk 15.0 ms/step, 2 chunks 15.0, 4 chunks 17.7 -- beyond three concurrent
            # streams (teacher, student chain, weight-gradient lane) the step gets slower, so the default is 1.
            if teacher_chunks is None:
                teacher_chunks = 1
            while batch % teacher_chunks:
                teacher_chunks -= 1
            cb = batch // teacher_chunks
            self.teacher = GraphInstance(teacher_state, teacher_cfg, cb, height, width, train=False).finalize()
            self.teachers = [self.teacher] + [
                GraphInstance(teacher_state, teacher_cfg, cb, height, width, train=False,
                              share_weights_with=self.teacher).finalize() for _ in range(1, teacher_chunks)]
            self.teacher.run('prep')           # frozen weights: working copies are prepared once, shared by the chunks
            nt = self.teacher.g.outputs[-1].numel * teacher_chunks
            # the teacher runs one batch ahead on its own streams: its last-stack map is staged in two slots
            self.tmap = [torch.zeros(nt, dtype=act_torch_dtype(self.dtype), device=dev) for _ in range(2)]
            self.t_streams = [torch.cuda.Stream(device=dev) for _ in range(teacher_chunks)]
            self.t_stream = self.t_streams[0]
            self.ev_t = [torch.cuda.Event(), torch.cuda.Event()]
            self.ev_chunk = [torch.cuda.Event() for _ in range(teacher_chunks)]
        self._k_t = self._k_s = 0
        # 'loss' (default): the student step waits for the teacher's map in front of the fused loss; 'start': before its forward
        self._late_teacher_wait = True         # (round 3: 10.62 -> 10.51 ms; the FPD_TEACHER_WAIT=start spelling of the old order is gone)
        self.student = GraphInstance(student_state, student_cfg, batch, height, width, train=True)
        g = self.student.g
        self.hh, self.hw = g.outputs[0].shape[1:3]
        A = self.student.A
        A.alloc('target', self.B * self.J * self.hh * self.hw)
        A.alloc('weight', self.B * self.J)
        if self.use_w[0] != self.use_w[1]:
            A.t['weight_kd'] = torch.ones(self.B * self.J, dtype=torch.float32, device=dev)
        if not self.use_w[0]:
            A.tensor('weight').fill_(1.0)
        A.alloc('losses', 4)
        self.student.mid_ops = [G.Op('loss', extra_in=list(g.outputs), extra_out=list(g.out_grads))]
        self.student.mid_native = [lambda plan: self._add_loss(plan, 0)]
        self.student.finalize()
        b = len(self.student.plan)                 # second loss op reading the other staged teacher map
        self._add_loss(self.student.plan, 1)
        self.student.rng['mid1'] = (b, len(self.student.plan))
        # optimizer state (torch.optim.Adam semantics, lib/utils/utils.py:69-73)
        n = student_state.table.sizes['param']
        self.n_param = n
        if adam is not None:        # share moments / step / lr with a lib.utils.utils.FusedAdam (checkpointable state)
            self.m, self.v, self.lr_dev, self.step_dev = adam.m, adam.v, adam.lr_dev, adam.step_dev
            betas, eps = adam.param_groups[0]['betas'], adam.param_groups[0]['eps']
        else:
            self.m = torch.zeros(n, dtype=torch.float32, device=dev)
            self.v = torch.zeros(n, dtype=torch.float32, device=dev)
            self.lr_dev = torch.full((1,), lr, dtype=torch.float32, device=dev)
            self.step_dev = torch.zeros(1, dtype=torch.int64, device=dev)
        self.betas, self.eps = betas, eps
        a = R.AdamT()
        a.n = n
        pa = student_state.A
        a.param, a.grad = pa.tensor('param').data_ptr(), pa.tensor('grad').data_ptr()
        a.m, a.v = self.m.data_ptr(), self.v.data_ptr()
        a.param_lp = None
        a.lr, a.beta1, a.beta2, a.eps = lr, betas[0], betas[1], eps
        a.bias_corr1 = a.bias_corr2 = 1.0
        a.grad_scale = 1.0        # the loss kernel already folds 1/world_size into the gradient
        a.lr_dev, a.step_dev = self.lr_dev.data_ptr(), self.step_dev.data_ptr()
        self._adam_args = a
        b = len(self.student.plan)
        self.student.plan.add(R.OP_ADAM, a)
        self.student.rng['adam'] = (b, len(self.student.plan))
        self._dist_work = None

    def enable_metric(self, min_slots=0):
        """Per-iteration PCK of the last student map against the target, on the device (lib.core.evaluate.DeviceAccuracy).
        `min_slots`: the number of iterations the caller may leave between two drain() calls (the ring is sized to at
        least that; an existing smaller ring is drained by the caller first and replaced)."""
        m = getattr(self, 'metric', None)
        if m is None or m.slots < min_slots:
            assert m is None or m.pending() == 0, 'drain() the metric ring before enlarging it'
            from .lib.core.evaluate import DeviceAccuracy
            g, A = self.student.g, self.student.A
            self.metric = DeviceAccuracy(self.B, self.J, self.hh, self.hw, self.dtype, self.student.state.device,
                                         slots=max(4096, int(min_slots))).bind(
                A.ptr(g.outputs[-1].buf), A.tensor('target').data_ptr(), A.tensor('losses').data_ptr())
        return self.metric

    def _add_loss(self, plan, slot):
        A, g = self.student.A, self.student.g
        plan.add(*self.student.low.memset('losses'))
        s = R.LossT()
        s.B, s.J, s.H, s.W, s.S, s.dtype = self.B, self.J, self.hh, self.hw, len(g.outputs), self.dtype
        s.target_nchw, s.alpha = 1, self.alpha
        for i, (o, d) in enumerate(zip(g.outputs, g.out_grads)):
            s.out[i] = A.ptr(o.buf)
            s.dout[i] = A.ptr(d.buf)
        if self.teacher is not None:
            s.teacher = self.tmap[slot].data_ptr()
        else:                                   # plain (non-KD) training: alpha must be 0, kd term reads the student map
            s.teacher = A.ptr(g.outputs[-1].buf)
        s.target, s.weight = A.tensor('target').data_ptr(), A.tensor('weight').data_ptr()
        s.weight_kd = A.tensor('weight_kd').data_ptr() if 'weight_kd' in A.t else None
        s.losses = A.tensor('losses').data_ptr()
        s.grad_scale = 1.0 / self.world_size
        plan.add(R.OP_LOSS, s)

    # ---- data ----
    def set_batch(self, inp, target, target_weight):
        """Copy one loader batch (any device) into the student's fixed HBM buffers (async on the current stream)."""
        self.student.image().copy_(inp, non_blocking=True)
        self.student.A.tensor('target').view(target.shape).copy_(target, non_blocking=True)
        if self.use_w[0]:
            self.student.A.tensor('weight').view(target_weight.shape).copy_(target_weight, non_blocking=True)
        elif self.use_w[1]:
            self.student.A.tensor('weight_kd').view(target_weight.shape).copy_(target_weight, non_blocking=True)
        self._last_inp = inp

    # ---- one iteration = teacher_async(batch) + student_step(batch) ----
    def teacher_async(self, inp=None):
        """Frozen-teacher forward of one batch on the teacher stream.  It may be submitted one batch ahead of the
        student step that consumes it: it then overlaps the previous batch's student forward/backward/Adam (it does
        not depend on the student weights).  `inp` None = reuse the image already in the teacher's buffer."""
        if self.teacher is None:
            return
        slot = self._k_t % 2
        cur = torch.cuda.current_stream()
        n = len(self.teachers)
        for k in reversed(range(n)):                    # chunk 0's stream collects the others at the end
            t, T = self.teachers[k], self.t_streams[k]
            T.wait_stream(cur)                          # inputs staged on the caller's stream; slot free (its loss ran)
            with torch.cuda.stream(T):
                cb = t.g.N
                if inp is not None:
                    t.image().copy_(inp[k * cb:(k + 1) * cb], non_blocking=True)
                t.run('fwd')
                o = t.g.outputs[-1].buf
                self.tmap[slot][k * o.numel:(k + 1) * o.numel].copy_(t.A.tensor('act')[o.off:o.off + o.numel])
                if k:
                    self.ev_chunk[k].record(T)
                else:
                    for j in range(1, n):
                        T.wait_event(self.ev_chunk[j])
                    self.ev_t[slot].record(T)
        self._k_t += 1

    def student_step(self, allreduce=None):
        """Student prep/forward, fused loss (against the staged teacher map), backward, [all-reduce], Adam."""
        s = self.student
        slot = self._k_s % 2
        late = self._late_teacher_wait
        if self.teacher is not None:
            assert self._k_t > self._k_s, 'teacher_async() must be submitted before student_step()'
            if not late:
                torch.cuda.current_stream().wait_event(self.ev_t[slot])
        if self._dist_work is not None:        # previous step's gradient exchange overlapped this step's teacher forward
            self._dist_work()
            self._dist_work = None
            s.run('adam')
        s.run('prep')
        s.run('fwd')
        if self.teacher is not None and late:  # only the fused loss reads the teacher's map: the student's own forward
            torch.cuda.current_stream().wait_event(self.ev_t[slot])      # need not wait for a teacher that is still running
        s.run('mid' if slot == 0 else 'mid1')
        if getattr(self, 'metric', None) is not None:
            self.metric.enqueue()
        s.run('bwd')
        if allreduce is not None:
            self._dist_work = allreduce(self.student.state.A.tensor('grad'), self.grad_buckets())
        else:
            s.run('adam')
        self._k_s += 1

    def grad_buckets(self):
        """[(first element, end element, wait)] of the flat gradient arena in completion order: `wait(stream)` makes a
        torch stream wait (on the device, no host sync) for the point of the just-enqueued backward where that slice is
        final, so its all-reduce can be issued on another stream while the rest of the backward runs.  One bucket (the
        whole arena, no wait needed: it is issued behind the backward) when the phases are replayed as hipGraphs."""
        s = self.student
        n = self.n_param
        if s.graphs.get('bwd') is not None or not getattr(s, 'bucket_ops', None):
            return [(0, n, None)]
        table = s.state.table
        out = []
        for b, (lo, hi) in enumerate(table.buckets):
            op = s.bucket_ops.get(b)
            if hi <= lo:
                continue
            assert op is not None, 'gradient bucket %d [%d, %d) has no completion op: it would never be all-reduced' % (b, lo, hi)
            out.append((lo, hi, (lambda stream, op=op: s.plan.wait_op(op, C.c_void_p(stream.cuda_stream)))))
        cover = sorted((lo, hi) for lo, hi, _ in out)
        assert cover and cover[0][0] == 0 and cover[-1][1] == n and all(a[1] == b[0] for a, b in zip(cover, cover[1:])), \
            'gradient buckets %r do not tile the arena [0, %d)' % (cover, n)
        return out

    def enable_graphs(self):
        """Replay every phase as a hipGraph from now on (call after at least one eager step)."""
        cap = torch.cuda.Stream(device=self.student.state.device)
        for t in self.teachers:
            t.capture(['fwd'], cap)
        self.student.capture(['prep', 'fwd', 'mid', 'mid1', 'bwd', 'adam'], cap)

    def step(self, allreduce=None):
        """Un-pipelined iteration on the batch given to set_batch(): teacher forward, then the student step."""
        self.teacher_async(getattr(self, '_last_inp', None))
        self.student_step(allreduce)

    def run_pipelined(self, n_steps, allreduce=None):
        """n_steps iterations on the staged batch with the teacher one batch ahead (exactly n_steps teacher forwards and
        n_steps student steps are enqueued; the pipeline starts and ends empty)."""
        inp = getattr(self, '_last_inp', None)
        self.teacher_async(inp)
        for i in range(n_steps):
            if i + 1 < n_steps:
                self.teacher_async(None)
            self.student_step(allreduce)
        self.flush()

    def flush(self):
        """Apply a pending (overlapped) optimizer update -- call after the last step()."""
        if self._dist_work is not None:
            self._dist_work()
            self._dist_work = None
            self.student.run('adam')

    def launches_per_step(self):
        """Plan ops (= kernel launches / memsets; no-ops excluded) one iteration enqueues, per phase."""
        def count(inst, name):
            b, e = inst.rng[name]
            return sum(1 for k in range(b, e) if inst.plan.op_type(k) != R.OP_NOP)
        out = {'teacher_fwd': count(self.teacher, 'fwd') * len(self.teachers)} if self.teacher is not None else {}
        for ph in ('prep', 'fwd', 'mid', 'bwd', 'adam'):
            out['student_' + ph] = count(self.student, ph)
        out['total'] = sum(out.values())
        return out

    def losses(self):
        """(pose, kd, total) of the last step -- synchronises."""
        l = self.student.A.tensor('losses')[:2].cpu()
        pose, kd = float(l[0]), float(l[1])
        return pose, kd, (1 - self.alpha) * pose + self.alpha * kd

    def set_lr(self, lr):
        self.lr_dev.fill_(lr)
