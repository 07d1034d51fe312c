"""Host-side graph of the FPD hot path: op list (IR), reverse-mode construction, memory planning.

Pure Python, no torch/GPU dependency: the same IR is lowered to the C ABI (executor.py) for the
MI355X, and interpreted on CPU by the test oracle (oracle/plan_interp.py) to check this host logic.

What is mirrored from the reference (citations into /root/reference):
  * lib/models/hourglass.py:32-52   Bottleneck   -> HourglassGraph.bottleneck
  * lib/models/hourglass.py:80-92   Hourglass    -> HourglassGraph.hour_glass
  * lib/models/hourglass.py:170-192 HourglassNet -> HourglassGraph.build_forward
  * autograd of all of the above                 -> HourglassGraph.build_backward
Design (MI355X-first): every BatchNorm+ReLU is folded into the *consumer* conv's
operand load; every producer accumulates the batch statistics the next BN needs in its epilogue;
activations are NHWC; tensors are placed in one arena by liveness over the whole fwd+bwd list.
"""
import os
from collections import OrderedDict

BN_EPS = 1e-5        # torch default, hourglass.py:18
BN_MOMENTUM = 0.1    # hourglass.py:10
STATS_REPLICAS = int(os.environ.get('FPD_STATS_REPLICAS', '4'))   # include/fpd_amd.h FPD_STATS_REPLICAS: statistics buffers are
# [R][2][C]; the environment override exists for same-box A/B runs against a library built with -DFPD_STATS_REPLICAS=n
# Multi-lane execution (see schedule.py).  Measured on MI355X/ROCm 7.2: a same-stream kernel->kernel dependency costs
# ~0.9 us, a cross-stream one (event record + wait) ~10 us, so lanes are coarse: only hourglass up-branches of the
# LANE_LEVELS largest resolutions get a lane, and weight gradients are issued in batches of WGRAD_BATCH on one lane.
WGRAD_LANES = 1
LANE_LEVELS = 0
FOLD_APPLY = os.environ.get('FPD_FOLD_APPLY', '1') != '0'     # BN-backward applies evaluated by the consuming data gradient where the library offers it
# FPD_WREDUCE_MODE: where the slabs of a gradient bucket are summed on the weight-gradient lane.
#   bucket  one reduction at the end of the bucket (round 3): the LAST bucket's reduction (130 us over 116 tensors) sits between the
#           last weight gradient and Adam, fully exposed (profiles/r04a trace: the chain ends 435 us before Adam starts)
#   batch   one behind every batch
#   lag     one IN FRONT of every batch, covering the batches before it: the lane sums finished slabs while it waits for the
#           next batch's operands anyway, and what remains behind the last weight gradient covers the last batch only
WREDUCE_MODE = os.environ.get('FPD_WREDUCE_MODE', 'lag')
WGRAD_BATCH = 8     # re-swept in round 2 on one box: 1/2/4/8/12/16/24/32 -> 11.92/11.83/11.69/11.67/11.81/11.83/11.90/11.99 ms (the lane tail before Adam)


class Buf:
    """A region of a named flat arena (offset/size in elements)."""
    __slots__ = ('arena', 'off', 'shape', 'name')

    def __init__(self, arena, off, shape, name=''):
        self.arena, self.off, self.shape, self.name = arena, off, tuple(shape), name

    @property
    def numel(self):
        n = 1
        for s in self.shape:
            n *= s
        return n

    def __repr__(self):
        return 'Buf(%s+%s %s %s)' % (self.arena, self.off, self.shape, self.name)


class Act:
    """NHWC activation (or activation-gradient) tensor; `buf` is assigned by plan_memory()."""
    __slots__ = ('shape', 'buf', 'stats', 'producer', 'grad', 'name', 'persistent', 'needs_grad', 'apply_op')

    def __init__(self, shape, name=''):
        self.shape = tuple(shape)      # (N,H,W,C)
        self.buf = None
        self.stats = None              # Buf in 'stats' arena: [2*C] fp64 {sum, sumsq}
        self.producer = None
        self.grad = None               # Act
        self.name = name
        self.persistent = False
        self.needs_grad = True
        self.apply_op = None           # gradient tensors: the BN-backward apply whose sole result this is (or None)

    @property
    def numel(self):
        n, h, w, c = self.shape
        return n * h * w * c


class BN:
    __slots__ = ('name', 'mode', 'relu', 'C', 'gamma', 'beta', 'rmean', 'rvar', 'nbt', 'stats', 'count')

    def __init__(self, name, mode, C, gamma, beta, rmean, rvar, nbt, stats=None, relu=True):
        self.name, self.mode, self.C, self.relu = name, mode, C, relu
        self.gamma, self.beta, self.rmean, self.rvar, self.nbt, self.stats = gamma, beta, rmean, rvar, nbt, stats
        self.count = 0                 # N*H*W of the normalised tensor


def f8_conv_domain(dims):
    """Shapes csrc/conv_tile_f8.hip takes (fpd_conv_f8_in_domain): stride-1 "same" 1x1 / 3x3, rows <= 128 pixels,
    C % 32 == 0, K % 8 == 0, and a halo that fits the kernel's staging registers."""
    n, h, w, C, K, R, S, stride, pad, P, Q = dims
    if stride != 1 or R != S or R not in (1, 3) or pad != (R - 1) // 2 or w > 128 or w < 2:
        return False
    if C % 32 or K % 8 or C > 512:
        return False
    bk = 64 if C % 64 == 0 else 32
    return (max(1, 128 // w) + R - 1) * w * (bk // 8) <= 2048


class Op:
    """kind in {conv, wgrad, stem_fwd, stem_wgrad, ew, loss, adam, memset, wprep, bnupd}; free-form fields."""

    def __init__(self, kind, **kw):
        self.kind = kind
        self.lane = None               # execution lane (HIP stream) -- stamped by the builder's op lists
        self.__dict__.update(kw)

    def accesses(self):
        """(reads, writes) as lists of Buf -- what the dependency analysis of the multi-lane schedule sees.
        None = unknown / whole-arena access: the op is scheduled as a barrier across all lanes."""
        def b(a):
            return a.buf if isinstance(a, Act) else a

        def bn_bufs(bn):
            if bn is None:
                return []
            return [bn.gamma, bn.beta] + ([bn.stats] if bn.mode == 'train' else [bn.rmean, bn.rvar])
        k = self.kind
        if k in ('conv2', 'bneck2', 'ew2'):    # two independent convs / fused Bottlenecks / elementwise ops, one launch
            (ra, wa), (rb, wb) = self.a.accesses(), self.b.accesses()
            return ra + rb, wa + wb
        if k == 'conv':
            rd = [b(self.x), self.w, self.bias, b(self.residual), b(self.epi_x)] + bn_bufs(self.bn) + bn_bufs(self.epi_bn)
            rd += [getattr(self, 'w8', None), getattr(self, 'w8s', None)]
            wr = [b(self.y), self.out_stats, self.epi_stats]
            fw = getattr(self, 'fused_wgrad', None)        # this data gradient also writes the slabs of that weight gradient
            if fw is not None and getattr(self, 'fused_active', False):
                wr += [fw.dw, fw.dbias]
            fa = getattr(self, 'fold_apply', None)         # it may evaluate that BN-backward apply itself (executor decides,
            if fa is not None and getattr(self, 'fold_active', False):     # before the schedule is built): then it reads the
                # apply's inputs and writes its outputs
                rd += [b(fa.x), b(fa.dy), fa.bstats] + bn_bufs(fa.bn)
                wr += [b(fa.y), fa.dgamma, fa.dbeta]
        elif k == 'head':
            rd = [b(self.y0), b(self.x), self.w_fc, self.b_fc, self.w_score, self.b_score, self.w_fc2, self.b_fc2,
                  self.w_score2, self.b_score2] + bn_bufs(self.bn)
            wr = [b(self.score), b(self.next)]
        elif k == 'bneck':
            rd = [b(self.x), self.w1, self.b1, self.w2, self.b2, self.w3, self.b3] + bn_bufs(self.bn1) + \
                bn_bufs(self.bn2) + bn_bufs(self.bn3)
            wr = [b(self.y)]
        elif k == 'wgrad':
            rd = [b(self.x), b(self.dy)] + bn_bufs(self.bn)
            wr = [self.dw, self.dbias]
        elif k == 'affsum':                    # y = relu?(sum_j bn_j(up_j(x_j))) -- HRNet block tails and fuse layers
            rd = [b(t) for t, _, _ in self.terms]
            for _, bn, _ in self.terms:
                rd += bn_bufs(bn)
            wr = [b(self.y), self.out_stats]
        elif k == 'nchw2nhwc':
            rd, wr = [self.image], [b(self.y)]
        elif k == 'wreduce':                   # sums the slabs of its weight gradients into dw / dbias
            rd, wr = list(self.bufs), list(self.bufs)
        elif k == 'grad_ready':                # marker: every gradient of one bucket is final once this op has run
            rd, wr = [self.region], []
        elif k == 'stem_fwd':
            rd, wr = [self.image, self.w, self.bias], [b(self.y), self.out_stats]
        elif k == 'stem_wgrad':
            rd, wr = [self.image, b(self.dy)], [self.dw, self.dbias]
        elif k == 'ew':
            rd = [b(self.x), b(self.x2), b(self.dy), b(self.add)] + bn_bufs(self.bn)
            wr = [b(self.y), self.out_stats, self.bstats, self.dgamma, self.dbeta]    # bstats: produced or consumed
        else:
            return None
        return [x for x in rd if x is not None], [x for x in wr if x is not None]

    def acts_in(self):
        if self.kind in ('conv2', 'bneck2', 'ew2'):
            return self.a.acts_in() + self.b.acts_in()
        if self.kind == 'head':
            return [t for t in (self.y0, self.x) if t is not None]
        fa = getattr(self, 'fold_apply', None)
        folded = [t for t in (fa.x, fa.dy) if isinstance(t, Act)] if fa is not None else []
        return [getattr(self, f) for f in ('x', 'x2', 'dy', 'add', 'residual', 'epi_x') if
                isinstance(getattr(self, f, None), Act)] + [a for a in getattr(self, 'extra_in', []) if a is not None] + folded

    def acts_out(self):
        if self.kind in ('conv2', 'bneck2', 'ew2'):
            return self.a.acts_out() + self.b.acts_out()
        if self.kind == 'head':
            return [t for t in (self.score, self.next) if t is not None]
        return [getattr(self, f) for f in ('y',) if isinstance(getattr(self, f, None), Act)] + \
               [a for a in getattr(self, 'extra_out', []) if a is not None]


# ------------------------------------------------------------------------------------------------
# parameters
# ------------------------------------------------------------------------------------------------
class ParamTable:
    """Flat layout of a model's state_dict.  'param' arena: trainable tensors (conv weights stored
    K,R,S,C = the OIHW tensor in channels-last memory order; biases; BN affine) in state_dict order;
    'rstat': BN running mean/var; 'nbt': num_batches_tracked."""

    def __init__(self, keys, bucket_of=None):
        """`bucket_of(key) -> int` (optional) groups the trainable tensors into gradient buckets: bucket 0 is the one
        whose gradients are complete FIRST in a backward pass.  The 'param' (and with it the 'grad') arena is laid out
        bucket by bucket, so that each bucket is ONE contiguous slice -- one RCCL all-reduce issued while the rest of
        the backward still runs.  state_dict order (self.keys / self.entries) is unaffected."""
        self.keys = list(keys)                     # [(key, logical_shape)]
        self.entries = OrderedDict()               # key -> Buf
        self.logical = dict(self.keys)
        self.sizes = {'param': 0, 'rstat': 0, 'nbt': 0}
        self.bucket = {k: (bucket_of(k) if bucket_of else 0) for k, _ in self.keys}
        placed = {}
        self.buckets = []                          # [(first element, end element)] of the param/grad arena per bucket
        for k, shp in sorted(self.keys, key=lambda ks: self.bucket[ks[0]]):   # stable: key order within a bucket
            if k.endswith('num_batches_tracked'):
                arena = 'nbt'
            elif k.endswith('running_mean') or k.endswith('running_var'):
                arena = 'rstat'
            else:
                arena = 'param'
            n = 1
            for s in shp:
                n *= s
            phys = (shp[0], shp[2], shp[3], shp[1]) if len(shp) == 4 else tuple(shp)
            off = self.sizes[arena]
            placed[k] = Buf(arena, off, phys, k)
            # keep every tensor 16-byte aligned in its arena
            step = 4 if arena != 'nbt' else 2
            self.sizes[arena] = off + (n + step - 1) // step * step
            if arena == 'param':
                b = self.bucket[k]
                while len(self.buckets) <= b:
                    self.buckets.append([off, off])
                self.buckets[b][1] = self.sizes[arena]
        for k, _ in self.keys:
            self.entries[k] = placed[k]
        self.buckets = [tuple(b) for b in self.buckets]

    def __getitem__(self, k):
        return self.entries[k]

    def grad(self, k):
        b = self.entries[k]
        return Buf('grad', b.off, b.shape, 'grad:' + k)

    def conv_keys(self):
        return [k for k, shp in self.keys if len(shp) == 4]

    def trainable_keys(self):
        return [k for k, b in self.entries.items() if b.arena == 'param']

    def grad_bucket(self, b):
        """The gradient arena slice of bucket b as a Buf (dependency analysis of the 'grad_ready' ops)."""
        lo, hi = self.buckets[b]
        return Buf('grad', lo, (hi - lo,), 'grad:bucket%d' % b)


def hourglass_bucket_of(num_stacks):
    """Gradient buckets of a stacked hourglass: one per stack, in the order their gradients complete in a backward pass
    (last stack first); the stem (conv1, bn1, layer1-3) finishes last and joins stack 0's bucket."""
    def f(key):
        p = key.split('.')
        if p[0] in ('hg', 'res', 'fc', 'score', 'fc_', 'score_'):
            return num_stacks - 1 - int(p[1])
        return num_stacks - 1
    return f


# ------------------------------------------------------------------------------------------------
# memory planning
# ------------------------------------------------------------------------------------------------
def plan_memory(ops, align=64, reuse_delay=0):
    """Assign every Act reachable from `ops` an offset in the 'act' arena by liveness: a tensor is
    allocated at its first appearance and released after the last op that mentions it (persistent
    tensors never).  Outputs of an op are placed before its inputs are released, so an op never
    overwrites its own inputs unless the IR aliases them on purpose.  `reuse_delay` keeps a released
    block out of circulation for that many further ops: with several lanes in flight an immediate
    reuse would chain the new tensor's writer behind the old tensor's last reader on another lane
    (a false write-after-read dependency); HBM is plentiful, latency is not.  Returns the arena size."""
    last = {}
    for i, op in enumerate(ops):
        for a in op.acts_in() + op.acts_out():
            last[id(a)] = i
    free = []      # list of (off, size)
    limbo = []     # (release index, off, size) not yet reusable
    top = 0

    def alloc(n):
        nonlocal top
        n = (n + align - 1) // align * align
        best = None
        for j, (o, s) in enumerate(free):
            if s >= n and (best is None or s < free[best][1]):
                best = j
        if best is not None:
            o, s = free.pop(best)
            if s > n:
                free.append((o + n, s - n))
            return o, n
        o = top
        top += n
        return o, n

    def release(o, n):
        free.append((o, n))
        free.sort()
        merged = []
        for o2, s2 in free:
            if merged and merged[-1][0] + merged[-1][1] == o2:
                merged[-1] = (merged[-1][0], merged[-1][1] + s2)
            else:
                merged.append((o2, s2))
        free[:] = merged

    sizes = {}
    for i, op in enumerate(ops):
        while limbo and limbo[0][0] + reuse_delay <= i:
            _, o, n = limbo.pop(0)
            release(o, n)
        for a in op.acts_in() + op.acts_out():
            if a.buf is None:
                o, n = alloc(a.numel)
                a.buf = Buf('act', o, a.shape, a.name)
                sizes[id(a)] = n
        seen = set()
        for a in op.acts_in() + op.acts_out():
            if id(a) in seen:
                continue
            seen.add(id(a))
            if last[id(a)] == i and not a.persistent and id(a) in sizes:
                limbo.append((i, a.buf.off, sizes.pop(id(a))))
    return top


class _OpList(list):
    """Op list that stamps the builder's current lane on every op appended without one."""

    def __init__(self, owner):
        super().__init__()
        self.owner = owner

    def append(self, op):
        if op.lane is None:
            op.lane = self.owner._lane
        super().append(op)


# ------------------------------------------------------------------------------------------------
# hourglass graph
# ------------------------------------------------------------------------------------------------
class HourglassGraph:
    """Op lists for one (model, batch shape, train|eval) instance."""

    def __init__(self, params, num_feats, num_stacks, num_joints, batch, height, width, train, num_blocks=1,
                 depth=4, wlp_is_master=True, lane_levels=None, wgrad_batch=None, fuse_bneck=False, pair_branches=True, fuse_head=True):
        self.p = params
        self.F, self.S, self.J = num_feats, num_stacks, num_joints
        self.N, self.H, self.W = batch, height, width
        self.train, self.num_blocks, self.depth = train, num_blocks, depth
        self.wlp_is_master = wlp_is_master     # fp32 build: forward convs read the master weights directly
        self.fuse_bneck = fuse_bneck and not train   # frozen bf16 networks: whole Bottleneck in one launch
        self.fuse_head = fuse_head and self.fuse_bneck
        # The up-branch and the low-branch bottleneck of an hourglass level are independent and have the same channel
        # shapes: their convolutions (and their data gradients) are issued pairwise as ONE launch ('conv2' ops), which
        # takes ~100 launches off the latency-bound critical chain of a training step.
        self.pair_branches = pair_branches
        self._pair_op = None
        self._pair_ew = None
        self.stats_size = 0
        self.fold_size = 0                     # folded BN tables of fused Bottlenecks ('fold' arena, fp32)
        self.wlp_size = 0
        self.wfwd, self.wbwd = {}, {}
        self._lane = 0                         # lane 0 = the caller's stream; 1..depth = hourglass up-branches
        self.lane_levels = LANE_LEVELS if lane_levels is None else lane_levels
        self.wgrad_batch = WGRAD_BATCH if wgrad_batch is None else wgrad_batch
        self._setup()

    MASTER_ONLY = ('conv1.weight',)            # convolutions whose kernels read the fp32 master weights (hourglass stem)

    def _setup(self):
        """Shared tail of the constructors: op lists, working-weight copies, forward and backward construction."""
        self._wg_pending = []
        self.n_lanes = 1 + self.depth + WGRAD_LANES
        self.fwd, self.bwd = _OpList(self), _OpList(self)
        self.bns = []                          # train-mode BNs in forward order (running-stat update table)
        self.image = Buf('image', 0, (self.N, 3, self.H, self.W), 'image')
        self.outputs = []
        self._bn_pending = {}
        self._bn_uses = {}
        for k in self.p.conv_keys():
            if k in self.MASTER_ONLY:
                continue
            b = self.p[k]
            if not self.wlp_is_master:
                self.wfwd[k] = self._wlp(b)
            if self.train:
                K, R, S, C = b.shape
                self.wbwd[k] = self._wlp(Buf('param', 0, (C, R, S, K)))
        self.build_forward()
        if self.train:
            self.build_backward()

    # ---- small allocators ----
    def _w8(self, wkey):
        """e4m3 copy + per-output-channel scales of convolution weight `wkey` ('w8' arena: bytes, 'w8s': fp32)."""
        tab = self.__dict__.setdefault('w8', {})
        if wkey not in tab:
            b = self.p[wkey]
            q = Buf('w8', getattr(self, 'w8_size', 0), b.shape, 'w8:' + wkey)
            self.w8_size = q.off + (b.numel + 15) // 16 * 16
            sc = Buf('w8s', getattr(self, 'w8s_size', 0), (b.shape[0],), 'w8s:' + wkey)
            self.w8s_size = sc.off + (b.shape[0] + 3) // 4 * 4
            tab[wkey] = (q, sc)
        return tab[wkey]

    def _wlp(self, like):
        off = self.wlp_size
        self.wlp_size += (like.numel + 7) // 8 * 8
        return Buf('wlp', off, like.shape, 'wlp')

    def _stats(self, C, name=''):
        off = self.stats_size
        self.stats_size += STATS_REPLICAS * 2 * C
        return Buf('stats', off, (STATS_REPLICAS, 2, C), name)

    def _bn(self, name, C):
        g = self.p
        mode = 'train' if self.train else 'eval'
        return BN(name, mode, C, g[name + '.weight'], g[name + '.bias'], g[name + '.running_mean'],
                  g[name + '.running_var'], g[name + '.num_batches_tracked'])

    # ---- forward primitives ----
    def conv(self, x, name, bn=None, residual=None, pad=0, sink=None, stride=1):
        wkey = name + '.weight'
        K, R, S, C = self.p[wkey].shape
        n, h, w, c = x.shape
        assert c == C, (name, x.shape, self.p[wkey].shape)
        y = Act((n, (h + 2 * pad - R) // stride + 1, (w + 2 * pad - S) // stride + 1, K), name)
        wbuf = self.p[wkey] if self.wlp_is_master else self.wfwd[wkey]
        bkey = name + '.bias' if (name + '.bias') in self.p.entries else None      # HRNet convolutions carry no bias
        op = Op('conv', x=x, w=wbuf, wkey=wkey, bias=self.p[bkey] if bkey else None, bkey=bkey, residual=residual,
                y=y, out_stats=None, bn=bn, epi='plain', epi_x=None, epi_bn=None, epi_stats=None,
                dims=(n, h, w, C, K, R, S, stride, pad, y.shape[1], y.shape[2]))
        y.producer = op
        op.w8 = op.w8s = None
        if getattr(self, 'fp8', False) and sink is None and f8_conv_domain(op.dims):
            op.w8, op.w8s = self._w8(wkey)       # forward on the fp8 matrix pipe (csrc/conv_tile_f8.hip)
            op.w_master = self.p[wkey]
        if bn is not None:
            self._use_bn(x, bn)
        (self.fwd if sink is None else sink).append(op)      # sink: collected by bottleneck_pair instead of issued
        return y

    def _use_bn(self, x, bn):
        if bn.mode == 'train':
            if x.stats is None:
                x.stats = self._stats(x.shape[3], 'stats:' + x.name)
                prod = x.producer
                assert prod is not None and hasattr(prod, 'out_stats'), 'no stats producer for ' + x.name
                prod.out_stats = x.stats
            bn.stats = x.stats
            key = (id(x), bn.name)
            self._bn_uses[key] = self._bn_uses.get(key, 0) + 1
            if not any(b.name == bn.name for b in self.bns):
                self.bns.append(bn)
        bn.count = x.shape[0] * x.shape[1] * x.shape[2]
        return bn

    def ew(self, opname, shape_large, y_shape, name, **kw):
        y = Act(y_shape, name)
        op = Op('ew', op=opname, dims=tuple(shape_large), y=y, out_stats=None, x=kw.get('x'), x2=kw.get('x2'),
                dy=None, add=None, bstats=None, dgamma=None, dbeta=None, bn=kw.get('bn'))
        y.producer = op
        self.fwd.append(op)
        return y

    def maxpool(self, x, name):
        n, h, w, c = x.shape
        return self.ew('maxpool_fwd', x.shape, (n, h // 2, w // 2, c), name, x=x)

    def upadd(self, a, b, name):
        assert a.shape[1] == 2 * b.shape[1] and a.shape[2] == 2 * b.shape[2] and a.shape[3] == b.shape[3]
        return self.ew('upadd_fwd', a.shape, a.shape, name, x=a, x2=b)

    def bottleneck(self, x, p, sink=None):
        """hourglass.py:32-52 with bn_k+relu folded into conv_k's operand load."""
        c_in = x.shape[3]
        planes = self.p[p + 'conv1.weight'].shape[0]
        if self._fused(x, p):
            n, h, w, _ = x.shape
            y = Act(x.shape, p + 'out')
            wb = (lambda k: self.p[k]) if self.wlp_is_master else (lambda k: self.wfwd[k])
            fold = Buf('fold', self.fold_size, (3 * c_in + 4 * planes,), 'fold:' + p)
            self.fold_size += 3 * c_in + 4 * planes
            op = Op('bneck', x=x, y=y, dims=(n, h, w, c_in, planes), folded=fold,
                    w1=wb(p + 'conv1.weight'), b1=self.p[p + 'conv1.bias'], w2=wb(p + 'conv2.weight'),
                    b2=self.p[p + 'conv2.bias'], w3=wb(p + 'conv3.weight'), b3=self.p[p + 'conv3.bias'],
                    bn1=self._bn(p + 'bn1', c_in), bn2=self._bn(p + 'bn2', planes), bn3=self._bn(p + 'bn3', planes))
            y.producer = op
            (self.fwd if sink is None else sink).append(op)
            return y
        t = self.conv(x, p + 'conv1', bn=self._bn(p + 'bn1', c_in), sink=sink)
        t = self.conv(t, p + 'conv2', bn=self._bn(p + 'bn2', planes), pad=1, sink=sink)
        skip = x
        if (p + 'downsample.0.weight') in self.p.entries:
            assert sink is None
            skip = self.conv(x, p + 'downsample.0')
        return self.conv(t, p + 'conv3', bn=self._bn(p + 'bn3', planes), residual=skip, sink=sink)

    def _fused(self, x, p):
        planes = self.p[p + 'conv1.weight'].shape[0]
        return (self.fuse_bneck and self.bneck_fusable(x.shape, planes) and
                (p + 'downsample.0.weight') not in self.p.entries)

    def bottleneck_pair(self, xa, pa, xb, pb):
        """Two independent bottlenecks with equal channel shapes, convolution k of both issued as one 'conv2' op."""
        la, lb = [], []
        ya = self.bottleneck(xa, pa, sink=la)
        yb = self.bottleneck(xb, pb, sink=lb)
        assert len(la) == len(lb) and len(la) in (1, 3)
        for oa, ob in zip(la, lb):
            self.fwd.append(Op('conv2' if oa.kind == 'conv' else 'bneck2', a=oa, b=ob))
        return ya, yb

    @staticmethod
    def bneck_fusable(shape, planes):
        """Domain of the fused kernel (csrc/bneck_fused.hip, fpd_bneck_t)."""
        n, h, w, c = shape
        return (c == 2 * planes and planes in (64, 128) and 4 <= w <= 64 and w & (w - 1) == 0 and
                ((h * w) % 128 == 0 or 128 % (h * w) == 0))

    def residual_seq(self, x, p, nb):
        for b in range(nb):
            x = self.bottleneck(x, '%s%d.' % (p, b))
        return x

    def hour_glass(self, x, p, n):
        """hourglass.py:80-92."""
        q = '%s%d.' % (p, n - 1)
        nb = self.num_blocks
        # the up-branch is independent of the whole lower hourglass (a long chain of small, launch-latency-bound
        # kernels): it gets a lane of its own per level, so the two overlap on the GPU
        laned = n > self.depth - self.lane_levels
        pa, pb = q + '0.0.', q + '1.0.'
        n_, h_, w_, c_ = x.shape
        if (self.pair_branches and nb == 1 and not laned and
                self._fused(x, pa) == self._fused(Act((n_, h_ // 2, w_ // 2, c_)), pb) and     # both fused or both not
                (pa + 'downsample.0.weight') not in self.p.entries and (pb + 'downsample.0.weight') not in self.p.entries):
            low = self.maxpool(x, q + 'pool')
            up1, low = self.bottleneck_pair(x, pa, low, pb)
        else:
            outer = self._lane
            if laned:
                self._lane = n
            up1 = self.residual_seq(x, q + '0.', nb)
            self._lane = outer
            low = self.maxpool(x, q + 'pool')
            low = self.residual_seq(low, q + '1.', nb)
        if n > 1:
            low = self.hour_glass(low, p, n - 1)
        else:
            low = self.residual_seq(low, q + '3.', nb)
        low = self.residual_seq(low, q + '2.', nb)
        return self.upadd(up1, low, q + 'upadd')

    def build_forward(self):
        """hourglass.py:170-192."""
        K = self.p['conv1.weight'].shape[0]
        P, Q = (self.H + 6 - 7) // 2 + 1, (self.W + 6 - 7) // 2 + 1
        x = Act((self.N, P, Q, K), 'stem')
        op = Op('stem_fwd', image=self.image, w=self.p['conv1.weight'], bias=self.p['conv1.bias'], y=x, out_stats=None,
                dims=(self.N, self.H, self.W, K, P, Q))
        x.producer = op
        self.fwd.append(op)
        bn1 = self._bn('bn1', K)
        self._use_bn(x, bn1)
        x = self.ew('bnrelu_fwd', x.shape, x.shape, 'stem_act', x=x, bn=bn1)
        x = self.residual_seq(x, 'layer1.', 1)
        x = self.maxpool(x, 'pool1')
        x = self.residual_seq(x, 'layer2.', 1)
        x = self.residual_seq(x, 'layer3.', 1)
        ch = x.shape[3]
        for i in range(self.S):
            y = self.hour_glass(x, 'hg.%d.hg.' % i, self.depth)
            y = self.residual_seq(y, 'res.%d.' % i, self.num_blocks)
            if self.fuse_head and ch == 256 and self.J == 16:
                # frozen bf16 network: fc -> BN+ReLU -> score -> fc_ + score_ + residuals as ONE launch (head_fused.hip)
                wb = (lambda k: self.p[k]) if self.wlp_is_master else (lambda k: self.wfwd[k])
                last = i == self.S - 1
                score = Act((y.shape[0], y.shape[1], y.shape[2], self.J), 'score.%d' % i)
                score.persistent = True
                nxt = None if last else Act(x.shape, 'stack%d.out' % i)
                fold = Buf('fold', self.fold_size, (3 * ch + 32,), 'fold:head%d' % i)
                self.fold_size += 3 * ch + 32
                op = Op('head', y0=y, x=None if last else x, score=score, next=nxt, dims=(y.shape[0], y.shape[1], y.shape[2], ch, self.J),
                        w_fc=wb('fc.%d.0.weight' % i), b_fc=self.p['fc.%d.0.bias' % i],
                        w_score=wb('score.%d.weight' % i), b_score=self.p['score.%d.bias' % i],
                        w_fc2=None if last else wb('fc_.%d.weight' % i), b_fc2=None if last else self.p['fc_.%d.bias' % i],
                        w_score2=None if last else wb('score_.%d.weight' % i),
                        b_score2=None if last else self.p['score_.%d.bias' % i],
                        bn=self._bn('fc.%d.1' % i, ch), folded=fold)
                score.producer = op
                if nxt is not None:
                    nxt.producer = op
                self.fwd.append(op)
                self.outputs.append(score)
                x = nxt
                continue
            y = self.conv(y, 'fc.%d.0' % i)
            fcbn = self._bn('fc.%d.1' % i, ch)
            score = self.conv(y, 'score.%d' % i, bn=fcbn)
            score.persistent = True
            self.outputs.append(score)
            if i < self.S - 1:
                fcbn2 = self._bn('fc.%d.1' % i, ch)
                t = self.conv(y, 'fc_.%d' % i, bn=fcbn2, residual=x)
                x = self.conv(score, 'score_.%d' % i, residual=t)
        if self.train:
            self.fwd.append(Op('bnupd', bns=list(self.bns)))

    # ---- backward construction ----
    def _contribute(self, t):
        """(add_src, out) for an op that writes out = add_src + <its contribution to dL/dt>.  Gradient tensors are
        never modified once written (the sum goes to a fresh tensor): deferred readers -- the batched weight-gradient
        kernels, ops on other lanes -- always see the value the sequential semantics gave them."""
        prev = t.grad
        t.grad = Act(t.shape, 'd:' + t.name)
        return prev, t.grad

    def _contribute_identity(self, t, g):
        if t.grad is None:
            t.grad = g                      # alias: g is final and immutable
        else:
            prev = t.grad
            t.grad = Act(t.shape, 'd:' + t.name)
            self.bwd.append(Op('ew', op='add', dims=t.shape, x=prev, x2=g, y=t.grad, dy=None, add=None,
                               out_stats=None, bstats=None, dgamma=None, dbeta=None, bn=None, lane=self._home(t)))

    def _home(self, t):
        return t.producer.lane if t.producer is not None and t.producer.lane is not None else 0

    def _bn_backward_contribution(self, x, bn, make_dgrad):
        """Route a gradient through the fused BN+ReLU prologue of tensor x.  `make_dgrad(dz, add, bstats)`
        emits the op producing dz = relu'(.) * dA (accumulated onto `add`) and the sums into bstats."""
        key = (id(x), bn.name)
        pend = self._bn_pending.get(key)
        bstats = self._stats(bn.C, 'bstats:' + bn.name)
        if pend is None:
            dz = Act(x.shape, 'dz:' + bn.name)
            make_dgrad(dz, None, bstats)
            pend = {'dz': dz, 'left': self._bn_uses[key]}
            self._bn_pending[key] = pend
        else:
            make_dgrad(pend['dz'], pend['dz'], bstats)       # mask(acc + dz_prev) == mask(acc) + dz_prev
        pend['left'] -= 1
        if pend['left'] == 0:
            add, out = self._contribute(x)
            # x's gradient is consumed on the lane of x's producer: finishing it there keeps a side lane's last hop
            # off the main chain (main -> side -> main would cost two cross-stream dependencies)
            apply = Op('ew', op='bn_bwd_apply', dims=x.shape, x=x, x2=None, dy=pend['dz'], add=add, y=out,
                       out_stats=None, bstats=bstats, dgamma=self.p.grad(bn.name + '.weight'),
                       dbeta=self.p.grad(bn.name + '.bias'), bn=bn, lane=self._home(x))
            if add is None:
                out.apply_op = apply            # `out` is exactly this apply's result: its consumer may evaluate it itself
            c = self._pair_ew
            if c is None or (apply.lane or 0) != 0:
                self.bwd.append(apply)
            elif c.a is None:                   # the paired chains' BN-backward applies also go out as one launch
                c.a = apply
                self.bwd.append(c)
            else:
                assert c.b is None
                c.b = apply
            del self._bn_pending[key]

    def build_backward(self):
        self.out_grads = []
        for o in self.outputs:
            o.grad = Act(o.shape, 'd:' + o.name)
            o.grad.persistent = True
            self.out_grads.append(o.grad)
        # liveness marker: the output gradients are written (by the loss kernel or by autograd's seeds) BEFORE the
        # first backward op runs, so they must be placed here, not at their first use deep inside the backward list
        self.bwd.append(Op('seed', extra_in=list(self.outputs), extra_out=list(self.out_grads)))
        self._cur_bucket, self._wg_bucket = None, []
        for op in reversed(self.fwd):
            self._lane = op.lane                 # gradients flow on the lane of the forward op they belong to
            bk = self._op_bucket(op)
            if bk is not None and bk != self._cur_bucket:
                if self._cur_bucket is not None:
                    assert bk > self._cur_bucket, 'gradient buckets must complete in increasing order'
                    self._lane = 0
                    self._close_bucket(self._cur_bucket)
                    self._lane = op.lane
                self._cur_bucket = bk
            if len(self._wg_pending) >= self.wgrad_batch and op.lane == 0:
                self._flush_wgrads()
            if op.kind == 'conv':
                self._conv_backward(op)
            elif op.kind == 'conv2':
                # the two data-gradient convolutions go out as one launch too: the container is placed where the first
                # of them is due (both output gradients are final here; the chains below it are independent)
                c = self._pair_op = Op('conv2', a=None, b=None)
                e = self._pair_ew = Op('ew2', a=None, b=None)
                self._conv_backward(op.a)
                self._conv_backward(op.b)
                self._pair_op = self._pair_ew = None
                for cont in (c, e):
                    if cont.a is not None and cont.b is None:        # only one of the two emitted this op
                        cont.a.lane = cont.lane
                        self.bwd[self.bwd.index(cont)] = cont.a
            elif op.kind == 'stem_fwd':
                dy = op.y.grad
                sw = Op('stem_wgrad', image=self.image, dy=dy, dw=self.p.grad('conv1.weight'),
                        dbias=self.p.grad('conv1.bias'), dims=op.dims, lane=1 + self.depth)
                self._wg_pending.append(sw)      # issued with the last batch; its slabs are summed like every other weight gradient's
            elif op.kind == 'ew':
                self._ew_backward(op)
            elif op.kind == 'affsum':
                self._affsum_backward(op)
        self._lane = 0
        self._close_bucket(self._cur_bucket if self._cur_bucket is not None else 0)
        assert not self._bn_pending, 'unfinished BN backward: %r' % list(self._bn_pending)

    def _op_bucket(self, op):
        """Gradient bucket of the parameters a forward op owns (None: the op has no parameters)."""
        if op.kind in ('conv2', 'bneck2', 'ew2'):
            return self._op_bucket(op.a)
        key = getattr(op, 'wkey', None)
        if key is None and op.kind == 'stem_fwd':
            key = 'conv1.weight'
        if key is None and getattr(op, 'bn', None) is not None:
            key = op.bn.name + '.weight'
        if key is None and op.kind == 'affsum':
            names = [bn.name for _, bn, _ in op.terms if bn is not None]
            key = names[0] + '.weight' if names else None
        return None if key is None else self.p.bucket[key]

    def _close_bucket(self, b):
        """All forward ops owning parameters of bucket b have been differentiated: issue the remaining weight gradients,
        reduce the bucket's slabs and mark the point where its slice of the gradient arena is final (the data-parallel
        all-reduce of that slice can start there, overlapping the rest of the backward)."""
        self._flush_wgrads(b, reduce=True)
        self.bwd.append(Op('grad_ready', bucket=b, region=self.p.grad_bucket(b), lane=1 + self.depth))

    def _flush_wgrads(self, b=None, reduce=None):
        """Weight gradients are leaves: they are collected and issued in batches on their own lane, newest first, so
        that the batch's first kernel carries the one cross-lane wait that covers the whole batch.  The slabs of a batch
        are summed right behind it (WREDUCE_PER_BATCH; always at the end of a bucket): the reduction that remains in
        front of Adam after the last data gradient then covers the last batch only, not the whole last bucket."""
        def emit_reduce():
            if self._wg_bucket:
                bufs = [x for w in self._wg_bucket for x in (w.dw, w.dbias) if x is not None]
                self.bwd.append(Op('wreduce', bucket=self._cur_bucket if b is None else b, wgrads=list(self._wg_bucket), bufs=bufs,
                                   lane=1 + self.depth))
                self._wg_bucket = []
        if WREDUCE_MODE == 'lag' and self._wg_pending:
            emit_reduce()                        # the batches issued before this one: their slabs are complete (or about to be)
        for w in reversed(self._wg_pending):
            self.bwd.append(w)
            if w.kind in ('wgrad', 'stem_wgrad'):
                self._wg_bucket.append(w)
        self._wg_pending = []
        if (WREDUCE_MODE == 'batch') if reduce is None else reduce:
            emit_reduce()

    def _emit_dgrad(self, op):
        c = self._pair_op
        if c is None:
            self.bwd.append(op)
        elif c.a is None:
            c.a = op
            self.bwd.append(c)
        else:
            assert c.b is None
            c.b = op

    def _conv_backward(self, op):
        dy = op.y.grad
        if dy is None:
            return
        x = op.x
        self._wg_pending.append(Op('wgrad', x=x, dy=dy, dw=self.p.grad(op.wkey), dbias=self.p.grad(op.bkey) if op.bkey else None,
                                   bn=op.bn, dims=op.dims, lane=1 + self.depth))
        if op.residual is not None:
            self._contribute_identity(op.residual, dy)
        if not x.needs_grad:
            return
        n, h, w, C, K, R, S, stride, pad, P, Q = op.dims
        ddims = (n, P, Q, K, C, R, S, 1, R - 1 - pad, h, w)      # dgrad = conv of dy with the flipped IO-swapped weights
        if stride == 2:
            # stride-2 convolution (HRNet stem / transition / fuse layers, pose_hrnet.py:223-242,349-372): its data gradient
            # is the stride-1 convolution of the ZERO-DILATED output gradient (dy written to the even positions of an
            # input-sized grid) with the same flipped weights
            assert h == 2 * P and w == 2 * Q, 'stride-2 data gradient needs even input dims, got %dx%d' % (h, w)
            dil = Act((n, h, w, K), 'dil:' + op.wkey)
            self.bwd.append(Op('ew', op='dilate2', dims=(n, h, w, K), x=dy, x2=None, dy=None, add=None, y=dil,
                               out_stats=None, bstats=None, dgamma=None, dbeta=None, bn=None))
            dy, ddims = dil, (n, h, w, K, C, R, S, 1, R - 1 - pad, h, w)
        else:
            assert stride == 1
        wb = self.wbwd[op.wkey]
        if op.bn is not None:
            wg = self._wg_pending[-1]

            def make(dz, add, bstats, op=op, dy=dy, x=x):
                d = Op('conv', x=dy, w=wb, wkey=op.wkey, bias=None, bkey=None, residual=add, y=dz,
                       out_stats=None, bn=None, epi='bnrelu_bwd', epi_x=x, epi_bn=op.bn, epi_stats=bstats,
                       dims=ddims)
                # The data gradient of a 1x1 convolution reads dy and the forward operand's source anyway: where the library
                # offers it (fpd_conv_fused_wgrad_partials, asked at lowering time) the launch also forms that convolution's
                # weight gradient and the separate 'wgrad' op becomes a no-op (executor.Lowering.conv).
                # (only while that 'wgrad' is still pending: once its batch -- and the batch's slab reduction -- has been
                # issued, the slabs of a later launch would never be summed)
                if R == 1 and stride == 1 and dy is op.y.grad and any(w is wg for w in self._wg_pending):
                    d.fused_wgrad = wg
                # dy itself is the output of a BN-backward apply and nothing else reads it but this convolution's data and
                # weight gradient: the data gradient may evaluate the apply on its operand load (fpd_conv_t.fold_x) and the
                # separate launch becomes a no-op -- decided at lowering time like the fused weight gradient
                ap = getattr(op.y.grad, 'apply_op', None)
                if FOLD_APPLY and ap is not None and stride == 1 and dy is op.y.grad and (ap.lane or 0) == (self._lane or 0):      # (the lane d is about to be stamped with)
                    d.fold_apply = ap
                    d.fold_wgrad = wg
                self._emit_dgrad(d)
            self._bn_backward_contribution(x, op.bn, make)
        else:
            add, out = self._contribute(x)
            self._emit_dgrad(Op('conv', x=dy, w=wb, wkey=op.wkey, bias=None, bkey=None, residual=add, y=out,
                                out_stats=None, bn=None, epi='plain', epi_x=None, epi_bn=None, epi_stats=None, dims=ddims,
                                lane=self._home(x) if add is not None else None))

    def _affsum_backward(self, op):
        """y = relu(sum_j bn_j(up_j(x_j))): the masked gradient g = dy * (y > 0) reaches every term; an up-sampled term
        receives its f x f block sums; a normalised term goes through the BatchNorm backward (two sums, then apply)."""
        dy = op.y.grad
        if dy is None:
            return

        def ew(name, dims, **kw):
            f = dict(x=None, x2=None, dy=None, add=None, y=None, out_stats=None, bstats=None, dgamma=None, dbeta=None, bn=None)
            f.update(kw)
            o = Op('ew', op=name, dims=tuple(dims), **f)
            self.bwd.append(o)
            return o
        g = dy
        fused = None        # a full-resolution train-BN term takes the ReLU mask into its statistics pass (one launch for both)
        if op.relu:
            g = Act(op.y.shape, 'g:' + op.y.name)
            if True:      # (the ReLU mask always rides the statistics pass of a full-resolution train-BN term)
                fused = next((t for t, bn, up in op.terms if up == 1 and bn is not None and bn.mode == 'train' and t.needs_grad), None)
            if fused is None:
                ew('relu_mask', op.y.shape, x=op.y, dy=dy, y=g)
        for t, bn, up in sorted(op.terms, key=lambda tm: tm[0] is not fused):      # the producer of g first
            if not t.needs_grad:
                continue
            gj = g
            f = up
            while f > 1:                      # nearest x2^k up-sampling backward = k 2x2 block sums
                nxt = Act((gj.shape[0], gj.shape[1] // 2, gj.shape[2] // 2, gj.shape[3]), 'gs:' + t.name)
                ew('sumpool', gj.shape, x=gj, y=nxt)
                gj, f = nxt, f // 2
            if bn is None:
                self._contribute_identity(t, gj)
                continue
            if bn.mode != 'train':
                raise AssertionError('backward through an eval-mode BN term')
            bstats = self._stats(bn.C, 'bstats:' + bn.name)
            if t is fused:     # g = dy * (y > 0) written AND its two BN-backward sums, in one pass
                ew('bnrelu_bwd_r', t.shape, x=t, x2=op.y, dy=dy, y=g, bstats=bstats, bn=bn)
            else:
                ew('bnrelu_bwd_r', t.shape, x=t, dy=gj, y=None, bstats=bstats, bn=bn)      # statistics only (y = None)
            add, out = self._contribute(t)
            ew('bn_bwd_apply', t.shape, x=t, dy=gj, add=add, y=out, bstats=bstats, dgamma=self.p.grad(bn.name + '.weight'),
               dbeta=self.p.grad(bn.name + '.bias'), bn=bn)

    def _ew_backward(self, op):
        dy = op.y.grad
        if dy is None:
            return
        if op.op == 'maxpool_fwd':
            add, out = self._contribute(op.x)
            self.bwd.append(Op('ew', op='maxpool_bwd', dims=op.dims, x=op.x, x2=None, dy=dy, add=add, y=out,
                               out_stats=None, bstats=None, dgamma=None, dbeta=None, bn=None))
        elif op.op == 'upadd_fwd':
            self._contribute_identity(op.x, dy)
            add, out = self._contribute(op.x2)
            self.bwd.append(Op('ew', op='sumpool', dims=op.dims, x=dy, x2=None, dy=None, add=add, y=out,
                               out_stats=None, bstats=None, dgamma=None, dbeta=None, bn=None))
        elif op.op == 'bnrelu_fwd':
            x, bn = op.x, op.bn

            def make(dz, add, bstats, dy=dy, x=x, bn=bn):
                assert add is None
                self.bwd.append(Op('ew', op='bnrelu_bwd_r', dims=x.shape, x=x, x2=None, dy=dy, add=None, y=dz,
                                   out_stats=None, bstats=bstats, dgamma=None, dbeta=None, bn=bn))
            self._bn_backward_contribution(x, bn, make)
        else:
            raise AssertionError(op.op)


# ------------------------------------------------------------------------------------------------
# HRNet graph
# ------------------------------------------------------------------------------------------------
HRNET_EXPANSION = {'BASIC': 1, 'BOTTLENECK': 4}


def hrnet_bucket_of(key):
    """Gradient buckets of an HRNet in backward completion order: stage4 + head, stage3 (+ transition3), stage2
    (+ transition2), then layer1 / stem / transition1."""
    p = key.split('.')[0]
    return {'final_layer': 0, 'stage4': 0, 'transition3': 1, 'stage3': 1, 'transition2': 2, 'stage2': 2}.get(p, 3)


class HRNetGraph(HourglassGraph):
    """Op lists of /root/reference/lib/models/pose_hrnet.py (PoseHighResolutionNet.forward :425-460 and everything it
    calls: Bottleneck :78-98, BasicBlock :41-57, HighResolutionModule :247-265 with its fuse layers :187-242, the
    transition layers :333-372) and of its autograd, in the same IR as the hourglass.

    HRNet is post-activation: conv -> BN -> ReLU.  Where a normalised tensor has ONE consumer and that consumer is a
    convolution (inside the blocks, inside the stride-2 chains, the first stem BN), BN+ReLU are folded into the
    consumer's operand load exactly like the hourglass' pre-activation BNs; where it ends a block, a fuse sum or a
    transition (several consumers / a residual add), ONE 'affsum' op materialises y = relu(sum_j bn_j(up_j(x_j))):
    the block tail relu(bn2(conv2) + skip), the fuse layer relu(sum of identity / 1x1+BN+nearest-up / strided terms)."""

    MASTER_ONLY = ()

    def __init__(self, params, extra, num_joints, batch, height, width, train, wlp_is_master=True, wgrad_batch=None,
                 fp8=False):
        self.p = params
        self.fp8 = bool(fp8) and not wlp_is_master      # fp8 forward convolutions (bf16 storage build only)
        self.extra, self.J = extra, num_joints
        self.S = 1                                 # one heat-map
        self.N, self.H, self.W = batch, height, width
        self.train, self.depth = train, 4
        self.wlp_is_master = wlp_is_master
        self.fuse_bneck = self.fuse_head = False
        self.pair_branches = False
        self._pair_op = self._pair_ew = None
        self.stats_size = self.fold_size = self.wlp_size = 0
        self.wfwd, self.wbwd = {}, {}
        self._lane = 0
        self.lane_levels = 0
        self.wgrad_batch = WGRAD_BATCH if wgrad_batch is None else wgrad_batch
        self._setup()

    # ---- forward primitives ----
    def _bnx(self, name, C, relu):
        bn = self._bn(name, C)
        bn.relu = relu
        return bn

    def affsum(self, terms, name, relu=True):
        """terms: [(act, bn or None, up factor)]; the first term fixes the output shape."""
        t0, _, f0 = terms[0]
        n, h, w, c = t0.shape
        shape = (n, h * f0, w * f0, c)
        for t, bn, f in terms:
            assert (t.shape[0], t.shape[1] * f, t.shape[2] * f, t.shape[3]) == shape, (name, t.shape, f, shape)
            if bn is not None:
                self._use_bn(t, bn)
        y = Act(shape, name)
        op = Op('affsum', terms=list(terms), y=y, relu=relu, out_stats=None, dims=shape, extra_in=[t for t, _, _ in terms])
        y.producer = op
        self.fwd.append(op)
        return y

    def block(self, x, p, kind, stride=1):
        """pose_hrnet.py:41-57 (BASIC) / :78-98 (BOTTLENECK)."""
        planes = self.p[p + 'conv1.weight'].shape[0]
        if kind == 'BASIC':
            t = self.conv(x, p + 'conv1', pad=1, stride=stride)
            t = self.conv(t, p + 'conv2', bn=self._bnx(p + 'bn1', planes, True), pad=1)
            tail = (t, self._bnx(p + 'bn2', planes, False), 1)
        else:
            t = self.conv(x, p + 'conv1')
            t = self.conv(t, p + 'conv2', bn=self._bnx(p + 'bn1', planes, True), pad=1, stride=stride)
            t = self.conv(t, p + 'conv3', bn=self._bnx(p + 'bn2', planes, True))
            tail = (t, self._bnx(p + 'bn3', 4 * planes, False), 1)
        if (p + 'downsample.0.weight') in self.p.entries:
            d = self.conv(x, p + 'downsample.0', stride=stride)
            skip = (d, self._bnx(p + 'downsample.1', d.shape[3], False), 1)
        else:
            skip = (x, None, 1)
        return self.affsum([tail, skip], p + 'out')

    def hr_module(self, xs, p, sc, multi_scale_output):
        """pose_hrnet.py:247-265."""
        nb = len(xs)
        ys = []
        for i in range(nb):
            y = xs[i]
            for b in range(sc['NUM_BLOCKS'][i]):
                y = self.block(y, '%sbranches.%d.%d.' % (p, i, b), sc['BLOCK'])
            ys.append(y)
        if nb == 1:
            return ys
        outs = []
        for i in range(nb if multi_scale_output else 1):
            terms = []
            for j in range(nb):
                q = '%sfuse_layers.%d.%d.' % (p, i, j)
                if j == i:
                    terms.append((ys[j], None, 1))
                elif j > i:                      # :199-211: 1x1 conv + BN, nearest up-sampling by 2^(j-i)
                    t = self.conv(ys[j], q + '0')
                    terms.append((t, self._bnx(q + '1', t.shape[3], False), 2 ** (j - i)))
                else:                            # :214-241: (i-j) stride-2 3x3 convs, BN after each, ReLU except after the last
                    t, bn = ys[j], None
                    for k in range(i - j):
                        t = self.conv(t, '%s%d.0' % (q, k), bn=bn, pad=1, stride=2)
                        bn = self._bnx('%s%d.1' % (q, k), t.shape[3], k != i - j - 1)
                    terms.append((t, bn, 1))
            terms.sort(key=lambda tm: tm[2])     # a full-resolution term first: it fixes the output shape
            outs.append(self.affsum(terms, '%sfuse%d' % (p, i)))
        return outs

    def transition(self, ys, p, pre, cur):
        """pose_hrnet.py:333-372 as applied in forward() (:436-457): missing layer = pass-through; every layer is fed the
        LAST branch of the previous stage."""
        xs = []
        for i, c in enumerate(cur):
            q = '%s%d.' % (p, i)
            if i < len(pre):
                if c != pre[i]:
                    t = self.conv(ys[-1], q + '0', pad=1)
                    xs.append(self.affsum([(t, self._bnx(q + '1', c, False), 1)], q + 'out'))
                else:
                    xs.append(ys[i])
            else:
                t, bn = ys[-1], None
                nlay = i + 1 - len(pre)
                for j in range(nlay):
                    t = self.conv(t, '%s%d.0' % (q, j), bn=bn, pad=1, stride=2)
                    bn = self._bnx('%s%d.1' % (q, j), t.shape[3], True)
                bn.relu = False                  # the affsum op applies the final ReLU itself
                xs.append(self.affsum([(t, bn, 1)], q + 'out'))
        return xs

    def build_forward(self):
        """pose_hrnet.py:425-460."""
        x0 = Act((self.N, self.H, self.W, 3), 'image_nhwc')
        x0.needs_grad = False
        op = Op('nchw2nhwc', image=self.image, y=x0, dims=(self.N, 3, self.H, self.W), extra_out=[])
        x0.producer = op
        self.fwd.append(op)
        t = self.conv(x0, 'conv1', pad=1, stride=2)
        t = self.conv(t, 'conv2', bn=self._bnx('bn1', 64, True), pad=1, stride=2)
        x = self.affsum([(t, self._bnx('bn2', 64, False), 1)], 'stem')
        for b in range(4):
            x = self.block(x, 'layer1.%d.' % b, 'BOTTLENECK')
        chans = [[c * HRNET_EXPANSION[self.extra[s]['BLOCK']] for c in self.extra[s]['NUM_CHANNELS']]
                 for s in ('STAGE2', 'STAGE3', 'STAGE4')]
        ys, pre = [x], [256]
        for si, sname in enumerate(('STAGE2', 'STAGE3', 'STAGE4')):
            sc = self.extra[sname]
            xs = self.transition(ys, 'transition%d.' % (si + 1), pre, chans[si])
            for m in range(sc['NUM_MODULES']):
                last = sname == 'STAGE4' and m == sc['NUM_MODULES'] - 1
                xs = self.hr_module(xs, 'stage%d.%d.' % (si + 2, m), sc, not last)
            ys, pre = xs, chans[si]
        k = self.extra.get('FINAL_CONV_KERNEL', 1)
        out = self.conv(ys[0], 'final_layer', pad=1 if k == 3 else 0)
        out.persistent = True
        self.outputs.append(out)
        if self.train:
            self.fwd.append(Op('bnupd', bns=list(self.bns)))
