"""C-ABI surface (no GPU needed): the library builds/loads here, exports every function include/fpd_amd.h declares,
the ctypes mirrors have the library's struct sizes, and argument validation fails loudly without touching a device."""
import os
import re
import subprocess

import pytest

from tests.conftest import ROOT


@pytest.fixture(scope='module')
def runtime():
    from fpd_amd import runtime as R
    if not os.path.exists(R.LIB_PATH):
        subprocess.check_call(['bash', os.path.join(os.path.dirname(R.LIB_PATH), 'build.sh')])
    R.lib()
    return R


def declared_functions():
    src = open(os.path.join(ROOT, 'include', 'fpd_amd.h')).read()
    src = re.sub(r'/\*.*?\*/', '', src, flags=re.S)
    return sorted(set(re.findall(r'\b(fpd_[a-z0-9_]+)\s*\(', src)) - {'fpd_stream_t'})


def test_every_declared_symbol_is_exported_and_bound(runtime):
    names = declared_functions()
    assert len(names) >= 25
    lib = runtime.lib()
    for n in names:
        assert hasattr(lib, n), 'libfpd_amd.so does not export ' + n
        assert n in runtime.SYMBOLS, 'runtime.py has no ctypes prototype for ' + n
    assert sorted(runtime.SYMBOLS) == names


def test_struct_sizes_match(runtime):
    import ctypes
    for name, st in runtime._STRUCTS.items():
        assert runtime.lib().fpd_abi_sizeof(name.encode()) == ctypes.sizeof(st), name
    assert runtime.lib().fpd_abi_sizeof(b'nope') == -1
    assert runtime.lib().fpd_abi_version() == runtime.ABI_VERSION == 2
    assert runtime.lib().fpd_stats_words(64) == 4 * 4 * 64


def test_validation_errors_without_device(runtime):
    lib = runtime.lib()
    a = runtime.ConvT()
    assert lib.fpd_conv_forward(a, None) != 0 and b'null' in lib.fpd_last_error()
    a.x = a.w = a.y = 1
    a.N, a.H, a.W, a.C, a.K, a.R, a.S, a.stride, a.pad, a.P, a.Q = 1, 8, 8, 16, 16, 3, 3, 1, 1, 7, 8
    assert lib.fpd_conv_forward(a, None) != 0 and b'inconsistent' in lib.fpd_last_error()
    p = lib.fpd_plan_create()
    assert lib.fpd_plan_add(p, 99, runtime.ctypes_byref(a) if hasattr(runtime, 'ctypes_byref') else None, 0) < 0
    lib.fpd_plan_destroy(p)
    with pytest.raises(runtime.FpdError):
        runtime.check(-2, 'x')


def test_missing_library_is_an_error_not_a_fallback(runtime, monkeypatch):
    monkeypatch.setattr(runtime, '_lib', None)
    monkeypatch.setattr(runtime, 'LIB_PATH', '/nonexistent/libfpd_amd.so')
    with pytest.raises(runtime.FpdError):
        runtime.lib()


def test_round2_entry_points_validate_their_arguments_without_a_device(runtime):
    """validate / data-pipeline / fp8 entry points: bad arguments are refused before anything is launched."""
    R, lib = runtime, runtime.lib()
    err = lambda: lib.fpd_last_error().decode()
    assert lib.fpd_flip_w(None, None, 4, 4, None) != 0 and 'null' in err()
    m = R.FlipMergeT()
    assert lib.fpd_flip_merge(m, None) != 0 and 'null' in err()
    m.a, m.b, m.y = 8, 16, 24
    m.N, m.J, m.H, m.W = 1, 40, 4, 4
    assert lib.fpd_flip_merge(m, None) != 0 and 'J <=' in err()
    m.J = 4
    m.src[2] = 9
    assert lib.fpd_flip_merge(m, None) != 0 and 'out of range' in err()
    m.src[2] = 2
    m.y = m.b                                     # in-place over the flipped map would read what it overwrites
    assert lib.fpd_flip_merge(m, None) != 0 and 'aliased' in err()
    f = R.FinalPredsT()
    f.N, f.J, f.H, f.W = 1, 2, 4, 4
    f.hm, f.coords, f.maxvals, f.trans = 8, 16, 24, 32      # trans without preds
    assert lib.fpd_final_preds(f, None) != 0 and 'go together' in err()
    t = R.TargetsT()
    t.B, t.J, t.H, t.W, t.patch = 1, 2, 8, 8, 12             # even patch edge
    t.joints, t.vis, t.g, t.target, t.weight = 8, 16, 24, 32, 40
    t.stride_x = t.stride_y = 4.0
    assert lib.fpd_render_targets(t, None) != 0 and 'bad dims' in err()
    w = R.WarpT()
    w.B, w.H, w.W, w.src, w.out = 1, 8, 8, 8, 16
    assert lib.fpd_warp_affine(w, None) != 0 and 'std' in err()
    c8 = R.ConvF8T()
    c = c8.c
    c.x = c.w = c.y = 8
    c.N, c.H, c.W, c.C, c.K, c.R, c.S, c.stride, c.pad, c.P, c.Q, c.dtype = 1, 8, 8, 32, 32, 3, 3, 1, 1, 8, 8, R.BF16
    assert lib.fpd_conv_forward_f8(c8, None) != 0 and 'null fp8' in err()
    c8.w8, c8.w8_scale = 24, 32                               # e4m3 weights must be 16-byte aligned
    assert lib.fpd_conv_forward_f8(c8, None) != 0 and 'aligned' in err()
    assert lib.fpd_conv_f8_in_domain(c) == 1
    c.stride, c.P, c.Q = 2, 4, 4
    assert lib.fpd_conv_f8_in_domain(c) == 0
    assert lib.fpd_weight_quant_f8(None, 3, None) != 0 and 'null table' in err()
    p = lib.fpd_plan_create()
    assert lib.fpd_plan_add(p, R.OP_CONV_F8, R.C.byref(c8), R.C.sizeof(c8)) == 0
    assert lib.fpd_plan_add(p, R.OP_CONV_F8, R.C.byref(c), R.C.sizeof(c)) < 0        # wrong args size for the op
    lib.fpd_plan_destroy(p)


def _conv(R, N, H, W, C, K, r, bwd=True, bn=False, dtype=None):
    a = R.ConvT()
    a.x = a.w = a.y = 64                          # never dereferenced by the predicates below
    a.N, a.H, a.W, a.C, a.K, a.R, a.S, a.stride, a.pad, a.P, a.Q = N, H, W, C, K, r, r, 1, (r - 1) // 2, H, W
    a.dtype = R.BF16 if dtype is None else dtype
    a.epi = R.EPI_BNRELU_BWD if bwd else R.EPI_PLAIN
    if bwd:
        a.epi_x, a.epi_stats = 64, 64
        a.epi_bn.mode, a.epi_bn.gamma, a.epi_bn.beta, a.epi_bn.stats = R.BN_TRAIN, 64, 64, 64
    if bn:
        a.bn.mode, a.bn.gamma, a.bn.beta, a.bn.stats = R.BN_TRAIN, 64, 64, 64
    return a


def test_host_side_dispatch_predicates_of_round_3_without_a_device(runtime):
    """Pure host logic of the round-3 entry points: which launches evaluate a folded BN-backward apply (persistent kernel
    from 256 pixel tiles, the halo-tile kernel's FOLD variants below that only while its blocks carry <= 64 output
    channels), how many slabs the deterministic weight-gradient kernels write, and the options of fpd_set_option."""
    import ctypes
    R, lib = runtime, runtime.lib()
    if not all(hasattr(R, n) for n in ('BF16', 'EPI_BNRELU_BWD', 'EPI_PLAIN', 'BN_TRAIN')):
        pytest.skip('runtime.py does not export the enum names this test uses')
    fold = lambda a: lib.fpd_conv_fold_supported(ctypes.byref(a))
    # data gradients of the student's Bottlenecks at batch 32: 64x64 and 32x32 -> persistent kernel, 16x16 and below -> halo tile
    assert fold(_conv(R, 32, 64, 64, 64, 128, 1)) == 1          # dgrad of conv1 (1x1, 128 <- 64)
    assert fold(_conv(R, 32, 64, 64, 64, 64, 3)) == 1           # dgrad of conv2 (3x3)
    assert fold(_conv(R, 32, 32, 32, 64, 128, 1)) == 1
    assert fold(_conv(R, 32, 16, 16, 64, 128, 1)) == 1          # 64 tiles: halo-tile kernel, TN drops to 2
    assert fold(_conv(R, 32, 4, 4, 64, 64, 3)) == 1
    # 128 tiles x 128 output channels: the halo-tile kernel keeps TN = 4, for which no FOLD variant is compiled -- since round 6
    # the streaming 1x1 kernel (conv_c1, from 2048 pixels) serves this launch, so the fold is available; without it: not
    assert fold(_conv(R, 4, 64, 64, 64, 128, 1)) == 1
    prev_c1 = lib.fpd_set_option(b'conv_c1', 0)
    assert fold(_conv(R, 4, 64, 64, 64, 128, 1)) == 0
    lib.fpd_set_option(b'conv_c1', prev_c1)
    assert fold(_conv(R, 2, 16, 16, 64, 128, 1)) == 1          # 512 pixels: below conv_c1's threshold, 4 tiles of the halo-tile kernel
    assert fold(_conv(R, 32, 64, 64, 64, 64, 3, bwd=False)) == 0        # not a BNRELU_BWD data gradient
    assert fold(_conv(R, 32, 64, 64, 64, 64, 3, bn=True)) == 0          # a prologue BN of its own
    pair = R.ConvPairT()
    pair.a, pair.b = _conv(R, 32, 64, 64, 64, 128, 1), _conv(R, 32, 32, 32, 64, 128, 1)
    assert lib.fpd_conv_pair_fold_supported(ctypes.byref(pair)) == 1
    pair.a, pair.b = _conv(R, 4, 64, 64, 64, 128, 1), _conv(R, 4, 32, 32, 64, 128, 1)      # 160 tiles: halo-tile pair at TN = 4 ...
    assert lib.fpd_conv_pair_fold_supported(ctypes.byref(pair)) == 1                     # ... taken by conv_c1 since round 6
    prev_c1 = lib.fpd_set_option(b'conv_c1', 0)
    assert lib.fpd_conv_pair_fold_supported(ctypes.byref(pair)) == 0
    lib.fpd_set_option(b'conv_c1', prev_c1)
    # the new kernels' own predicates: the 3x3 strip kernel takes 64 -> 64 on 16..64-wide maps from 32768 pixels; options round-trip
    prev_c3 = lib.fpd_set_option(b'conv_c3', 2)
    assert fold(_conv(R, 2, 16, 16, 64, 64, 3)) == 1 and fold(_conv(R, 2, 16, 16, 32, 32, 3)) == 1      # (32 -> 32: conv_tile's FOLD variant)
    assert lib.fpd_set_option(b'conv_c3', prev_c3) == 2
    assert lib.fpd_set_option(b'conv_c1_launches', 0) == 0 and lib.fpd_set_option(b'conv_c3_launches', 0) == 0
    # deterministic weight gradients: slab counts follow the dispatch order tile -> mfma -> small-C -> direct
    w = R.WgradT()
    w.x = w.dy = w.dw = 64
    w.N, w.H, w.W, w.C, w.K, w.R, w.S, w.stride, w.pad, w.P, w.Q = 32, 64, 64, 64, 64, 3, 3, 1, 1, 64, 64
    w.dtype = R.BF16
    assert lib.fpd_wgrad_num_partials(ctypes.byref(w)) == 32                        # wgrad3 (round 5): one slab per contiguous pixel range, 4 pieces each
    w.C = w.K = 128
    assert lib.fpd_wgrad_num_partials(ctypes.byref(w)) == 32                        # 3x3 halo-tile kernel: one slab per block, 128 blocks over 4 (k, c) tiles
    w.C = w.K = 64
    w.H = w.W = w.P = w.Q = 24                                                     # HRNet width: H4 variant of the same kernel
    assert lib.fpd_wgrad_num_partials(ctypes.byref(w)) > 0
    w.N, w.H, w.W, w.P, w.Q, w.C, w.K, w.stride = 32, 256, 192, 128, 96, 3, 64, 2   # HRNet stem: the small-C kernel
    assert lib.fpd_wgrad_num_partials(ctypes.byref(w)) == 512
    assert lib.fpd_set_option(b'no_such_option', 1) < 0 and b'unknown' in lib.fpd_last_error()
    prev = lib.fpd_set_option(b'conv_pp_blocks', 64)
    assert prev == 256 and lib.fpd_set_option(b'conv_pp_blocks', prev) == 64


def test_exact_statistics_encoding_round_trip_and_order_independence():
    """include/fpd_amd.h fpd_stat_t as the host sees it (executor.Arenas.stats_write / stats_read on CPU tensors): a value
    travels as two 64-bit integer limbs, hi = rint(v * 2^20) and lo = rint((v - hi * 2^-20) * 2^60); the split is exact to
    2^-61 over the whole range a BatchNorm sum can take, and -- the point of the format -- integer limb sums do not depend
    on the order of the addends, where fp64 sums of the same addends do."""
    import numpy as np
    import torch
    from fpd_amd import executor as E, graph as G
    A = E.Arenas(torch.device('cpu'), 0)
    C, R = 8, G.STATS_REPLICAS
    A.alloc('stats', R * 2 * C + 64)
    buf = G.Buf('stats', 16, (R, 2, C))
    assert A.tensor('stats').dtype == torch.int64 and A.tensor('stats').numel() == 2 * (R * 2 * C + 64)
    assert A.ptr(buf) - A.tensor('stats').data_ptr() == 16 * 16          # two 8-byte limbs per logical element
    rng = np.random.RandomState(0)
    vals = torch.from_numpy(np.concatenate([rng.standard_normal(R * C) * 10.0 ** rng.uniform(-12, 11, R * C),
                                            np.array([0.0, 1.0, -1.0, 2.0 ** -40, -2.0 ** 40, 1e-15, 123456.789, -0.1] * (R * C // 8))]
                                           ).reshape(R, 2, C))
    A.stats_write(buf, vals)
    back = A.stats_read(buf)
    err = (back - vals).abs()
    assert float(err.max()) <= 2.0 ** -60 and float((err / vals.abs().clamp_min(1e-300))[vals.abs() > 1e-9].max()) < 1e-9
    raw = A.view(buf)
    assert raw.shape == (R, 2, 2, C) and int(raw[:, :, 1, :].abs().max()) <= 2 ** 39      # |lo| <= 2^39: 2^24 addends fit 63 bits
    # order independence: sum 2048 "block partials" as limbs in two different orders -> identical limbs; as doubles -> not
    parts = rng.standard_normal(2048) * 10.0 ** rng.uniform(-6, 6, 2048)
    hi = np.rint(parts * 2.0 ** 20).astype(np.int64)
    lo = np.rint((parts - hi / 2.0 ** 20) * 2.0 ** 60).astype(np.int64)
    perm = rng.permutation(2048)
    assert hi.sum() == hi[perm].sum() and lo.sum() == lo[perm].sum()
    s1 = s2 = 0.0
    for v in parts:
        s1 += v
    for v in parts[perm]:
        s2 += v
    assert s1 != s2                                          # what the fp64 atomics of rounds 1-3 did to the last bits
    exact = float(hi.sum()) / 2.0 ** 20 + float(lo.sum()) / 2.0 ** 60
    assert abs(exact - s1) <= 1e-9 * np.abs(parts).sum()
    # headroom: 32 768 same-sign contributions below 2^-9 -- the per-thread sums of y^2 of a channel with
    # rms 0.009 -- wrapped the lo limb of the first encoding (|lo| <= 2^51: 4 096 worst-case addends); |lo| <= 2^39 leaves 2^24
    small = np.full(32768, 1.3e-3) * (1.0 + 1e-3 * rng.standard_normal(32768))
    hi = np.rint(small * 2.0 ** 20).astype(np.int64)
    lo = np.rint((small - hi / 2.0 ** 20) * 2.0 ** 60).astype(np.int64)
    assert int(np.abs(lo).max()) <= 2 ** 39 and abs(int(lo.sum())) < 2 ** 55
    joined = float(hi.sum()) / 2.0 ** 20 + float(lo.sum()) / 2.0 ** 60
    assert abs(joined - small.sum()) <= 1e-12 * small.sum()
    old_lo = np.rint((small - np.rint(small * 256.0) / 256.0) * 2.0 ** 60)      # what version 1 summed: past 2^63
    assert float(old_lo.sum()) > 2.0 ** 63
