"""ctypes binding of libfpd_amd.so (include/fpd_amd.h).

The library is the product: if it is missing or an ABI struct size disagrees, import fails loudly
-- there is no CPU / eager fallback (build with `python __graft_entry__.py` or csrc/build.sh).
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get('FPD_AMD_LIB') or os.path.join(_HERE, 'csrc', 'libfpd_amd.so')

F32, BF16 = 0, 1
BN_NONE, BN_TRAIN, BN_EVAL = 0, 1, 2
EPI_PLAIN, EPI_BNRELU_BWD = 0, 1
BACKEND_MFMA, BACKEND_NAIVE, BACKEND_MFMA_GENERIC = 0, 1, 2
(EW_BNRELU_FWD, EW_BNRELU_BWD_R, EW_BN_BWD_APPLY, EW_MAXPOOL_FWD, EW_MAXPOOL_BWD, EW_UPADD_FWD, EW_SUMPOOL,
 EW_ADD, EW_RELU_MASK, EW_DILATE2) = range(10)
(OP_CONV, OP_WGRAD, OP_STEM_FWD, OP_STEM_WGRAD, OP_EW, OP_LOSS, OP_ADAM, OP_MEMSET, OP_WPREP, OP_BNUPD,
 OP_WREDUCE, OP_BNECK, OP_BNECK_FOLD, OP_CONV_PAIR, OP_BNECK_PAIR, OP_EW_PAIR, OP_PCK, OP_HEAD, OP_HEAD_FOLD,
 OP_NOP, OP_AFFSUM, OP_NCHW2NHWC, OP_CONV_F8, OP_WQUANT) = range(24)
AFFSUM_MAX = 4
MAX_STACKS = 8
MAXC = 512

_i32, _i64, _f32, _f64, _vp = C.c_int32, C.c_int64, C.c_float, C.c_double, C.c_void_p


class BnT(C.Structure):
    _fields_ = [('mode', _i32), ('relu', _i32), ('eps', _f32), ('_pad', _i32), ('stats', _vp), ('gamma', _vp),
                ('beta', _vp), ('running_mean', _vp), ('running_var', _vp)]


class ConvT(C.Structure):
    _fields_ = [('N', _i32), ('H', _i32), ('W', _i32), ('C', _i32), ('K', _i32), ('R', _i32), ('S', _i32),
                ('stride', _i32), ('pad', _i32), ('P', _i32), ('Q', _i32), ('dtype', _i32), ('epi', _i32),
                ('_pad', _i32), ('x', _vp), ('w', _vp), ('bias', _vp), ('residual', _vp), ('y', _vp),
                ('out_stats', _vp), ('bn', BnT), ('epi_x', _vp), ('epi_bn', BnT), ('epi_stats', _vp),
                ('wg_partial', _vp), ('wg_stride', _i64), ('wg_bias', _i32), ('wg_count', _i32),
                ('fold_x', _vp), ('fold_bn', BnT), ('fold_stats', _vp), ('fold_out', _vp), ('fold_dgamma', _vp), ('fold_dbeta', _vp)]


class ConvF8T(C.Structure):
    _fields_ = [('c', ConvT), ('w8', _vp), ('w8_scale', _vp)]


class WquantEntryT(C.Structure):
    _fields_ = [('w', _vp), ('w8', _vp), ('scale', _vp), ('K', _i32), ('RSC', _i32)]


class ConvPairT(C.Structure):
    _fields_ = [('a', ConvT), ('b', ConvT)]


class BneckT(C.Structure):
    _fields_ = [('N', _i32), ('H', _i32), ('W', _i32), ('C', _i32), ('P', _i32), ('dtype', _i32), ('_pad', _i32 * 2),
                ('x', _vp), ('y', _vp), ('w1', _vp), ('b1', _vp), ('w2', _vp), ('b2', _vp), ('w3', _vp), ('b3', _vp),
                ('bn1', BnT), ('bn2', BnT), ('bn3', BnT), ('folded', _vp)]


class BneckPairT(C.Structure):
    _fields_ = [('a', BneckT), ('b', BneckT)]


class WgradT(C.Structure):
    _fields_ = [('N', _i32), ('H', _i32), ('W', _i32), ('C', _i32), ('K', _i32), ('R', _i32), ('S', _i32),
                ('stride', _i32), ('pad', _i32), ('P', _i32), ('Q', _i32), ('dtype', _i32), ('x', _vp), ('dy', _vp),
                ('dw', _vp), ('dbias', _vp), ('bn', BnT), ('partial', _vp), ('partial_stride', _i64)]


class StemT(C.Structure):
    _fields_ = [('N', _i32), ('H', _i32), ('W', _i32), ('K', _i32), ('P', _i32), ('Q', _i32), ('dtype', _i32),
                ('_pad', _i32), ('x', _vp), ('w', _vp), ('bias', _vp), ('y', _vp), ('out_stats', _vp), ('dy', _vp),
                ('dw', _vp), ('dbias', _vp), ('partial', _vp), ('partial_stride', _i64)]


class EwT(C.Structure):
    _fields_ = [('op', _i32), ('dtype', _i32), ('N', _i32), ('H', _i32), ('W', _i32), ('C', _i32), ('x', _vp),
                ('x2', _vp), ('dy', _vp), ('add', _vp), ('y', _vp), ('out_stats', _vp), ('bstats', _vp),
                ('dgamma', _vp), ('dbeta', _vp), ('bn', BnT)]


class EwPairT(C.Structure):
    _fields_ = [('a', EwT), ('b', EwT)]


class HeadT(C.Structure):
    _fields_ = [('N', _i32), ('H', _i32), ('W', _i32), ('C', _i32), ('J', _i32), ('dtype', _i32), ('_pad', _i32 * 2),
                ('y0', _vp), ('x', _vp), ('score', _vp), ('next', _vp), ('w_fc', _vp), ('b_fc', _vp), ('w_score', _vp),
                ('b_score', _vp), ('w_fc2', _vp), ('b_fc2', _vp), ('w_score2', _vp), ('b_score2', _vp), ('bn', BnT),
                ('folded', _vp)]


class AffTermT(C.Structure):
    _fields_ = [('x', _vp), ('bn', BnT), ('up', _i32), ('_pad', _i32)]


class AffsumT(C.Structure):
    _fields_ = [('N', _i32), ('H', _i32), ('W', _i32), ('C', _i32), ('dtype', _i32), ('relu', _i32), ('nterms', _i32),
                ('_pad', _i32), ('t', AffTermT * 4), ('y', _vp)]


class LayoutT(C.Structure):
    _fields_ = [('src', _vp), ('dst', _vp), ('N', _i32), ('C', _i32), ('H', _i32), ('W', _i32), ('dtype', _i32), ('_pad', _i32)]


class PckT(C.Structure):
    _fields_ = [('B', _i32), ('J', _i32), ('H', _i32), ('W', _i32), ('dtype', _i32), ('log_slots', _i32), ('thr', C.c_float),
                ('_pad', _i32), ('out', _vp), ('target', _vp), ('counts', _vp), ('log', _vp), ('cursor', _vp), ('losses', _vp)]


class FlipMergeT(C.Structure):
    _fields_ = [('N', _i32), ('J', _i32), ('H', _i32), ('W', _i32), ('shift', _i32), ('_pad', _i32), ('a', _vp), ('b', _vp),
                ('y', _vp), ('src', _i32 * 32)]


class FinalPredsT(C.Structure):
    _fields_ = [('N', _i32), ('J', _i32), ('H', _i32), ('W', _i32), ('post_process', _i32), ('_pad', _i32), ('hm', _vp),
                ('trans', _vp), ('coords', _vp), ('preds', _vp), ('maxvals', _vp)]


class TargetsT(C.Structure):
    _fields_ = [('B', _i32), ('J', _i32), ('H', _i32), ('W', _i32), ('patch', _i32), ('_pad', _i32), ('stride_x', _f64),
                ('stride_y', _f64), ('joints', _vp), ('vis', _vp), ('g', _vp), ('target', _vp), ('weight', _vp)]


class WarpSrcT(C.Structure):
    _fields_ = [('img', _vp), ('h', _i32), ('w', _i32), ('row_bytes', _i64), ('minv', _f64 * 6)]


class WarpT(C.Structure):
    _fields_ = [('B', _i32), ('H', _i32), ('W', _i32), ('_pad', _i32), ('src', _vp), ('mean', _f32 * 3), ('std', _f32 * 3),
                ('out', _vp)]


class LossT(C.Structure):
    _fields_ = [('B', _i32), ('J', _i32), ('H', _i32), ('W', _i32), ('S', _i32), ('dtype', _i32),
                ('target_nchw', _i32), ('alpha', _f32), ('out', _vp * MAX_STACKS), ('dout', _vp * MAX_STACKS),
                ('teacher', _vp), ('target', _vp), ('weight', _vp), ('losses', _vp), ('grad_scale', _f32),
                ('_pad', _i32), ('weight_kd', _vp)]


class AdamT(C.Structure):
    _fields_ = [('n', _i64), ('param', _vp), ('grad', _vp), ('m', _vp), ('v', _vp), ('param_lp', _vp),
                ('lr', _f32), ('beta1', _f32), ('beta2', _f32), ('eps', _f32), ('bias_corr1', _f32),
                ('bias_corr2', _f32), ('grad_scale', _f32), ('_pad', _i32), ('lr_dev', _vp), ('step_dev', _vp)]


class WprepEntryT(C.Structure):
    _fields_ = [('w', _vp), ('w_fwd', _vp), ('w_bwd', _vp), ('K', _i32), ('R', _i32), ('S', _i32), ('C', _i32)]


class BnupdEntryT(C.Structure):
    _fields_ = [('stats', _vp), ('running_mean', _vp), ('running_var', _vp), ('num_batches_tracked', _vp),
                ('count', _f64), ('momentum', _f32), ('C', _i32)]


class WreduceEntryT(C.Structure):
    _fields_ = [('partial', _vp), ('dw', _vp), ('n', _i64), ('stride', _i64), ('count', _i32), ('_pad', _i32)]


class MemsetT(C.Structure):
    _fields_ = [('ptr', _vp), ('bytes', _i64)]


class TableT(C.Structure):
    _fields_ = [('table', _vp), ('n', _i32), ('dtype', _i32), ('max_elems', _i64)]


_STRUCTS = {'fpd_bn_t': BnT, 'fpd_conv_t': ConvT, 'fpd_wgrad_t': WgradT, 'fpd_stem_t': StemT, 'fpd_ew_t': EwT,
            'fpd_loss_t': LossT, 'fpd_adam_t': AdamT, 'fpd_wprep_entry_t': WprepEntryT,
            'fpd_bnupd_entry_t': BnupdEntryT, 'fpd_memset_t': MemsetT, 'fpd_table_t': TableT,
            'fpd_wreduce_entry_t': WreduceEntryT, 'fpd_bneck_t': BneckT, 'fpd_conv_pair_t': ConvPairT, 'fpd_bneck_pair_t': BneckPairT, 'fpd_ew_pair_t': EwPairT, 'fpd_pck_t': PckT, 'fpd_head_t': HeadT, 'fpd_affsum_t': AffsumT, 'fpd_layout_t': LayoutT,
            'fpd_conv_f8_t': ConvF8T, 'fpd_wquant_entry_t': WquantEntryT, 'fpd_flipmerge_t': FlipMergeT, 'fpd_finalpreds_t': FinalPredsT, 'fpd_targets_t': TargetsT,
            'fpd_warp_src_t': WarpSrcT, 'fpd_warp_t': WarpT}

# every symbol include/fpd_amd.h declares: name -> (restype, argtypes)
ABI_VERSION = 2      # include/fpd_amd.h FPD_ABI_VERSION

SYMBOLS = {
    'fpd_conv_forward': (C.c_int, [C.POINTER(ConvT), _vp]),
    'fpd_conv_forward_pair': (C.c_int, [C.POINTER(ConvPairT), _vp]),
    'fpd_conv_forward_f8': (C.c_int, [C.POINTER(ConvF8T), _vp]),
    'fpd_conv_f8_in_domain': (C.c_int, [C.POINTER(ConvT)]),
    'fpd_weight_quant_f8': (C.c_int, [_vp, _i32, _vp]),
    'fpd_bottleneck_forward': (C.c_int, [C.POINTER(BneckT), _vp]),
    'fpd_bottleneck_fold': (C.c_int, [C.POINTER(BneckT), _vp]),
    'fpd_bottleneck_forward_pair': (C.c_int, [C.POINTER(BneckPairT), _vp]),
    'fpd_conv_fused_wgrad_partials': (C.c_int, [C.POINTER(ConvT)]),
    'fpd_conv_fold_supported': (C.c_int, [_vp]),
    'fpd_conv_pair_fold_supported': (C.c_int, [_vp]),
    'fpd_conv_pair_fused_wgrad_partials': (C.c_int, [C.POINTER(ConvPairT), C.POINTER(C.c_int32), C.POINTER(C.c_int32)]),
    'fpd_conv_wgrad': (C.c_int, [C.POINTER(WgradT), _vp]),
    'fpd_wgrad_num_partials': (C.c_int, [C.POINTER(WgradT)]),
    'fpd_wgrad_reduce': (C.c_int, [_vp, _i32, _i64, _vp]),
    'fpd_stem_forward': (C.c_int, [C.POINTER(StemT), _vp]),
    'fpd_stem_wgrad': (C.c_int, [C.POINTER(StemT), _vp]),
    'fpd_stem_wgrad_num_partials': (C.c_int, [C.POINTER(StemT)]),
    'fpd_elementwise': (C.c_int, [C.POINTER(EwT), _vp]),
    'fpd_elementwise_pair': (C.c_int, [C.POINTER(EwPairT), _vp]),
    'fpd_pck': (C.c_int, [C.POINTER(PckT), _vp]),
    'fpd_affsum': (C.c_int, [C.POINTER(AffsumT), _vp]),
    'fpd_flip_w': (C.c_int, [_vp, _vp, _i64, _i32, _vp]),
    'fpd_flip_merge': (C.c_int, [C.POINTER(FlipMergeT), _vp]),
    'fpd_final_preds': (C.c_int, [C.POINTER(FinalPredsT), _vp]),
    'fpd_render_targets': (C.c_int, [C.POINTER(TargetsT), _vp]),
    'fpd_warp_affine': (C.c_int, [C.POINTER(WarpT), _vp]),
    'fpd_head_forward': (C.c_int, [C.POINTER(HeadT), _vp]),
    'fpd_head_fold': (C.c_int, [C.POINTER(HeadT), _vp]),
    'fpd_loss': (C.c_int, [C.POINTER(LossT), _vp]),
    'fpd_adam': (C.c_int, [C.POINTER(AdamT), _vp]),
    'fpd_weight_prep': (C.c_int, [_vp, _i32, _i64, _i32, _vp]),
    'fpd_bn_update_running': (C.c_int, [_vp, _i32, _vp]),
    'fpd_cast': (C.c_int, [_vp, _vp, _i64, _i32, _i32, _vp]),
    'fpd_nchw_to_nhwc': (C.c_int, [_vp, _vp, _i32, _i32, _i32, _i32, _i32, _vp]),
    'fpd_nhwc_to_nchw': (C.c_int, [_vp, _vp, _i32, _i32, _i32, _i32, _i32, _vp]),
    'fpd_plan_create': (_vp, []),
    'fpd_plan_destroy': (None, [_vp]),
    'fpd_plan_add': (C.c_int, [_vp, _i32, _vp, _i64]),
    'fpd_plan_size': (C.c_int, [_vp]),
    'fpd_plan_set_schedule': (C.c_int, [_vp, _i32, _i32, C.POINTER(C.c_int32), _i32]),
    'fpd_plan_mark_event': (C.c_int, [_vp, _i32]),
    'fpd_plan_wait_op': (C.c_int, [_vp, _i32, _vp]),
    'fpd_plan_run': (C.c_int, [_vp, _i32, _i32, _vp]),
    'fpd_plan_run_op': (C.c_int, [_vp, _i32, _vp]),
    'fpd_plan_capture': (C.c_int, [_vp, _i32, _i32, _vp]),
    'fpd_plan_replay': (C.c_int, [_vp, _i32, _vp]),
    'fpd_last_error': (C.c_char_p, []),
    'fpd_set_backend': (C.c_int, [_i32]),
    'fpd_set_option': (C.c_int, [C.c_char_p, _i32]),
    'fpd_abi_sizeof': (C.c_int, [C.c_char_p]),
    'fpd_abi_version': (C.c_int, []),
    'fpd_stats_words': (C.c_int64, [C.c_int32]),
    'fpd_event_create': (_vp, []),
    'fpd_event_record': (C.c_int, [_vp, _vp]),
    'fpd_event_elapsed_ms': (C.c_float, [_vp, _vp]),
    'fpd_event_destroy': (None, [_vp]),
}

_lib = None


class FpdError(RuntimeError):
    pass


def lib_sha16():
    """First 16 hex digits of the SHA-256 of the library file in use: ties a measured record (profiles/*.json written by the GPU
    tests) to the build it was measured on."""
    import hashlib
    with open(LIB_PATH, 'rb') as f:
        return hashlib.sha256(f.read()).hexdigest()[:16]


def lib():
    """Load libfpd_amd.so once; raise (never fall back) if it is absent or ABI-incompatible."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise FpdError('HIP extension %s is missing: run `python __graft_entry__.py` (build()) first; '
                       'there is no CPU fallback for the FPD path' % LIB_PATH)
    # torch ships its own HIP runtime (torch/lib/libamdhip64.so).  It must be in the process BEFORE our library is
    # dlopen'ed so that libfpd_amd.so binds to the SAME runtime instance torch uses (streams, device pointers);
    # loaded the other way round our kernels would talk to /opt/rocm's runtime and see "no ROCm-capable device".
    import torch  # noqa: F401
    try:
        torch.cuda.is_available()
    except Exception:
        pass
    l = C.CDLL(LIB_PATH)
    for name, (res, args) in SYMBOLS.items():
        fn = getattr(l, name)          # AttributeError if the symbol is not exported
        fn.restype = res
        fn.argtypes = args
    if l.fpd_abi_version() != ABI_VERSION:       # buffer encodings sizeof cannot see (include/fpd_amd.h FPD_ABI_VERSION)
        raise FpdError('ABI version mismatch: library %d, python host %d' % (l.fpd_abi_version(), ABI_VERSION))
    for name, st in _STRUCTS.items():
        got = l.fpd_abi_sizeof(name.encode())
        if got != C.sizeof(st):
            raise FpdError('ABI mismatch for %s: library %d bytes, python %d bytes' % (name, got, C.sizeof(st)))
    _lib = l
    return l


def check(rc, what=''):
    if rc != 0:
        raise FpdError('%s failed (rc=%d): %s' % (what or 'fpd call', rc, lib().fpd_last_error().decode()))


def set_backend(backend):
    return lib().fpd_set_backend(backend)


def set_option(name, value):
    """Process-wide run-time knob of the library (include/fpd_amd.h fpd_set_option); returns the previous value."""
    prev = lib().fpd_set_option(name.encode(), int(value))
    if prev < 0:
        check(prev, 'fpd_set_option(%s)' % name)
    return prev


def current_stream():
    import torch
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


class Plan:
    """Owns a native fpd_plan (recorded op list) and the ctypes arg structs' lifetime."""

    def __init__(self):
        self._l = lib()
        self._p = C.c_void_p(self._l.fpd_plan_create())
        self._graphs = {}
        self._types = []                 # op code of every plan op (host-side bookkeeping: launches per phase)

    def add(self, op, args):
        rc = self._l.fpd_plan_add(self._p, op, C.byref(args), C.sizeof(args))
        if rc < 0:
            check(rc, 'fpd_plan_add')
        self._types.append(op)
        return rc

    def op_type(self, k):
        return self._types[k]

    def __len__(self):
        return self._l.fpd_plan_size(self._p)

    def set_schedule(self, op, lane, waits=()):
        arr = (C.c_int32 * max(len(waits), 1))(*waits)
        check(self._l.fpd_plan_set_schedule(self._p, op, lane, arr, len(waits)), 'fpd_plan_set_schedule')

    def mark_event(self, op):
        check(self._l.fpd_plan_mark_event(self._p, op), 'fpd_plan_mark_event')

    def wait_op(self, op, stream):
        """Make `stream` (a raw hipStream_t as c_void_p) wait for the last run of plan op `op`."""
        check(self._l.fpd_plan_wait_op(self._p, op, stream), 'fpd_plan_wait_op')

    def run(self, begin, end, stream=None):
        check(self._l.fpd_plan_run(self._p, begin, end, stream if stream is not None else current_stream()),
              'fpd_plan_run')

    def run_op(self, op, stream=None):
        """One recorded op on `stream` itself (lane and waits ignored) -- for timing a kernel exactly as the step launches it."""
        check(self._l.fpd_plan_run_op(self._p, op, stream if stream is not None else current_stream()), 'fpd_plan_run_op')

    def capture(self, begin, end, stream):
        gid = self._l.fpd_plan_capture(self._p, begin, end, stream)
        if gid < 0:
            check(gid, 'fpd_plan_capture')
        return gid

    def replay(self, gid, stream=None):
        check(self._l.fpd_plan_replay(self._p, gid, stream if stream is not None else current_stream()),
              'fpd_plan_replay')

    def __del__(self):
        try:
            if self._p:
                self._l.fpd_plan_destroy(self._p)
                self._p = None
        except Exception:
            pass

