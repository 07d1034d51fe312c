// C-ABI entry points (include/fpd_amd.h): argument validation, backend dispatch, execution plan,
// hipGraph capture/replay, HIP-event timing helpers.
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "common.h"

int fpd_conv_mfma_launch(const fpd_conv_t& a, hipStream_t st);
int fpd_conv_tile_launch(const fpd_conv_t& a, hipStream_t st);
int fpd_conv_pp_launch(const fpd_conv_t& a, hipStream_t st);
int fpd_conv_c1_launch(const fpd_conv_t& a, hipStream_t st);
int fpd_conv_c3_launch(const fpd_conv_t& a, hipStream_t st);
int fpd_conv_c3_pair_launch(const fpd_conv_t& a, const fpd_conv_t& b, hipStream_t st);
int fpd_conv_c3_fold_ok(const fpd_conv_t& a, const fpd_conv_t* b);
int fpd_conv_c3_option(int which, int value);
int fpd_conv_c1_pair_launch(const fpd_conv_t& a, const fpd_conv_t& b, hipStream_t st);
int fpd_conv_c1_fold_ok(const fpd_conv_t& a, const fpd_conv_t* b);
int fpd_conv_c1_option(int which, int value);
int fpd_conv_c1_wgrad_partials(const fpd_conv_t& a);
int fpd_conv_c1_pair_wgrad_partials(const fpd_conv_t& a, const fpd_conv_t& b, int* na, int* nb);
int fpd_conv_pp_pair_launch(const fpd_conv_t& a, const fpd_conv_t& b, hipStream_t st);
int fpd_conv_pp_fold_ok(const fpd_conv_t& a, const fpd_conv_t* b);
int fpd_conv_tile_fold_ok(const fpd_conv_t& a);
int fpd_conv_tile_pair_fold_ok(const fpd_conv_t& a, const fpd_conv_t& b);
int fpd_conv_pp_option(int which, int value);
int fpd_conv_pp_wgrad_partials(const fpd_conv_t& a);
int fpd_conv_pp_pair_wgrad_partials(const fpd_conv_t& a, const fpd_conv_t& b, int* na, int* nb);
int fpd_conv_tile_pair_launch(const fpd_conv_t& a, const fpd_conv_t& b, hipStream_t st);
int fpd_bneck_fused_launch(const fpd_bneck_t& a, hipStream_t st);
int fpd_bneck_fold_launch(const fpd_bneck_t& a, float* out, hipStream_t st);
int fpd_pck_launch(const fpd_pck_t& a, hipStream_t st);
int fpd_conv_smallc_launch(const fpd_conv_t& a, hipStream_t st);
bool fpd_conv_f8_domain(const fpd_conv_t& a);
int fpd_conv_tile_f8_launch(const fpd_conv_t& a, const void* w8, const float* wscale, hipStream_t st);
int fpd_weight_quant_f8_launch(const fpd_wquant_entry_t* table, int n, hipStream_t st);
int fpd_flip_w_launch(const float* x, float* y, int64_t rows, int W, hipStream_t st);
int fpd_flip_merge_launch(const fpd_flipmerge_t& p, hipStream_t st);
int fpd_final_preds_launch(const fpd_finalpreds_t& p, hipStream_t st);
int fpd_render_targets_launch(const fpd_targets_t& a, hipStream_t st);
int fpd_warp_affine_launch(const fpd_warp_t& a, hipStream_t st);
int fpd_head_fused_launch(const fpd_head_t& a, hipStream_t st);
int fpd_head_fold_launch(const fpd_head_t& a, float* out, hipStream_t st);
int fpd_bneck_fused_pair_launch(const fpd_bneck_t& a, const fpd_bneck_t& b, hipStream_t st);
int fpd_wgrad_mfma_launch(const fpd_wgrad_t& a, hipStream_t st);
int fpd_wgrad_tile_launch(const fpd_wgrad_t& a, hipStream_t st);
int fpd_wgrad_tile_partials(const fpd_wgrad_t& a);
int fpd_wgrad_smallc_launch(const fpd_wgrad_t& a, hipStream_t st);
int fpd_wgrad_smallc_partials(const fpd_wgrad_t& a);
int fpd_wgrad_mfma_partials(const fpd_wgrad_t& a);
int fpd_wgrad_naive_partials(const fpd_wgrad_t& a);
int fpd_stem_wgrad_mfma_partials(const fpd_stem_t& a);
int fpd_stem_wgrad_partials(const fpd_stem_t& a);
int fpd_wreduce_launch(const fpd_wreduce_entry_t* table, int n, int64_t max_elems, hipStream_t st);
int fpd_conv_naive_launch(const fpd_conv_t& a, hipStream_t st);
int fpd_wgrad_naive_launch(const fpd_wgrad_t& a, hipStream_t st);
int fpd_stem_forward_launch(const fpd_stem_t& a, hipStream_t st);
int fpd_stem_forward_mfma_launch(const fpd_stem_t& a, hipStream_t st);
int fpd_stem_forward_s2d_launch(const fpd_stem_t& a, hipStream_t st);
int fpd_stem_wgrad_s2d_launch(const fpd_stem_t& a, hipStream_t st);
int fpd_stem_wgrad_s2d_partials(const fpd_stem_t& a);
int fpd_stem_wgrad_mfma_launch(const fpd_stem_t& a, hipStream_t st);
int fpd_stem_wgrad_launch(const fpd_stem_t& a, hipStream_t st);
int fpd_elementwise_launch(const fpd_ew_t& a, hipStream_t st);
int fpd_elementwise_pair_launch(const fpd_ew_t& a, const fpd_ew_t& b, hipStream_t st);
int fpd_affsum_launch(const fpd_affsum_t& a, hipStream_t st);
int fpd_loss_launch(const fpd_loss_t& a, hipStream_t st);
int fpd_adam_launch(const fpd_adam_t& a, hipStream_t st);
int fpd_weight_prep_launch(const fpd_wprep_entry_t* table, int n, int64_t max_elems, int dtype, hipStream_t st);
int fpd_bn_update_running_launch(const fpd_bnupd_entry_t* table, int n, hipStream_t st);
int fpd_cast_launch(const void* src, void* dst, int64_t n, int sd, int dd, hipStream_t st);
int fpd_nchw_to_nhwc_launch(const float* src, void* dst, int N, int C, int H, int W, int dtype, hipStream_t st);
int fpd_nhwc_to_nchw_launch(const void* src, float* dst, int N, int C, int H, int W, int dtype, hipStream_t st);

int g_fpd_backend = FPD_BACKEND_MFMA;
static thread_local char g_err[512] = "";

// ---- launch log (common.h FPD_LAUNCH) ----
static FILE* g_launch_log_file = nullptr;
static bool launch_log_open() {
    const char* path = getenv("FPD_LAUNCH_LOG");
    if (path == nullptr || path[0] == 0) return false;
    g_launch_log_file = fopen(path, "w");
    if (g_launch_log_file == nullptr) fprintf(stderr, "[fpd_amd] FPD_LAUNCH_LOG: cannot open %s\n", path);
    return g_launch_log_file != nullptr;
}
bool g_fpd_launch_log = launch_log_open();
static thread_local char g_op_tag[256] = "-";          // the plan op being issued (set by fpd_plan_run / fpd_plan_run_op)
void fpd_log_launch(const char* kernel, const dim3& grid, const dim3& block) {
    if (g_launch_log_file == nullptr) return;
    fprintf(g_launch_log_file, "%s\t%u\t%u\t%s\n", kernel, grid.x * grid.y * grid.z, block.x * block.y * block.z, g_op_tag);
    fflush(g_launch_log_file);
}

int fpd_fail(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

static int check_launch() {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fpd_fail(-100 - (int)e, "kernel launch failed: %s", hipGetErrorString(e));
    return 0;
}

static int validate_conv_dims(int N, int H, int W, int C, int K, int R, int S, int stride, int pad, int P, int Q) {
    FPD_REQUIRE(N > 0 && H > 0 && W > 0 && C > 0 && K > 0 && R > 0 && S > 0 && stride > 0 && pad >= 0,
                "conv: non-positive dimension");
    FPD_REQUIRE(P == (H + 2 * pad - R) / stride + 1 && Q == (W + 2 * pad - S) / stride + 1,
                "conv: output %dx%d inconsistent with input %dx%d k=%dx%d stride=%d pad=%d", P, Q, H, W, R, S, stride, pad);
    FPD_REQUIRE((int64_t)N * H * W * C < (1ll << 31) && (int64_t)N * P * Q * K < (1ll << 31), "conv: tensor too large for 32-bit indexing");
    return 0;
}

extern "C" {

const char* fpd_last_error(void) { return g_err; }
int fpd_abi_version(void) { return FPD_ABI_VERSION; }
int64_t fpd_stats_words(int32_t channels) { return FPD_STATS_WORDS(channels); }
int fpd_set_backend(int32_t backend) {
    const int prev = g_fpd_backend;
    if (backend == FPD_BACKEND_MFMA || backend == FPD_BACKEND_NAIVE || backend == FPD_BACKEND_MFMA_GENERIC) g_fpd_backend = backend;
    return prev;
}

static int g_wgrad_tile_only = 0;      // tests: fail instead of falling through to the generic weight-gradient kernels

int fpd_set_option(const char* name, int32_t value) {
    FPD_REQUIRE(name, "set_option: null name");
    if (!strcmp(name, "conv_pp")) return fpd_conv_pp_option(0, value);
    if (!strcmp(name, "conv_pp_blocks")) return fpd_conv_pp_option(1, value);
    if (!strcmp(name, "conv_c1")) return fpd_conv_c1_option(0, value);
    if (!strcmp(name, "conv_c1_blocks")) return fpd_conv_c1_option(1, value);
    if (!strcmp(name, "conv_c1_launches")) return fpd_conv_c1_option(2, value);      // (read-only: launches served so far)
    if (!strcmp(name, "conv_c3")) return fpd_conv_c3_option(0, value);
    if (!strcmp(name, "conv_c3_blocks")) return fpd_conv_c3_option(1, value);
    if (!strcmp(name, "conv_c3_launches")) return fpd_conv_c3_option(2, value);
    if (!strcmp(name, "wgrad_tile_only")) { g_wgrad_tile_only = value; return 0; }
    return fpd_fail(-2, "set_option: unknown option '%s'", name);
}

int fpd_abi_sizeof(const char* n) {
#define SZ(T) if (!strcmp(n, #T)) return (int)sizeof(T)
    SZ(fpd_bn_t); SZ(fpd_conv_t); SZ(fpd_wgrad_t); SZ(fpd_stem_t); SZ(fpd_ew_t); SZ(fpd_loss_t); SZ(fpd_adam_t);
    SZ(fpd_wprep_entry_t); SZ(fpd_bnupd_entry_t); SZ(fpd_memset_t); SZ(fpd_table_t); SZ(fpd_wreduce_entry_t); SZ(fpd_bneck_t); SZ(fpd_conv_pair_t); SZ(fpd_bneck_pair_t); SZ(fpd_ew_pair_t); SZ(fpd_pck_t); SZ(fpd_head_t); SZ(fpd_affsum_t); SZ(fpd_layout_t);
    SZ(fpd_conv_f8_t); SZ(fpd_wquant_entry_t); SZ(fpd_flipmerge_t); SZ(fpd_finalpreds_t); SZ(fpd_targets_t); SZ(fpd_warp_src_t); SZ(fpd_warp_t);
#undef SZ
    return -1;
}

static int validate_conv(const fpd_conv_t* a) {
    FPD_REQUIRE(a && a->x && a->w && a->y, "conv: null pointer");
    int rc = validate_conv_dims(a->N, a->H, a->W, a->C, a->K, a->R, a->S, a->stride, a->pad, a->P, a->Q);
    if (rc) return rc;
    FPD_REQUIRE(a->dtype == FPD_F32 || a->dtype == FPD_BF16, "conv: bad dtype %d", a->dtype);
    FPD_REQUIRE(a->bn.mode == FPD_BN_NONE || (a->bn.gamma && a->bn.beta), "conv: BN prologue without gamma/beta");
    FPD_REQUIRE(a->bn.mode != FPD_BN_TRAIN || a->bn.stats, "conv: train-mode BN without statistics");
    FPD_REQUIRE(a->bn.mode != FPD_BN_EVAL || (a->bn.running_mean && a->bn.running_var), "conv: eval-mode BN without running stats");
    FPD_REQUIRE(a->epi == FPD_EPI_PLAIN || (a->epi_x && a->epi_stats && a->epi_bn.mode == FPD_BN_TRAIN && a->epi_bn.stats),
                "conv: BNRELU_BWD epilogue needs epi_x, epi_stats and a train-mode epi_bn");
    return 0;
}

static int dispatch_conv(const fpd_conv_t* a, hipStream_t st) {
    int rc = 1;
    if (g_fpd_backend == FPD_BACKEND_MFMA) rc = fpd_conv_c1_launch(*a, st);      // big maps, 1x1: streaming kernel
    if (rc == 1 && g_fpd_backend == FPD_BACKEND_MFMA) rc = fpd_conv_c3_launch(*a, st);      // big maps, 3x3 64 -> 64: strip kernel
    if (rc == 1 && g_fpd_backend == FPD_BACKEND_MFMA) rc = fpd_conv_pp_launch(*a, st);      // big maps: persistent kernel
    FPD_REQUIRE(rc != 1 || a->wg_partial == nullptr, "conv: a fused weight gradient (wg_partial) needs the persistent kernel; "
                "fpd_conv_fused_wgrad_partials() reports 0 for this launch");
    if (rc == 1 && g_fpd_backend == FPD_BACKEND_MFMA) rc = fpd_conv_tile_launch(*a, st);
    FPD_REQUIRE(rc != 1 || a->fold_x == nullptr, "conv: a folded BN-backward apply (fold_x) needs the persistent or the halo-tile "
                "kernel; fpd_conv_fold_supported() reports 0 for this launch");
    if (rc == 1 && g_fpd_backend != FPD_BACKEND_NAIVE) rc = fpd_conv_mfma_launch(*a, st);
    if (rc == 1 && g_fpd_backend != FPD_BACKEND_NAIVE) rc = fpd_conv_smallc_launch(*a, st);   // tiny input-channel counts (3, 17)
    if (rc == 1) rc = fpd_conv_naive_launch(*a, st);
    return rc;
}

int fpd_conv_forward(const fpd_conv_t* a, fpd_stream_t stream) {
    int rc = validate_conv(a);
    if (rc) return rc;
    rc = dispatch_conv(a, (hipStream_t)stream);
    return rc ? rc : check_launch();
}

int fpd_conv_fold_supported(const fpd_conv_t* a) {
    if (!a || g_fpd_backend != FPD_BACKEND_MFMA || validate_conv(a) != 0) return 0;
    return (fpd_conv_c1_fold_ok(*a, nullptr) || fpd_conv_c3_fold_ok(*a, nullptr) || fpd_conv_pp_fold_ok(*a, nullptr) || fpd_conv_tile_fold_ok(*a)) ? 1 : 0;
}
int fpd_conv_pair_fold_supported(const fpd_conv_pair_t* p) {
    if (!p || g_fpd_backend != FPD_BACKEND_MFMA || validate_conv(&p->a) != 0 || validate_conv(&p->b) != 0) return 0;
    // the dispatch order of fpd_conv_forward_pair(): streaming pair, persistent pair, halo-tile pair, two single launches
    if (fpd_conv_c1_fold_ok(p->a, &p->b)) return 1;
    if (fpd_conv_c3_fold_ok(p->a, &p->b)) return 1;
    if (fpd_conv_pp_fold_ok(p->a, &p->b)) return 1;
    const int r = fpd_conv_tile_pair_fold_ok(p->a, p->b);
    if (r >= 0) return r;
    return (fpd_conv_fold_supported(&p->a) && fpd_conv_fold_supported(&p->b)) ? 1 : 0;
}

int fpd_conv_fused_wgrad_partials(const fpd_conv_t* a) {
    if (!a || g_fpd_backend != FPD_BACKEND_MFMA || validate_conv(a) != 0) return 0;
    const int n = fpd_conv_c1_wgrad_partials(*a);      // (-1: the streaming kernel does not take this launch)
    return n >= 0 ? n : fpd_conv_pp_wgrad_partials(*a);
}
int fpd_conv_pair_fused_wgrad_partials(const fpd_conv_pair_t* p, int32_t* n_a, int32_t* n_b) {
    FPD_REQUIRE(p && n_a && n_b, "conv_pair_fused_wgrad_partials: null pointer");
    *n_a = *n_b = 0;
    if (g_fpd_backend != FPD_BACKEND_MFMA || validate_conv(&p->a) != 0 || validate_conv(&p->b) != 0) return 0;
    int na = 0, nb = 0;
    if (fpd_conv_c1_pair_wgrad_partials(p->a, p->b, &na, &nb) < 0) fpd_conv_pp_pair_wgrad_partials(p->a, p->b, &na, &nb);
    *n_a = na; *n_b = nb;
    return 0;
}

int fpd_conv_f8_in_domain(const fpd_conv_t* c) { return (c && fpd_conv_f8_domain(*c)) ? 1 : 0; }

int fpd_conv_forward_f8(const fpd_conv_f8_t* a, fpd_stream_t stream) {
    FPD_REQUIRE(a, "conv_f8: null pointer");
    int rc = validate_conv(&a->c);
    if (rc) return rc;
    FPD_REQUIRE(a->w8 && a->w8_scale, "conv_f8: null fp8 weights / scales");
    FPD_REQUIRE(((uintptr_t)a->w8 & 15) == 0, "conv_f8: w8 must be 16-byte aligned");
    rc = 1;
    if (g_fpd_backend == FPD_BACKEND_MFMA) rc = fpd_conv_tile_f8_launch(a->c, a->w8, a->w8_scale, (hipStream_t)stream);
    if (rc == 1) rc = dispatch_conv(&a->c, (hipStream_t)stream);      // outside the fp8 domain: the bf16 weights
    return rc ? rc : check_launch();
}

int fpd_weight_quant_f8(const fpd_wquant_entry_t* t, int32_t n, fpd_stream_t stream) {
    FPD_REQUIRE(n == 0 || t != nullptr, "weight_quant_f8: null table");
    int rc = fpd_weight_quant_f8_launch(t, n, (hipStream_t)stream);
    return rc ? rc : check_launch();
}

int fpd_conv_forward_pair(const fpd_conv_pair_t* p, fpd_stream_t stream) {
    FPD_REQUIRE(p, "conv_pair: null pointer");
    int rc = validate_conv(&p->a);
    if (rc) return rc;
    rc = validate_conv(&p->b);
    if (rc) return rc;
    hipStream_t st = (hipStream_t)stream;
    rc = 1;
    if (g_fpd_backend == FPD_BACKEND_MFMA) rc = fpd_conv_c1_pair_launch(p->a, p->b, st);
    if (rc == 1 && g_fpd_backend == FPD_BACKEND_MFMA) rc = fpd_conv_c3_pair_launch(p->a, p->b, st);
    if (rc == 1 && g_fpd_backend == FPD_BACKEND_MFMA) rc = fpd_conv_pp_pair_launch(p->a, p->b, st);
    FPD_REQUIRE(rc != 1 || (p->a.wg_partial == nullptr && p->b.wg_partial == nullptr),
                "conv_pair: fused weight gradients need the persistent kernel; fpd_conv_pair_fused_wgrad_partials() reports 0 for this launch");
    if (rc == 1 && g_fpd_backend == FPD_BACKEND_MFMA) rc = fpd_conv_tile_pair_launch(p->a, p->b, st);
    if (rc == 1) {                       // not pairable: same result from two launches
        rc = dispatch_conv(&p->a, st);
        if (rc == 0) rc = dispatch_conv(&p->b, st);
    }
    return rc ? rc : check_launch();
}

static int validate_bneck_bns(const fpd_bneck_t* a) {
    const fpd_bn_t* bns[3] = {&a->bn1, &a->bn2, &a->bn3};
    for (int i = 0; i < 3; ++i)
        FPD_REQUIRE(bns[i]->mode == FPD_BN_EVAL && bns[i]->relu && bns[i]->gamma && bns[i]->beta && bns[i]->running_mean &&
                    bns[i]->running_var, "bottleneck: bn%d must be an eval-mode BN+ReLU with running statistics", i + 1);
    return 0;
}

int fpd_bottleneck_fold(const fpd_bneck_t* a, fpd_stream_t stream) {
    FPD_REQUIRE(a && a->folded, "bottleneck_fold: null pointer");
    int rc = validate_bneck_bns(a);
    if (rc) return rc;
    rc = fpd_bneck_fold_launch(*a, const_cast<float*>(a->folded), (hipStream_t)stream);
    if (rc == 1) return fpd_fail(-3, "bottleneck_fold: C=%d P=%d is outside the fused kernel's domain", a->C, a->P);
    return rc ? rc : check_launch();
}

int fpd_bottleneck_forward_pair(const fpd_bneck_pair_t* p, fpd_stream_t stream) {
    FPD_REQUIRE(p, "bottleneck_pair: null pointer");
    const fpd_bneck_t* ab[2] = {&p->a, &p->b};
    for (const fpd_bneck_t* a : ab) {
        FPD_REQUIRE(a->x && a->y && a->w1 && a->w2 && a->w3 && a->x != a->y, "bottleneck_pair: null pointer or y aliases x");
        FPD_REQUIRE((int64_t)a->N * a->H * a->W * a->C < ((int64_t)1 << 31), "bottleneck_pair: tensor too large for 32-bit indexing");
        int rc = validate_bneck_bns(a);
        if (rc) return rc;
    }
    int rc = fpd_bneck_fused_pair_launch(p->a, p->b, (hipStream_t)stream);
    if (rc == 1) {                       // not pairable: same result from two launches
        rc = fpd_bottleneck_forward(&p->a, stream);
        return rc ? rc : fpd_bottleneck_forward(&p->b, stream);
    }
    return rc ? rc : check_launch();
}

int fpd_bottleneck_forward(const fpd_bneck_t* a, fpd_stream_t stream) {
    FPD_REQUIRE(a && a->x && a->y && a->w1 && a->w2 && a->w3, "bottleneck: null pointer");
    FPD_REQUIRE(a->x != a->y, "bottleneck: y must not alias x");
    FPD_REQUIRE(a->N > 0 && a->H > 0 && a->W > 0 && a->C > 0 && a->P > 0, "bottleneck: bad dims");
    FPD_REQUIRE((int64_t)a->N * a->H * a->W * a->C < ((int64_t)1 << 31), "bottleneck: tensor too large for 32-bit indexing");
    int rc = validate_bneck_bns(a);
    if (rc) return rc;
    rc = fpd_bneck_fused_launch(*a, (hipStream_t)stream);
    if (rc == 1)
        return fpd_fail(-3, "bottleneck: shape N=%d H=%d W=%d C=%d P=%d dtype=%d is outside the fused kernel's domain",
                        a->N, a->H, a->W, a->C, a->P, a->dtype);
    return rc ? rc : check_launch();
}

int fpd_conv_wgrad(const fpd_wgrad_t* a, fpd_stream_t stream) {
    FPD_REQUIRE(a && a->x && a->dy && a->dw, "wgrad: null pointer");
    int rc = validate_conv_dims(a->N, a->H, a->W, a->C, a->K, a->R, a->S, a->stride, a->pad, a->P, a->Q);
    if (rc) return rc;
    FPD_REQUIRE(a->dtype == FPD_F32 || a->dtype == FPD_BF16, "wgrad: bad dtype %d", a->dtype);
    hipStream_t st = (hipStream_t)stream;
    rc = 1;
    FPD_REQUIRE(a->partial == nullptr || a->partial_stride >= (int64_t)a->K * a->R * a->S * a->C + a->K,
                "wgrad: partial_stride %lld smaller than weight + bias", (long long)a->partial_stride);
    if (g_fpd_backend == FPD_BACKEND_MFMA) rc = fpd_wgrad_tile_launch(*a, st);
    FPD_REQUIRE(!(g_wgrad_tile_only && rc == 1), "wgrad: option wgrad_tile_only is set and the halo-tile kernel declines N=%d H=%d W=%d C=%d K=%d R=%d", a->N, a->H, a->W, a->C, a->K, a->R);
    if (rc == 1 && g_fpd_backend != FPD_BACKEND_NAIVE) rc = fpd_wgrad_mfma_launch(*a, st);
    if (rc == 1 && g_fpd_backend != FPD_BACKEND_NAIVE) rc = fpd_wgrad_smallc_launch(*a, st);
    if (rc == 1) rc = fpd_wgrad_naive_launch(*a, st);
    return rc ? rc : check_launch();
}

/* slabs the kernel fpd_conv_wgrad() dispatches these dimensions to will write (same dispatch order as above) */
int fpd_wgrad_num_partials(const fpd_wgrad_t* a) {
    if (!a) return 0;
    int n = 0;
    if (g_fpd_backend == FPD_BACKEND_MFMA) n = fpd_wgrad_tile_partials(*a);
    if (n == 0 && g_fpd_backend != FPD_BACKEND_NAIVE) n = fpd_wgrad_mfma_partials(*a);
    if (n == 0 && g_fpd_backend != FPD_BACKEND_NAIVE) n = fpd_wgrad_smallc_partials(*a);
    if (n == 0) n = fpd_wgrad_naive_partials(*a);
    return n;
}

int fpd_stem_wgrad_num_partials(const fpd_stem_t* a) {
    if (!a) return 0;
    int n = 0;
    if (g_fpd_backend == FPD_BACKEND_MFMA) n = fpd_stem_wgrad_s2d_partials(*a);
    if (n == 0 && g_fpd_backend == FPD_BACKEND_MFMA) n = fpd_stem_wgrad_mfma_partials(*a);
    if (n == 0) n = fpd_stem_wgrad_partials(*a);
    return n;
}

int fpd_wgrad_reduce(const fpd_wreduce_entry_t* t, int32_t n, int64_t max_elems, fpd_stream_t stream) {
    int rc = fpd_wreduce_launch(t, n, max_elems, (hipStream_t)stream);
    return rc ? rc : check_launch();
}

int fpd_stem_forward(const fpd_stem_t* a, fpd_stream_t stream) {
    FPD_REQUIRE(a && a->x && a->w && a->bias && a->y, "stem: null pointer");
    FPD_REQUIRE(a->P == (a->H + 6 - 7) / 2 + 1 && a->Q == (a->W + 6 - 7) / 2 + 1, "stem: bad output size");
    int rc = 1;
    if (g_fpd_backend == FPD_BACKEND_MFMA) rc = fpd_stem_forward_s2d_launch(*a, (hipStream_t)stream);      // space-to-depth 4x4 form (round 5)
    if (rc == 1 && g_fpd_backend == FPD_BACKEND_MFMA) rc = fpd_stem_forward_mfma_launch(*a, (hipStream_t)stream);
    if (rc == 1) rc = fpd_stem_forward_launch(*a, (hipStream_t)stream);
    return rc ? rc : check_launch();
}

int fpd_stem_wgrad(const fpd_stem_t* a, fpd_stream_t stream) {
    FPD_REQUIRE(a && a->x && a->dy && a->dw, "stem wgrad: null pointer");
    FPD_REQUIRE(a->partial == nullptr || a->partial_stride >= (int64_t)a->K * 148, "stem wgrad: partial_stride %lld smaller than weight + bias",
                (long long)a->partial_stride);
    int rc = 1;
    if (g_fpd_backend == FPD_BACKEND_MFMA) rc = fpd_stem_wgrad_s2d_launch(*a, (hipStream_t)stream);
    if (rc == 1 && g_fpd_backend == FPD_BACKEND_MFMA) rc = fpd_stem_wgrad_mfma_launch(*a, (hipStream_t)stream);
    if (rc == 1) rc = fpd_stem_wgrad_launch(*a, (hipStream_t)stream);
    return rc ? rc : check_launch();
}

int fpd_affsum(const fpd_affsum_t* a, fpd_stream_t stream) {
    FPD_REQUIRE(a && a->y, "affsum: null pointer");
    FPD_REQUIRE(a->nterms >= 1 && a->nterms <= FPD_AFFSUM_MAX, "affsum: %d terms (1..%d supported)", a->nterms, FPD_AFFSUM_MAX);
    FPD_REQUIRE(a->dtype == FPD_F32 || a->dtype == FPD_BF16, "affsum: bad dtype %d", a->dtype);
    for (int j = 0; j < a->nterms; ++j) {
        const int up = a->t[j].up;
        FPD_REQUIRE(a->t[j].x, "affsum: term %d has no input", j);
        FPD_REQUIRE((up == 1 || up == 2 || up == 4 || up == 8) && a->H % up == 0 && a->W % up == 0,
                    "affsum: term %d: up-sampling factor %d does not divide %dx%d", j, up, a->H, a->W);
        const fpd_bn_t& bn = a->t[j].bn;
        FPD_REQUIRE(bn.mode == FPD_BN_NONE || (bn.gamma && bn.beta), "affsum: term %d: BN without gamma/beta", j);
        FPD_REQUIRE(bn.mode != FPD_BN_TRAIN || bn.stats, "affsum: term %d: train-mode BN without statistics", j);
        FPD_REQUIRE(bn.mode != FPD_BN_EVAL || (bn.running_mean && bn.running_var), "affsum: term %d: eval-mode BN without running stats", j);
    }
    int rc = fpd_affsum_launch(*a, (hipStream_t)stream);
    return rc ? rc : check_launch();
}

int fpd_elementwise(const fpd_ew_t* a, fpd_stream_t stream) {
    FPD_REQUIRE(a && (a->y || a->op == FPD_EW_BNRELU_BWD_R), "elementwise: null pointer");
    int rc = fpd_elementwise_launch(*a, (hipStream_t)stream);
    return rc ? rc : check_launch();
}

static int validate_head(const fpd_head_t* a) {
    FPD_REQUIRE(a && a->y0 && a->score && a->w_fc && a->w_score, "head: null pointer");
    FPD_REQUIRE(a->next == nullptr || (a->x && a->w_fc2 && a->w_score2 && a->next != a->x && a->next != a->y0),
                "head: next needs x, w_fc2, w_score2 and must not alias an input");
    FPD_REQUIRE(a->N > 0 && a->H > 0 && a->W > 0 && (int64_t)a->N * a->H * a->W * a->C < ((int64_t)1 << 31), "head: bad dims");
    FPD_REQUIRE(a->bn.mode == FPD_BN_EVAL && a->bn.relu && a->bn.gamma && a->bn.beta && a->bn.running_mean && a->bn.running_var,
                "head: bn must be an eval-mode BN+ReLU with running statistics");
    return 0;
}
int fpd_head_forward(const fpd_head_t* a, fpd_stream_t stream) {
    int rc = validate_head(a);
    if (rc) return rc;
    rc = fpd_head_fused_launch(*a, (hipStream_t)stream);
    if (rc == 1) return fpd_fail(-3, "head: C=%d J=%d dtype=%d is outside the fused kernel's domain", a->C, a->J, a->dtype);
    return rc ? rc : check_launch();
}
int fpd_head_fold(const fpd_head_t* a, fpd_stream_t stream) {
    FPD_REQUIRE(a && a->folded, "head_fold: null pointer");
    FPD_REQUIRE(a->bn.mode == FPD_BN_EVAL && a->bn.gamma && a->bn.beta && a->bn.running_mean && a->bn.running_var, "head_fold: bn");
    int rc = fpd_head_fold_launch(*a, const_cast<float*>(a->folded), (hipStream_t)stream);
    if (rc == 1) return fpd_fail(-3, "head_fold: C=%d is outside the fused kernel's domain", a->C);
    return rc ? rc : check_launch();
}

int fpd_pck(const fpd_pck_t* a, fpd_stream_t stream) {
    FPD_REQUIRE(a && a->out && a->target && a->counts && a->log && a->cursor, "pck: null pointer");
    FPD_REQUIRE(a->B > 0 && a->J > 0 && a->H > 0 && a->W > 0 && a->log_slots > 0, "pck: bad dims");
    FPD_REQUIRE(a->dtype == FPD_F32 || a->dtype == FPD_BF16, "pck: bad dtype %d", a->dtype);
    int rc = fpd_pck_launch(*a, (hipStream_t)stream);
    return rc ? rc : check_launch();
}

int fpd_flip_w(const float* x, float* y, int64_t rows, int32_t W, fpd_stream_t stream) {
    FPD_REQUIRE(x && y && x != y, "flip_w: null or aliased pointer");
    FPD_REQUIRE(rows > 0 && W > 0, "flip_w: bad dims");
    int rc = fpd_flip_w_launch(x, y, rows, W, (hipStream_t)stream);
    return rc ? rc : check_launch();
}
int fpd_flip_merge(const fpd_flipmerge_t* a, fpd_stream_t stream) {
    FPD_REQUIRE(a && a->b && a->y && a->b != a->y, "flip_merge: null or aliased pointer");
    FPD_REQUIRE(a->N > 0 && a->H > 0 && a->W > 0 && a->J > 0 && a->J <= FPD_MAX_JOINTS, "flip_merge: bad dims (J <= %d)", FPD_MAX_JOINTS);
    for (int j = 0; j < a->J; ++j) FPD_REQUIRE(a->src[j] >= 0 && a->src[j] < a->J, "flip_merge: src[%d] = %d out of range", j, a->src[j]);
    int rc = fpd_flip_merge_launch(*a, (hipStream_t)stream);
    return rc ? rc : check_launch();
}
int fpd_final_preds(const fpd_finalpreds_t* a, fpd_stream_t stream) {
    FPD_REQUIRE(a && a->hm && a->coords && a->maxvals, "final_preds: null pointer");
    FPD_REQUIRE((a->trans == nullptr) == (a->preds == nullptr), "final_preds: trans and preds go together");
    FPD_REQUIRE(a->N > 0 && a->H > 0 && a->W > 0 && a->J > 0, "final_preds: bad dims");
    int rc = fpd_final_preds_launch(*a, (hipStream_t)stream);
    return rc ? rc : check_launch();
}
int fpd_render_targets(const fpd_targets_t* a, fpd_stream_t stream) {
    FPD_REQUIRE(a && a->joints && a->vis && a->g && a->target && a->weight, "render_targets: null pointer");
    FPD_REQUIRE(a->B > 0 && a->J > 0 && a->H > 0 && a->W > 0 && a->patch > 0 && (a->patch & 1), "render_targets: bad dims");
    FPD_REQUIRE(a->stride_x > 0 && a->stride_y > 0, "render_targets: bad stride");
    int rc = fpd_render_targets_launch(*a, (hipStream_t)stream);
    return rc ? rc : check_launch();
}
int fpd_warp_affine(const fpd_warp_t* a, fpd_stream_t stream) {
    FPD_REQUIRE(a && a->src && a->out, "warp_affine: null pointer");
    FPD_REQUIRE(a->B > 0 && a->B <= 65535 && a->H > 0 && a->W > 0, "warp_affine: bad dims");
    for (int c = 0; c < 3; ++c) FPD_REQUIRE(a->std[c] != 0.f, "warp_affine: std[%d] == 0", c);
    int rc = fpd_warp_affine_launch(*a, (hipStream_t)stream);
    return rc ? rc : check_launch();
}

int fpd_elementwise_pair(const fpd_ew_pair_t* p, fpd_stream_t stream) {
    FPD_REQUIRE(p && p->a.y && p->b.y, "elementwise_pair: null pointer");
    int rc = fpd_elementwise_pair_launch(p->a, p->b, (hipStream_t)stream);
    if (rc == 1) {
        rc = fpd_elementwise_launch(p->a, (hipStream_t)stream);
        if (rc == 0) rc = fpd_elementwise_launch(p->b, (hipStream_t)stream);
    }
    return rc ? rc : check_launch();
}

int fpd_loss(const fpd_loss_t* a, fpd_stream_t stream) {
    FPD_REQUIRE(a && a->teacher && a->target && a->weight && a->losses, "loss: null pointer");
    int rc = fpd_loss_launch(*a, (hipStream_t)stream);
    return rc ? rc : check_launch();
}

int fpd_adam(const fpd_adam_t* a, fpd_stream_t stream) {
    FPD_REQUIRE(a && a->param && a->grad && a->m && a->v && a->n >= 0, "adam: null pointer");
    int rc = fpd_adam_launch(*a, (hipStream_t)stream);
    return rc ? rc : check_launch();
}

int fpd_weight_prep(const fpd_wprep_entry_t* t, int32_t n, int64_t max_elems, int32_t dtype, fpd_stream_t stream) {
    int rc = fpd_weight_prep_launch(t, n, max_elems, dtype, (hipStream_t)stream);
    return rc ? rc : check_launch();
}
int fpd_bn_update_running(const fpd_bnupd_entry_t* t, int32_t n, fpd_stream_t stream) {
    int rc = fpd_bn_update_running_launch(t, n, (hipStream_t)stream);
    return rc ? rc : check_launch();
}
int fpd_cast(const void* src, void* dst, int64_t n, int32_t sd, int32_t dd, fpd_stream_t stream) {
    int rc = fpd_cast_launch(src, dst, n, sd, dd, (hipStream_t)stream);
    return rc ? rc : check_launch();
}
int fpd_nchw_to_nhwc(const float* src, void* dst, int32_t N, int32_t C, int32_t H, int32_t W, int32_t dtype, fpd_stream_t stream) {
    int rc = fpd_nchw_to_nhwc_launch(src, dst, N, C, H, W, dtype, (hipStream_t)stream);
    return rc ? rc : check_launch();
}
int fpd_nhwc_to_nchw(const void* src, float* dst, int32_t N, int32_t C, int32_t H, int32_t W, int32_t dtype, fpd_stream_t stream) {
    int rc = fpd_nhwc_to_nchw_launch(src, dst, N, C, H, W, dtype, (hipStream_t)stream);
    return rc ? rc : check_launch();
}

// ------------------------------------------------------------------------------------------
// execution plan
// ------------------------------------------------------------------------------------------
struct fpd_op {
    int32_t type;
    union {
        fpd_affsum_t affsum; fpd_layout_t layout; fpd_conv_f8_t conv8; fpd_conv_t conv; fpd_conv_pair_t pair; fpd_bneck_t bneck; fpd_bneck_pair_t bpair; fpd_ew_pair_t epair; fpd_pck_t pck; fpd_head_t head; fpd_wgrad_t wgrad; fpd_stem_t stem; fpd_ew_t ew; fpd_loss_t loss; fpd_adam_t adam;
        fpd_memset_t mset; fpd_table_t table;
    } u;
};
struct fpd_sched {                       // per-op multi-lane schedule (fpd_plan_set_schedule)
    int32_t lane = 0;
    bool record = false;                 // some op of another lane waits for this one
    hipEvent_t done = nullptr;
    std::vector<int32_t> waits;
};
struct fpd_plan {
    std::vector<fpd_op> ops;
    std::vector<fpd_sched> sched;
    std::vector<hipGraphExec_t> graphs;
    std::vector<hipGraph_t> graph_defs;
    hipStream_t lanes[FPD_MAX_LANES] = {};      // [0] unused: lane 0 is the caller's stream
    hipEvent_t lane_done[FPD_MAX_LANES] = {};
    hipEvent_t fork = nullptr;
};

fpd_plan* fpd_plan_create(void) { return new fpd_plan(); }
void fpd_plan_destroy(fpd_plan* p) {
    if (!p) return;
    for (auto g : p->graphs) (void)hipGraphExecDestroy(g);
    for (auto g : p->graph_defs) (void)hipGraphDestroy(g);
    for (auto& s : p->sched) if (s.done) (void)hipEventDestroy(s.done);
    for (int l = 0; l < FPD_MAX_LANES; ++l) {
        if (p->lanes[l]) (void)hipStreamDestroy(p->lanes[l]);
        if (p->lane_done[l]) (void)hipEventDestroy(p->lane_done[l]);
    }
    if (p->fork) (void)hipEventDestroy(p->fork);
    delete p;
}
int fpd_plan_size(const fpd_plan* p) { return p ? (int)p->ops.size() : -1; }

int fpd_plan_add(fpd_plan* p, int32_t op, const void* args, int64_t bytes) {
    FPD_REQUIRE(p && args, "plan_add: null");
    fpd_op o;
    memset(&o, 0, sizeof(o));
    o.type = op;
    size_t want = 0;
    switch (op) {
        case FPD_OP_CONV: want = sizeof(fpd_conv_t); break;
        case FPD_OP_CONV_F8: want = sizeof(fpd_conv_f8_t); break;
        case FPD_OP_WGRAD: want = sizeof(fpd_wgrad_t); break;
        case FPD_OP_CONV_PAIR: want = sizeof(fpd_conv_pair_t); break;
        case FPD_OP_BNECK_PAIR: want = sizeof(fpd_bneck_pair_t); break;
        case FPD_OP_EW_PAIR: want = sizeof(fpd_ew_pair_t); break;
        case FPD_OP_PCK: want = sizeof(fpd_pck_t); break;
        case FPD_OP_HEAD: case FPD_OP_HEAD_FOLD: want = sizeof(fpd_head_t); break;
        case FPD_OP_BNECK: case FPD_OP_BNECK_FOLD: want = sizeof(fpd_bneck_t); break;
        case FPD_OP_STEM_FWD: case FPD_OP_STEM_WGRAD: want = sizeof(fpd_stem_t); break;
        case FPD_OP_EW: want = sizeof(fpd_ew_t); break;
        case FPD_OP_LOSS: want = sizeof(fpd_loss_t); break;
        case FPD_OP_ADAM: want = sizeof(fpd_adam_t); break;
        case FPD_OP_MEMSET: case FPD_OP_NOP: want = sizeof(fpd_memset_t); break;
        case FPD_OP_AFFSUM: want = sizeof(fpd_affsum_t); break;
        case FPD_OP_NCHW2NHWC: want = sizeof(fpd_layout_t); break;
        case FPD_OP_WPREP: case FPD_OP_BNUPD: case FPD_OP_WREDUCE: case FPD_OP_WQUANT: want = sizeof(fpd_table_t); break;
        default: return fpd_fail(-2, "plan_add: unknown op %d", op);
    }
    FPD_REQUIRE((size_t)bytes == want, "plan_add: op %d expects %zu bytes of args, got %lld", op, want, (long long)bytes);
    memcpy(&o.u, args, want);
    p->ops.push_back(o);
    p->sched.emplace_back();
    return (int)p->ops.size() - 1;
}

int fpd_plan_set_schedule(fpd_plan* p, int32_t op, int32_t lane, const int32_t* wait_ops, int32_t n_waits) {
    FPD_REQUIRE(p && op >= 0 && op < (int)p->ops.size(), "plan_set_schedule: bad op index %d", op);
    FPD_REQUIRE(lane >= 0 && lane < FPD_MAX_LANES, "plan_set_schedule: lane %d outside [0,%d)", lane, FPD_MAX_LANES);
    FPD_REQUIRE(n_waits >= 0 && (n_waits == 0 || wait_ops != nullptr), "plan_set_schedule: bad wait list");
    for (int i = 0; i < n_waits; ++i)
        FPD_REQUIRE(wait_ops[i] >= 0 && wait_ops[i] < op, "plan_set_schedule: op %d may only wait for earlier ops (got %d)", op, wait_ops[i]);
    fpd_sched& s = p->sched[op];
    s.lane = lane;
    s.waits.assign(wait_ops, wait_ops + n_waits);
    for (int i = 0; i < n_waits; ++i) p->sched[wait_ops[i]].record = true;
    return 0;
}

int fpd_plan_mark_event(fpd_plan* p, int32_t op) {
    FPD_REQUIRE(p && op >= 0 && op < (int)p->ops.size(), "plan_mark_event: bad op index %d", op);
    p->sched[op].record = true;
    return 0;
}

int fpd_plan_wait_op(fpd_plan* p, int32_t op, fpd_stream_t stream) {
    FPD_REQUIRE(p && op >= 0 && op < (int)p->ops.size(), "plan_wait_op: bad op index %d", op);
    FPD_REQUIRE(p->sched[op].record && p->sched[op].done, "plan_wait_op: op %d has no recorded event (mark it, then run its range)", op);
    FPD_CHECK_HIP(hipStreamWaitEvent((hipStream_t)stream, p->sched[op].done, 0));
    return 0;
}

static void conv_tag(char* p, size_t n, const char* what, const fpd_conv_t& c) {
    snprintf(p, n, "%s N=%d H=%d W=%d C=%d K=%d R=%d s=%d %s%s%s%s", what, c.N, c.H, c.W, c.C, c.K, c.R, c.stride,
             c.epi == FPD_EPI_BNRELU_BWD ? "dgrad" : "fwd", c.bn.mode == FPD_BN_EVAL ? " evalbn" : "", c.wg_partial ? " +wgrad" : "", c.fold_x ? " +fold" : "");
}
// "<plan op index> <what> <shape>": the key tools/profile_summarize.py groups trace rows by
static void set_op_tag(int idx, const fpd_op& o) {
    char t[200] = "";
    static const char* ew_names[] = {"bnrelu_fwd", "bnrelu_bwd_r", "bn_bwd_apply", "maxpool_fwd", "maxpool_bwd", "upadd_fwd", "sumpool", "add", "relu_mask", "dilate2"};
    auto ewn = [&](int op) { return (op >= 0 && op < 10) ? ew_names[op] : "ew"; };
    switch (o.type) {
        case FPD_OP_CONV: conv_tag(t, sizeof(t), "conv", o.u.conv); break;
        case FPD_OP_CONV_F8: conv_tag(t, sizeof(t), "conv_f8", o.u.conv8.c); break;
        case FPD_OP_CONV_PAIR: {
            char a[90], b[90];
            conv_tag(a, sizeof(a), "conv2", o.u.pair.a);
            snprintf(b, sizeof(b), " | H=%d W=%d", o.u.pair.b.H, o.u.pair.b.W);
            snprintf(t, sizeof(t), "%s%s", a, b);
            break;
        }
        case FPD_OP_WGRAD: snprintf(t, sizeof(t), "wgrad N=%d H=%d W=%d C=%d K=%d R=%d s=%d", o.u.wgrad.N, o.u.wgrad.H, o.u.wgrad.W, o.u.wgrad.C, o.u.wgrad.K, o.u.wgrad.R, o.u.wgrad.stride); break;
        case FPD_OP_BNECK: snprintf(t, sizeof(t), "bneck N=%d H=%d W=%d C=%d P=%d", o.u.bneck.N, o.u.bneck.H, o.u.bneck.W, o.u.bneck.C, o.u.bneck.P); break;
        case FPD_OP_BNECK_PAIR: snprintf(t, sizeof(t), "bneck2 N=%d H=%d W=%d C=%d P=%d | H=%d W=%d", o.u.bpair.a.N, o.u.bpair.a.H, o.u.bpair.a.W, o.u.bpair.a.C, o.u.bpair.a.P, o.u.bpair.b.H, o.u.bpair.b.W); break;
        case FPD_OP_HEAD: snprintf(t, sizeof(t), "head N=%d H=%d W=%d C=%d J=%d", o.u.head.N, o.u.head.H, o.u.head.W, o.u.head.C, o.u.head.J); break;
        case FPD_OP_EW: snprintf(t, sizeof(t), "ew %s N=%d H=%d W=%d C=%d", ewn(o.u.ew.op), o.u.ew.N, o.u.ew.H, o.u.ew.W, o.u.ew.C); break;
        case FPD_OP_EW_PAIR: snprintf(t, sizeof(t), "ew2 %s N=%d H=%d W=%d C=%d | H=%d W=%d", ewn(o.u.epair.a.op), o.u.epair.a.N, o.u.epair.a.H, o.u.epair.a.W, o.u.epair.a.C, o.u.epair.b.H, o.u.epair.b.W); break;
        case FPD_OP_STEM_FWD: case FPD_OP_STEM_WGRAD: snprintf(t, sizeof(t), "%s N=%d H=%d W=%d K=%d", o.type == FPD_OP_STEM_FWD ? "stem_fwd" : "stem_wgrad", o.u.stem.N, o.u.stem.H, o.u.stem.W, o.u.stem.K); break;
        case FPD_OP_AFFSUM: snprintf(t, sizeof(t), "affsum N=%d H=%d W=%d C=%d terms=%d", o.u.affsum.N, o.u.affsum.H, o.u.affsum.W, o.u.affsum.C, o.u.affsum.nterms); break;
        case FPD_OP_LOSS: snprintf(t, sizeof(t), "loss B=%d J=%d H=%d W=%d S=%d", o.u.loss.B, o.u.loss.J, o.u.loss.H, o.u.loss.W, o.u.loss.S); break;
        case FPD_OP_ADAM: snprintf(t, sizeof(t), "adam n=%lld", (long long)o.u.adam.n); break;
        case FPD_OP_WREDUCE: snprintf(t, sizeof(t), "wreduce entries=%d", o.u.table.n); break;
        case FPD_OP_WPREP: snprintf(t, sizeof(t), "wprep entries=%d", o.u.table.n); break;
        case FPD_OP_BNUPD: snprintf(t, sizeof(t), "bnupd entries=%d", o.u.table.n); break;
        default: snprintf(t, sizeof(t), "op%d", o.type); break;
    }
    snprintf(g_op_tag, sizeof(g_op_tag), "%d\t%s", idx, t);
}

static int run_op(const fpd_op& o, fpd_stream_t s) {
    switch (o.type) {
        case FPD_OP_CONV: return fpd_conv_forward(&o.u.conv, s);
        case FPD_OP_CONV_F8: return fpd_conv_forward_f8(&o.u.conv8, s);
        case FPD_OP_WQUANT: return fpd_weight_quant_f8((const fpd_wquant_entry_t*)o.u.table.table, o.u.table.n, s);
        case FPD_OP_WGRAD: return fpd_conv_wgrad(&o.u.wgrad, s);
        case FPD_OP_CONV_PAIR: return fpd_conv_forward_pair(&o.u.pair, s);
        case FPD_OP_BNECK_PAIR: return fpd_bottleneck_forward_pair(&o.u.bpair, s);
        case FPD_OP_EW_PAIR: return fpd_elementwise_pair(&o.u.epair, s);
        case FPD_OP_PCK: return fpd_pck(&o.u.pck, s);
        case FPD_OP_HEAD: return fpd_head_forward(&o.u.head, s);
        case FPD_OP_HEAD_FOLD: return fpd_head_fold(&o.u.head, s);
        case FPD_OP_BNECK: return fpd_bottleneck_forward(&o.u.bneck, s);
        case FPD_OP_BNECK_FOLD: return fpd_bottleneck_fold(&o.u.bneck, s);
        case FPD_OP_STEM_FWD: return fpd_stem_forward(&o.u.stem, s);
        case FPD_OP_STEM_WGRAD: return fpd_stem_wgrad(&o.u.stem, s);
        case FPD_OP_EW: return fpd_elementwise(&o.u.ew, s);
        case FPD_OP_LOSS: return fpd_loss(&o.u.loss, s);
        case FPD_OP_ADAM: return fpd_adam(&o.u.adam, s);
        case FPD_OP_MEMSET: {
            FPD_CHECK_HIP(hipMemsetAsync(o.u.mset.ptr, 0, (size_t)o.u.mset.bytes, (hipStream_t)s));
            return 0;
        }
        case FPD_OP_NOP: return 0;
        case FPD_OP_AFFSUM: return fpd_affsum(&o.u.affsum, s);
        case FPD_OP_NCHW2NHWC:
            return fpd_nchw_to_nhwc(o.u.layout.src, o.u.layout.dst, o.u.layout.N, o.u.layout.C, o.u.layout.H, o.u.layout.W, o.u.layout.dtype, s);
        case FPD_OP_WPREP:
            return fpd_weight_prep((const fpd_wprep_entry_t*)o.u.table.table, o.u.table.n, o.u.table.max_elems, o.u.table.dtype, s);
        case FPD_OP_BNUPD: return fpd_bn_update_running((const fpd_bnupd_entry_t*)o.u.table.table, o.u.table.n, s);
        case FPD_OP_WREDUCE: return fpd_wgrad_reduce((const fpd_wreduce_entry_t*)o.u.table.table, o.u.table.n, o.u.table.max_elems, s);
    }
    return fpd_fail(-2, "run_op: unknown op %d", o.type);
}

static int lane_priority() { return 0; }      // side-lane streams at the default priority (stream priorities measured as a no-op eagerly, rounds 1-2; knob removed in round 6)

static int side_streams() { return FPD_MAX_LANES; }      // one physical stream per lane (the FPD_SIDE_STREAMS knob of rounds 1-5 never beat it)

int fpd_plan_run(fpd_plan* p, int32_t begin, int32_t end, fpd_stream_t stream) {
    FPD_REQUIRE(p && begin >= 0 && end <= (int)p->ops.size() && begin <= end, "plan_run: bad range [%d,%d)", begin, end);
    hipStream_t main_s = (hipStream_t)stream;
    bool multi = false;
    for (int i = begin; i < end; ++i) multi |= p->sched[i].lane != 0;
    bool used[FPD_MAX_LANES] = {};
    if (multi) {
        if (!p->fork) FPD_CHECK_HIP(hipEventCreateWithFlags(&p->fork, hipEventDisableTiming));
        FPD_CHECK_HIP(hipEventRecord(p->fork, main_s));
    }
    int rc = 0;
    for (int i = begin; i < end && rc == 0; ++i) {
        fpd_sched& sc = p->sched[i];
        hipStream_t st = main_s;
        if (sc.lane != 0) {
            const int l = 1 + (sc.lane - 1) % side_streams();
            if (!p->lanes[l]) {
                FPD_CHECK_HIP(hipStreamCreateWithPriority(&p->lanes[l], hipStreamNonBlocking, lane_priority()));
                FPD_CHECK_HIP(hipEventCreateWithFlags(&p->lane_done[l], hipEventDisableTiming));
            }
            st = p->lanes[l];
            if (!used[l]) {                      // fork: the lane starts after everything already queued on `stream`
                FPD_CHECK_HIP(hipStreamWaitEvent(st, p->fork, 0));
                used[l] = true;
            }
        }
        for (int32_t w : sc.waits)
            if (w >= begin && w < end) FPD_CHECK_HIP(hipStreamWaitEvent(st, p->sched[w].done, 0));
        if (g_fpd_launch_log) set_op_tag(i, p->ops[i]);
        rc = run_op(p->ops[i], (fpd_stream_t)st);
        if (rc) {
            char tmp[400];
            snprintf(tmp, sizeof(tmp), "%s", g_err);
            rc = fpd_fail(rc, "plan op %d (type %d): %s", i, p->ops[i].type, tmp);
            break;
        }
        if (sc.record) {
            if (!sc.done) FPD_CHECK_HIP(hipEventCreateWithFlags(&sc.done, hipEventDisableTiming));
            FPD_CHECK_HIP(hipEventRecord(sc.done, st));
        }
    }
    // join (also on failure, so that a stream capture in progress can be closed cleanly)
    for (int l = 1; l < FPD_MAX_LANES; ++l)
        if (used[l]) {
            FPD_CHECK_HIP(hipEventRecord(p->lane_done[l], p->lanes[l]));
            FPD_CHECK_HIP(hipStreamWaitEvent(main_s, p->lane_done[l], 0));
        }
    return rc;
}

int fpd_plan_run_op(fpd_plan* p, int32_t op, fpd_stream_t stream) {
    FPD_REQUIRE(p && op >= 0 && op < (int)p->ops.size(), "plan_run_op: bad op index %d", op);
    if (g_fpd_launch_log) set_op_tag(op, p->ops[op]);
    return run_op(p->ops[op], stream);
}

int fpd_plan_capture(fpd_plan* p, int32_t begin, int32_t end, fpd_stream_t stream) {
    FPD_REQUIRE(p, "plan_capture: null");
    // a captured launch is logged once (here) but traced on every replay: the log could no longer be aligned with a trace
    FPD_REQUIRE(!g_fpd_launch_log, "plan_capture: FPD_LAUNCH_LOG is set -- the launch log keys a kernel trace 1:1 and cannot follow hipGraph replays; unset one of them");
    hipStream_t st = (hipStream_t)stream;
    FPD_CHECK_HIP(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
    int rc = fpd_plan_run(p, begin, end, stream);
    hipGraph_t g = nullptr;
    hipError_t e = hipStreamEndCapture(st, &g);
    if (rc) { if (g) (void)hipGraphDestroy(g); return rc; }
    if (e != hipSuccess) return fpd_fail(-100 - (int)e, "hipStreamEndCapture: %s", hipGetErrorString(e));
    hipGraphExec_t ge = nullptr;
    FPD_CHECK_HIP(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    p->graph_defs.push_back(g);
    p->graphs.push_back(ge);
    return (int)p->graphs.size() - 1;
}

int fpd_plan_replay(fpd_plan* p, int32_t gid, fpd_stream_t stream) {
    FPD_REQUIRE(p && gid >= 0 && gid < (int)p->graphs.size(), "plan_replay: bad graph id %d", gid);
    FPD_CHECK_HIP(hipGraphLaunch(p->graphs[gid], (hipStream_t)stream));
    return 0;
}

void* fpd_event_create(void) {
    hipEvent_t e = nullptr;
    if (hipEventCreate(&e) != hipSuccess) return nullptr;
    return (void*)e;
}
int fpd_event_record(void* ev, fpd_stream_t stream) {
    FPD_CHECK_HIP(hipEventRecord((hipEvent_t)ev, (hipStream_t)stream));
    return 0;
}
float fpd_event_elapsed_ms(void* start, void* stop) {
    float ms = -1.f;
    if (hipEventSynchronize((hipEvent_t)stop) != hipSuccess) return -1.f;
    if (hipEventElapsedTime(&ms, (hipEvent_t)start, (hipEvent_t)stop) != hipSuccess) return -1.f;
    return ms;
}
void fpd_event_destroy(void* ev) { if (ev) (void)hipEventDestroy((hipEvent_t)ev); }

}  // extern "C"
