/*
 * fpd_amd.h -- C ABI of the MI355X (gfx950) Fast-Pose-Distillation training path.
 *
 * The reference (ilovepose/fast-human-pose-estimation.pytorch) has NO native/FFI interface on
 * this path: its boundary is the Python module API (lib/models/hourglass.py:170-197,
 * lib/core/loss.py:15-39, lib/core/function.py:99-187) and all arithmetic is torch.nn
 * (cuDNN/ATen).  This header therefore DEFINES the boundary a maintainer would bind from
 * Python (ctypes stub in INTEGRATION.md).  Each entry point names the reference call site
 * whose arithmetic it replaces.
 *
 * Conventions:
 *   - plain pointers + sizes, no torch types; the CALLER owns every buffer (incl. workspaces);
 *   - every call is asynchronous on the given hipStream_t, never synchronises, never allocates;
 *   - return 0 on success, negative on error; fpd_last_error() gives a thread-local message;
 *   - activations are NHWC ("channels last"), dtype FPD_F32 or FPD_BF16; conv weights are
 *     K,R,S,C (the reference's OIHW tensor stored channels-last); statistics are exact fixed-point sums (fpd_stat_t).
 */
#ifndef FPD_AMD_H
#define FPD_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* fpd_stream_t; /* hipStream_t */

enum { FPD_F32 = 0, FPD_BF16 = 1 };
/* Every per-channel statistics buffer (BN forward {sum, sumsq}; BN backward {sum dz, sum dz*xhat}) holds
 * FPD_STATS_REPLICAS independent partial copies.  Producer block b adds into replica b % R (thousands of blocks adding
 * into ONE address serialise at ~20 ns per atomic); consumers sum the replicas.
 * The sums are EXACT and therefore independent of the order in which the blocks' contributions arrive (a training step
 * is bit-repeatable): a block's fp64 partial v is added as two 64-bit INTEGER limbs
 *     hi = rint(v * 2^20)                     units of 2^-20  (|v| clamped to 2^42)
 *     lo = rint((v - hi * 2^-20) * 2^60)      units of 2^-60  (|lo| <= 2^39)
 * with integer atomics (associative, unlike floating-point ones; measured on MI355X: same cost as an fp64 atomic pair),
 * value = hi * 2^-20 + lo * 2^-60.  BOUND: a limb sum is a 64-bit integer, so one (replica, sum, channel) takes up to
 * 2^24 contributions whatever their signs (|lo| <= 2^39 each) and a total of magnitude < 2^42; both are checked nowhere
 * at run time -- the producers of this library add at most 2^17 contributions per channel (ABI version 2; version 1 used
 * hi units of 2^-8, which wrapped after 4 096 worst-case addends).  Layout of one buffer over C channels: fpd_stat_t [R][2 sums][2 limbs: hi, lo][C]
 * (FPD_STATS_WORDS(C) 64-bit words); the caller zeroes it (all-zero bytes = all-zero sums). */
#ifndef FPD_STATS_REPLICAS
#define FPD_STATS_REPLICAS 4
#endif
typedef int64_t fpd_stat_t;
#define FPD_STATS_WORDS(C) ((int64_t)FPD_STATS_REPLICAS * 4 * (C))
enum { FPD_BN_NONE = 0, FPD_BN_TRAIN = 1, FPD_BN_EVAL = 2 };
enum { FPD_EPI_PLAIN = 0, FPD_EPI_BNRELU_BWD = 1 };
enum { FPD_BACKEND_MFMA = 0, FPD_BACKEND_NAIVE = 1, FPD_BACKEND_MFMA_GENERIC = 2 };
/* MFMA: halo-tile kernel where it applies, else the generic MFMA kernel, else the direct kernel;
 * MFMA_GENERIC skips the halo-tile kernel; NAIVE forces the direct kernels (cross-check). */

/* nn.BatchNorm2d(momentum=0.1) (hourglass.py:18-25,118,162) applied on the fly while a consumer
 * loads the tensor: a = relu?(x*scale+shift).  TRAIN: scale/shift derive from the batch
 * statistics `stats` = {sum[C], sumsq[C]} accumulated by the tensor's producer; EVAL: from the
 * running estimates. */
typedef struct {
    int32_t mode;          /* FPD_BN_* */
    int32_t relu;          /* apply ReLU after the affine (hourglass.py:28) */
    float eps;
    int32_t _pad;
    const fpd_stat_t* stats; /* TRAIN: statistics buffer over C: sum, sum of squares over N*H*W (layout: fpd_stat_t above) */
    const float* gamma;    /* [C] */
    const float* beta;     /* [C] */
    const float* running_mean; /* EVAL: [C] */
    const float* running_var;  /* EVAL: [C] */
} fpd_bn_t;

/* nn.Conv2d forward as an implicit GEMM (hourglass.py:20,23,27,116,135-137,143-145,163), with
 * the preceding BN+ReLU fused on the input, and bias / residual add (hourglass.py:50,189) /
 * batch statistics of the result (for the next train-mode BN) fused on the output.
 * The same kernel computes the data gradient of a stride-1 conv when given the flipped,
 * IO-swapped weights; epi = FPD_EPI_BNRELU_BWD then masks the result with the forward ReLU of
 * tensor `epi_x` and accumulates the two BN-backward sums {sum dz, sum dz*xhat} into epi_stats. */
typedef struct {
    int32_t N, H, W, C;    /* input  [N,H,W,C] */
    int32_t K, R, S;       /* output channels, filter height/width */
    int32_t stride, pad;
    int32_t P, Q;          /* output [N,P,Q,K] */
    int32_t dtype;         /* FPD_F32 / FPD_BF16: x, w, residual, y, epi_x */
    int32_t epi;           /* FPD_EPI_* */
    int32_t _pad;
    const void* x;
    const void* w;         /* [K][R][S][C] */
    const float* bias;     /* [K] or NULL */
    const void* residual;  /* [N,P,Q,K] or NULL (may alias y) */
    void* y;
    fpd_stat_t* out_stats; /* statistics buffer over K += {sum y, sum y^2}, or NULL */
    fpd_bn_t bn;           /* prologue on x (mode NONE = raw x) */
    const void* epi_x;     /* BNRELU_BWD: forward tensor normalised by epi_bn, [N,P,Q,K] */
    fpd_bn_t epi_bn;       /* BNRELU_BWD: its (train-mode) BN */
    fpd_stat_t* epi_stats; /* BNRELU_BWD: statistics buffer over K += {sum dz, sum dz*xhat} */
    /* FUSED WEIGHT GRADIENT (optional; BNRELU_BWD data gradient of a 1x1 convolution only): this launch is the data gradient
     * of a forward convolution y = W * a(u), a(u) = relu?(bn(u)); it reads x = dy and epi_x = u anyway, so it can form the
     * forward convolution's weight gradient dW[k][c] = sum_pixels dy[.,k] * a(u)[.,c] (k < C of this launch, c < K of this
     * launch) and bias gradient sum_pixels dy[.,k] on the way, saving the separate fpd_conv_wgrad() launch and its HBM reads.
     * Persistent block range b stores its partial sums to wg_partial + b*wg_stride (layout [C][K] floats, then [C] bias
     * sums when wg_bias); fpd_wgrad_reduce() adds the slabs in order.  The number of slabs is fpd_conv_fused_wgrad_partials()
     * (0 = this shape / configuration is not served: leave wg_partial NULL and call fpd_conv_wgrad()). */
    float* wg_partial;
    int64_t wg_stride;     /* floats between slabs, >= C*K + C */
    int32_t wg_bias;       /* also accumulate the bias gradient */
    int32_t wg_count;      /* slabs the caller provided behind wg_partial = what fpd_conv_fused_wgrad_partials() returned when the
                            * workspace was sized; the launch fails (instead of writing past it) if its geometry has changed since
                            * -- e.g. fpd_set_option("conv_pp_blocks") between the query and the launch */
    /* FOLDED BN-BACKWARD APPLY (optional; BNRELU_BWD data gradients served by the persistent kernel only, bn.mode NONE): x is
     * not the operand itself but the masked gradient g that reached a train-mode BatchNorm whose input was fold_x = u, with
     * the two sums {sum g, sum g*xhat} in fold_stats.  The launch evaluates the BN backward on the way into its operand image,
     *     dy = gamma*invstd * (g - mean(g) - xhat*mean(g*xhat)),   xhat = (u - mean(u)) * invstd,
     * rounds it once (like the stand-alone FPD_EW_BN_BWD_APPLY it replaces: one launch and three tensor passes fewer) and
     * convolves THAT; if fold_out is given the evaluated operand is also written out once ([N,H,W,C], for a separate
     * weight-gradient launch), fold_dgamma / fold_dbeta receive sum g*xhat / sum g.  fpd_conv_fold_supported() tells
     * whether a launch is served (0: leave fold_x NULL and issue the apply). */
    const void* fold_x;
    fpd_bn_t fold_bn;
    const fpd_stat_t* fold_stats;
    void* fold_out;
    float* fold_dgamma;
    float* fold_dbeta;
} fpd_conv_t;

/* A whole pre-activation Bottleneck of a FROZEN network in one launch (hourglass.py:32-52 with eval-mode BN, no
 * downsample branch):  y = x + conv3(relu(bn3(conv2(relu(bn2(conv1(relu(bn1(x)))))))))  with conv1 1x1 C->P,
 * conv2 3x3 P->P pad 1, conv3 1x1 P->C.  Both P-channel intermediates stay in LDS (rounded to bf16 once, after
 * bias+BN+ReLU).  Domain: dtype BF16, C == 2P, P in {64,128}, W a power of two in [4,64], H*W a multiple or a divisor
 * of 128, every BN in EVAL mode; fpd_bottleneck_forward() returns an error outside it (callers then issue the three
 * fpd_conv_forward() calls). */
typedef struct {
    int32_t N, H, W, C, P, dtype;
    int32_t _pad[2];
    const void* x;         /* [N,H,W,C] */
    void* y;               /* [N,H,W,C] (must not alias x: neighbouring tiles re-read x rows) */
    const void* w1;        /* [P][1][1][C] working-precision weights */
    const float* b1;       /* [P] or NULL */
    const void* w2;        /* [P][3][3][P] */
    const float* b2;
    const void* w3;        /* [C][1][1][P] */
    const float* b3;       /* [C] or NULL */
    fpd_bn_t bn1, bn2, bn3;
    /* optional [3C + 4P] floats written by fpd_bottleneck_fold(): the three BNs folded to scale/shift with the conv1/conv2
     * biases folded into the following shift, + b3.  A frozen model folds once; NULL = every block folds for itself. */
    const float* folded;
} fpd_bneck_t;

/* The inter-stack head of a FROZEN hourglass stack in one launch (hourglass.py:134-137,184-190 with eval-mode BN):
 *   a = relu(bn(fc(y0)));  score = score_conv(a);  next = x + fc_(a) + score_(score)        (all 1x1 convolutions)
 * `a` stays in LDS (rounded to bf16 once); score is rounded to bf16 once (it is an output) and reused from LDS.
 * next == NULL (last stack): only score is produced and w_fc2 / w_score2 / x may be NULL.
 * Domain: dtype BF16, C == 256, J == 16; fpd_head_forward() returns an error outside it. */
typedef struct {
    int32_t N, H, W, C, J, dtype;
    int32_t _pad[2];
    const void* y0;        /* [N,H,W,C] output of the stack's last Bottleneck */
    const void* x;         /* [N,H,W,C] the stack's input (residual), or NULL */
    void* score;           /* [N,H,W,J] */
    void* next;            /* [N,H,W,C] input of the next stack, or NULL */
    const void* w_fc;      /* [C][1][1][C] */
    const float* b_fc;
    const void* w_score;   /* [J][1][1][C] */
    const float* b_score;
    const void* w_fc2;     /* fc_: [C][1][1][C] */
    const float* b_fc2;
    const void* w_score2;  /* score_: [C][1][1][J] */
    const float* b_score2;
    fpd_bn_t bn;           /* fc.1, EVAL */
    const float* folded;   /* optional [3C+32] floats written by fpd_head_fold() */
} fpd_head_t;
int fpd_head_forward(const fpd_head_t* a, fpd_stream_t stream);
int fpd_head_fold(const fpd_head_t* a, fpd_stream_t stream);

/* Two INDEPENDENT convolutions issued as one launch (the parallel up-/low-branch bottlenecks of an hourglass level,
 * hourglass.py:80-88, have identical channel shapes at full and half resolution): fpd_conv_forward_pair() runs them in a
 * single kernel when both are in the halo-tile domain with equal dtype/C/K/R, else one after the other. */
typedef struct { fpd_conv_t a, b; } fpd_conv_pair_t;

/* Two independent frozen Bottlenecks as one launch (same P; else / outside the domain: an error, see fpd_bneck_t). */
typedef struct { fpd_bneck_t a, b; } fpd_bneck_pair_t;

/* Weight + bias gradient of the same conv (autograd of hourglass.py convs): dw[K][R][S][C] +=
 * sum_pixels dy * a(x), dbias[K] += sum_pixels dy; the caller zeroes dw/dbias.  DETERMINISTIC: no floating-point
 * atomics -- every sum is formed in a fixed order, so two runs on the same inputs give identical bytes. */
typedef struct {
    int32_t N, H, W, C, K, R, S, stride, pad, P, Q, dtype;
    const void* x;         /* forward input (pre-BN) */
    const void* dy;        /* [N,P,Q,K] */
    float* dw;             /* [K][R][S][C] fp32 */
    float* dbias;          /* [K] or NULL */
    fpd_bn_t bn;           /* forward prologue, recomputed */
    /* two-stage reduction (what a training plan uses): block / pixel chunk b stores its accumulator to
     * partial + b*partial_stride (layout of dw, bias partials behind it); fpd_wgrad_reduce() adds the slabs to dw in slab
     * order afterwards.  The number of slabs written for these dimensions is fpd_wgrad_num_partials() (>= 1 for every
     * supported shape).  partial == NULL: the launch uses ONE block per output element group, which adds into dw
     * directly -- same result, no workspace, but no parallelism over the pixel axis (small problems / tests). */
    float* partial;
    int64_t partial_stride; /* floats between slabs, >= K*R*S*C + K (bias partials sit behind the weights) */
} fpd_wgrad_t;

/* Stem: Conv2d(3, K, 7, stride 2, pad 3) (hourglass.py:116,172) reading the fp32 NCHW image. */
typedef struct {
    int32_t N, H, W, K, P, Q, dtype, _pad;
    const float* x;        /* [N,3,H,W] fp32 NCHW */
    const float* w;        /* [K][7][7][3] fp32 */
    const float* bias;     /* [K] */
    void* y;               /* [N,P,Q,K] */
    fpd_stat_t* out_stats; /* statistics buffer over K, or NULL */
    const void* dy;        /* wgrad: [N,P,Q,K] */
    float* dw;             /* wgrad: [K][7][7][3] */
    float* dbias;          /* wgrad: [K] */
    float* partial;        /* wgrad: slabs of the two-stage reduction (see fpd_wgrad_t), fpd_stem_wgrad_num_partials() of them, or NULL */
    int64_t partial_stride; /* floats between slabs, >= K*147 + K */
} fpd_stem_t;

/* Elementwise / pooling ops on NHWC tensors.  `op` selects the function:
 *  BNRELU_FWD   y = relu(bn(x)); out_stats += stats(y)                  (hourglass.py:173-174)
 *  BNRELU_BWD_R dz = dy * (yfwd > 0) -> y ; bstats += {sum dz, sum dz*xhat(x)}   (yfwd = relu(bn(x)), or x2 when given: the
 *               forward output of an HRNet block tail relu(bn(conv) + skip), pose_hrnet.py:52-57 -- mask and sums in one pass)
 *  BN_BWD_APPLY y = add + gamma*invstd*(dz - s1/n - xhat(x)*s2/n)       (BatchNorm backward; dz in `dy`)
 *  MAXPOOL_FWD  y = maxpool2x2(x); out_stats += stats(y)               (hourglass.py:82,177)
 *  MAXPOOL_BWD  y = add + (x is the first max of its window ? dy : 0)
 *  UPADD_FWD    y = x + nearest_up2(x2); out_stats += stats(y)         (hourglass.py:90-91)
 *  SUMPOOL      y = add + sum2x2(x)                                    (Upsample backward)
 *  ADD          y = x + x2
 *  RELU_MASK    y = dy * (x > 0)   (x = the forward OUTPUT of a ReLU; pose_hrnet.py:55,96,261 autograd)
 *  DILATE2      y[n,2p,2q,:] = x[n,p,q,:], zero elsewhere; N,H,W = shape of y (even H,W).  The data gradient of a
 *               stride-2 convolution (pose_hrnet.py:223-242,349-372) is the stride-1 convolution of this tensor.
 * BNRELU_BWD_R with y == NULL accumulates the two sums only.
 */
enum {
    FPD_EW_BNRELU_FWD = 0, FPD_EW_BNRELU_BWD_R = 1, FPD_EW_BN_BWD_APPLY = 2, FPD_EW_MAXPOOL_FWD = 3,
    FPD_EW_MAXPOOL_BWD = 4, FPD_EW_UPADD_FWD = 5, FPD_EW_SUMPOOL = 6, FPD_EW_ADD = 7, FPD_EW_RELU_MASK = 8,
    FPD_EW_DILATE2 = 9
};
typedef struct {
    int32_t op, dtype;
    int32_t N, H, W, C;    /* shape of the LARGER spatial tensor involved (pool input / up output) */
    const void* x;         /* see table */
    const void* x2;
    const void* dy;
    const void* add;       /* optional accumulate source (may alias y) */
    void* y;
    fpd_stat_t* out_stats; /* statistics buffer over C, or NULL */
    fpd_stat_t* bstats;    /* BN-backward sums: statistics buffer over C */
    float* dgamma;         /* BN_BWD_APPLY: optional [C] <- sum dz*xhat (grad of BN weight) */
    float* dbeta;          /* BN_BWD_APPLY: optional [C] <- sum dz      (grad of BN bias) */
    fpd_bn_t bn;
} fpd_ew_t;

/* y = relu?( sum_j a_j(up_j(x_j)) ): the tail of an HRNet block (pose_hrnet.py:52-57,93-98: relu(bn(conv) + skip)), a fuse
 * layer (:252-265: relu(sum of identity / 1x1-conv+BN+nearest-up / strided-conv+BN terms)) or a transition output (:349-372)
 * in ONE pass.  Term j is x_j[n, h/up_j, w/up_j, :] (nearest up-sampling by up_j in {1,2,4,8}), normalised by bn_j (mode
 * NONE = identity; TRAIN statistics cover the N*(H/up)*(W/up) pixels of x_j; bn_j.relu applies to the term).  The sum is
 * rounded to the storage type once. */
#define FPD_AFFSUM_MAX 4
typedef struct { const void* x; fpd_bn_t bn; int32_t up; int32_t _pad; } fpd_affterm_t;
typedef struct {
    int32_t N, H, W, C;    /* output shape [N,H,W,C] */
    int32_t dtype, relu, nterms, _pad;
    fpd_affterm_t t[FPD_AFFSUM_MAX];
    void* y;
} fpd_affsum_t;
int fpd_affsum(const fpd_affsum_t* a, fpd_stream_t stream);

/* [N,C,H,W] fp32 image -> [N,H,W,C] activation as a plan op (HRNet stem: its first convolution is a generic NHWC conv) */
typedef struct { const float* src; void* dst; int32_t N, C, H, W, dtype, _pad; } fpd_layout_t;

/* JointsMSELoss pose + KD over all stacks, forward and backward in one pass
 * (lib/core/loss.py:21-39 called 2*S times at lib/core/function.py:128-134):
 *   pose = sum_s 0.5/(B*J*hw) sum w^2 (p_s-g)^2 ; kd likewise against the teacher map t;
 *   dout_s = w^2 [(1-alpha)(p_s-g) + alpha (p_s-t)] / (B*J*hw).
 * use_target_weight=False (loss.py:30-37) = a weight buffer of ones. */
#define FPD_MAX_STACKS 8
typedef struct {
    int32_t B, J, H, W, S, dtype;
    int32_t target_nchw;   /* 1: target is [B,J,H,W] (reference loader layout); 0: [B,H,W,J] */
    float alpha;
    const void* out[FPD_MAX_STACKS];  /* student maps, NHWC [B,H,W,J] */
    void* dout[FPD_MAX_STACKS];       /* gradients, same layout (NULL = forward only) */
    const void* teacher;   /* NHWC [B,H,W,J] in dtype */
    const float* target;   /* fp32 */
    const float* weight;   /* [B,J] fp32 */
    double* losses;        /* [2] += {pose, kd}; caller zeroes */
    float grad_scale;      /* extra factor on dout (1/world for DP averaging), normally 1 */
    int32_t _pad;
    const float* weight_kd; /* optional [B,J]: weights of the distillation term when its criterion's use_target_weight
                             * differs from the pose criterion's (tools/fpd_train.py:145-147,177-179); NULL = `weight` */
} fpd_loss_t;

/* torch.optim.Adam(lr) step over one flat fp32 arena (lib/utils/utils.py:69-73), optionally
 * emitting the bf16 working copy of the parameters. */
typedef struct {
    int64_t n;
    float* param; const float* grad; float* m; float* v;
    void* param_lp;        /* optional bf16 copy [n] */
    float lr, beta1, beta2, eps;
    float bias_corr1, bias_corr2;   /* 1-beta^t */
    float grad_scale;      /* multiply grads first (1/world_size) */
    int32_t _pad;
    const float* lr_dev;   /* optional: read lr from device memory (graph-replay safe) */
    int64_t* step_dev;     /* optional: device step counter t (bias corrections use t+1; incremented after) */
} fpd_adam_t;

/* Per-conv working copies of the weights: cast to `dtype` in K,R,S,C order (forward operand)
 * and the flipped, IO-swapped C,R,S,K copy (data-gradient operand).  One launch for a table. */
typedef struct {
    const float* w;        /* master [K][R][S][C] fp32 */
    void* w_fwd;           /* [K][R][S][C] dtype, or NULL */
    void* w_bwd;           /* [C][R][S][K] dtype with r,s flipped, or NULL */
    int32_t K, R, S, C;
} fpd_wprep_entry_t;

/* Running-statistics update of train-mode BNs after a forward (torch: momentum 0.1, unbiased
 * variance; hourglass.py:10).  One launch for a table. */
typedef struct {
    const fpd_stat_t* stats; /* statistics buffer over C */
    float* running_mean; float* running_var; int64_t* num_batches_tracked;
    double count; float momentum; int32_t C;
} fpd_bnupd_entry_t;

/* dw[i] += sum_{b < count} partial[b*stride + i], i < n : second stage of the weight-gradient reduction, one
 * launch for a table of convolutions. */
typedef struct {
    const float* partial; float* dw; int64_t n; int64_t stride; int32_t count; int32_t _pad;
} fpd_wreduce_entry_t;

/* ---- single-op entry points (asynchronous on `stream`) ---- */
int fpd_conv_forward(const fpd_conv_t* a, fpd_stream_t stream);
int fpd_conv_forward_pair(const fpd_conv_pair_t* p, fpd_stream_t stream);
/* fused frozen Bottleneck (three convs + three eval-mode BN+ReLU + residual), see fpd_bneck_t */
int fpd_bottleneck_forward(const fpd_bneck_t* a, fpd_stream_t stream);
/* writes a->folded (must be non-NULL) from a's BN / bias pointers; rerun whenever those parameters change */
int fpd_bottleneck_fold(const fpd_bneck_t* a, fpd_stream_t stream);
int fpd_bottleneck_forward_pair(const fpd_bneck_pair_t* p, fpd_stream_t stream);
/* slabs the fused weight gradient of data-gradient launch `a` writes (see fpd_conv_t.wg_partial); 0 = not available.  For the
 * two convolutions of a pair launch: fpd_conv_pair_fused_wgrad_partials (counts for p->a and p->b; returns 0 / error code). */
int fpd_conv_fused_wgrad_partials(const fpd_conv_t* a);
/* 1 if fpd_conv_forward() / fpd_conv_forward_pair() would serve these launches WITH a folded BN-backward apply (fold_x set
 * or not: only the dimensions, epilogue and prologue are looked at); 0 otherwise. */
int fpd_conv_fold_supported(const fpd_conv_t* a);
int fpd_conv_pair_fold_supported(const fpd_conv_pair_t* p);
int fpd_conv_pair_fused_wgrad_partials(const fpd_conv_pair_t* p, int32_t* n_a, int32_t* n_b);
int fpd_conv_wgrad(const fpd_wgrad_t* a, fpd_stream_t stream);
int fpd_wgrad_num_partials(const fpd_wgrad_t* a);   /* slabs fpd_conv_wgrad writes when a->partial is set */
int fpd_wgrad_reduce(const fpd_wreduce_entry_t* table_dev, int32_t n_entries, int64_t max_elems, fpd_stream_t stream);
int fpd_stem_forward(const fpd_stem_t* a, fpd_stream_t stream);
int fpd_stem_wgrad(const fpd_stem_t* a, fpd_stream_t stream);
int fpd_stem_wgrad_num_partials(const fpd_stem_t* a);   /* slabs fpd_stem_wgrad writes when a->partial is set */
int fpd_elementwise(const fpd_ew_t* a, fpd_stream_t stream);
/* two independent elementwise ops of the same kind in one launch (else one after the other) */
typedef struct { fpd_ew_t a, b; } fpd_ew_pair_t;
int fpd_elementwise_pair(const fpd_ew_pair_t* p, fpd_stream_t stream);
int fpd_loss(const fpd_loss_t* a, fpd_stream_t stream);
int fpd_adam(const fpd_adam_t* a, fpd_stream_t stream);
int fpd_weight_prep(const fpd_wprep_entry_t* table_dev, int32_t n_entries, int64_t max_elems, int32_t dtype,
                    fpd_stream_t stream);
int fpd_bn_update_running(const fpd_bnupd_entry_t* table_dev, int32_t n_entries, fpd_stream_t stream);
int fpd_cast(const void* src, void* dst, int64_t n, int32_t src_dtype, int32_t dst_dtype, fpd_stream_t stream);
/* [N,C,H,W] fp32 <-> [N,H,W,C] dtype layout changes at the module boundary */
int fpd_nchw_to_nhwc(const float* src, void* dst, int32_t N, int32_t C, int32_t H, int32_t W, int32_t dtype,
                     fpd_stream_t stream);
int fpd_nhwc_to_nchw(const void* src, float* dst, int32_t N, int32_t C, int32_t H, int32_t W, int32_t dtype,
                     fpd_stream_t stream);

/* Per-step training metric (lib/core/evaluate.py:16-71 accuracy, lib/core/inference.py:18-46 get_max_preds): heat-map
 * arg-max of prediction and target, PCK@thr over the batch, bit-exact with the reference (first maximum wins; the
 * normaliser is the reference's [h, w]/10 applied to (x, y), evaluate.py:55; float64 distances and averages).  One call
 * appends {avg_acc, cnt, pose_loss, kd_loss} (what the reference passes to its AverageMeters every iteration,
 * function.py:150-155) to the device log ring at slot *cursor % log_slots and advances the cursor -- no host
 * synchronisation; the host drains the ring when it prints a log line. */
typedef struct {
    int32_t B, J, H, W, dtype, log_slots;
    float thr;             /* 0.5 in the reference */
    int32_t _pad;
    const void* out;       /* prediction [B,H,W,J] (NHWC, dtype) */
    const float* target;   /* [B,J,H,W] fp32 (NCHW, as the loader delivers it) */
    float* counts;         /* [J][2] workspace {hits, valid}, zero on entry, zeroed again on exit */
    double* log;           /* [log_slots][4] {avg_acc, cnt, pose, kd} */
    long long* cursor;     /* device counter of appended entries */
    const double* losses;  /* optional [2] {pose, kd} of this iteration (fpd_loss_t.losses), copied into the log entry */
} fpd_pck_t;
int fpd_pck(const fpd_pck_t* a, fpd_stream_t stream);

/* ---- fp8 (OCP e4m3fn) forward convolutions on the CDNA4 fp8 matrix pipe: BASELINE configs[4] "HRNet-W32 student fp8
 * weights (CDNA4 fp8 MFMA)" ---- */
/* fpd_conv_forward() with e4m3 weights: `c` is the convolution exactly as fpd_conv_t describes it (c.w = the bf16
 * working weights, used when the shape is outside the fp8 kernel's domain); w8 = the same weights as e4m3 bytes
 * [K][R][S][C], w8_scale[k] = the fp32 scale of output channel k (w ~= w8 * scale).  The activation operand (after the
 * fused BN+ReLU prologue) is rounded to e4m3 at unit scale, saturating at +-448, on its way into LDS; products
 * accumulate in fp32 (v_mfma_f32_32x32x16_fp8_fp8); scale, bias, residual and the batch statistics follow in fp32.
 * Domain of the fp8 kernel: dtype BF16, plain epilogue, stride-1 "same" 1x1/3x3, W <= 128, C % 32 == 0, K % 8 == 0. */
typedef struct {
    fpd_conv_t c;
    const void* w8;
    const float* w8_scale;
} fpd_conv_f8_t;
int fpd_conv_forward_f8(const fpd_conv_f8_t* a, fpd_stream_t stream);
int fpd_conv_f8_in_domain(const fpd_conv_t* c);   /* 1 = the fp8 kernel takes this shape, 0 = falls back to c.w */

/* fp32 master weights -> e4m3 + per-output-channel scale (amax / 448; 1 for an all-zero row), one launch for a table. */
typedef struct {
    const float* w;        /* [K][RSC] fp32 master weights (K,R,S,C order) */
    void* w8;              /* [K][RSC] e4m3 bytes */
    float* scale;          /* [K] */
    int32_t K, RSC;
} fpd_wquant_entry_t;
int fpd_weight_quant_f8(const fpd_wquant_entry_t* table_dev, int32_t n_entries, fpd_stream_t stream);

/* ---- validate / flip-test post-processing (lib/core/function.py:189-332) ---- */
/* y[r][c] = x[r][W-1-c] for `rows` rows of W floats: the flipped input image of the flip test
 * (function.py:217-221: np.flip(input.cpu().numpy(), 3)); x is [N,3,H,W] fp32, rows = N*3*H. */
int fpd_flip_w(const float* x, float* y, int64_t rows, int32_t W, fpd_stream_t stream);

/* flip_back (lib/utils/transforms.py:15-29) + the one-pixel shift of the flipped map (function.py:233-236,
 * TEST.SHIFT_HEATMAP) + the average (function.py:238) in one pass over fp32 [N,J,H,W] maps:
 *   f[n,j,y,x]  = b[n, src[j], y, W-1-x]               (width reversed, left/right joint channels swapped)
 *   f'[.., x]   = shift && x >= 1 ? f[.., x-1] : f[.., x]
 *   y           = (a + f') * 0.5f
 * src[j] is the channel of b that lands in channel j after flip_back's sequential pair swaps. */
#define FPD_MAX_JOINTS 32
typedef struct {
    int32_t N, J, H, W;
    int32_t shift, _pad;
    const float* a;        /* [N,J,H,W] model(input)[-1]; NULL: y = f' (flip_back + shift only, no average) */
    const float* b;        /* [N,J,H,W] model(flipped input)[-1] */
    float* y;              /* [N,J,H,W] (may alias a, not b) */
    int32_t src[FPD_MAX_JOINTS];
} fpd_flipmerge_t;
int fpd_flip_merge(const fpd_flipmerge_t* a, fpd_stream_t stream);

/* get_final_preds (lib/core/inference.py:49-79): per (sample, joint) the heat-map arg-max (first maximum wins;
 * coordinates zeroed where the maximum is not positive, inference.py:18-46), the quarter-pixel shift towards the higher
 * neighbour when post_process (TEST.POST_PROCESS, inference.py:57-70) and transform_preds (lib/utils/transforms.py:50-55):
 * preds = trans[n] . [x, y, 1] in float64, stored as float32.  trans[n] is the 2x3 inverse affine of sample n
 * (get_affine_transform(center, scale, 0, [W, H], inv=1), transforms.py:57-92 -- six doubles per sample computed by
 * the caller); trans/preds NULL = heat-map coordinates only. */
typedef struct {
    int32_t N, J, H, W;
    int32_t post_process, _pad;
    const float* hm;       /* [N,J,H,W] fp32 */
    const double* trans;   /* [N][2][3] or NULL */
    float* coords;         /* [N,J,2] heat-map coordinates (x, y) after post-processing */
    float* preds;          /* [N,J,2] image coordinates, or NULL */
    float* maxvals;        /* [N,J] */
} fpd_finalpreds_t;
int fpd_final_preds(const fpd_finalpreds_t* a, fpd_stream_t stream);

/* ---- device data pipeline (lib/dataset/JointsDataset.py:113-198,233-289) ---- */
/* generate_target (JointsDataset.py:233-289) for a batch: Gaussian heat-map targets + target weights from joint
 * positions in network-input pixels.  mu = int(joint / feat_stride + 0.5) in float64 (feat_stride = image/heat-map size),
 * the (6*sigma+1)^2 patch `g` (built once by the caller with the reference's float32 numpy expression, so the values are
 * the reference's bit for bit) is pasted clipped at the border when weight > 0.5; weight = 0 when the patch lies fully
 * outside.  Index work is bit-exact with the reference. */
typedef struct {
    int32_t B, J, H, W;            /* heat-map shape */
    int32_t patch, _pad;           /* patch edge = 6*sigma + 1 */
    double stride_x, stride_y;     /* image_size / heatmap_size */
    const double* joints;          /* [B,J,3] (x, y, -) in network-input pixels */
    const float* vis;              /* [B,J] joints_vis[:, 0] */
    const float* g;                /* [patch][patch] */
    float* target;                 /* [B,J,H,W] */
    float* weight;                 /* [B,J] */
} fpd_targets_t;
int fpd_render_targets(const fpd_targets_t* a, fpd_stream_t stream);

/* The crop of JointsDataset.__getitem__ (:160-168) for a batch: cv2.warpAffine(img, trans, (W,H), INTER_LINEAR)
 * (constant zero border) -> ToTensor (/255, HWC->CHW) -> Normalize(mean, std) (tools/fpd_train.py:182-193), one pass,
 * reading interleaved 8-bit source images that differ in size.  The bilinear sample follows OpenCV's fixed-point scheme
 * (source coordinates in 1/32 pixel from 10-bit fixed-point row/column terms, 15-bit weights, 8-bit rounded result)
 * so that the 8-bit intermediate matches cv2's; `minv` is the DST->SRC map (the inverse of get_affine_transform's
 * matrix, what cv2.warpAffine computes internally with invertAffineTransform). */
typedef struct {
    const uint8_t* img;            /* [h][w][3] interleaved */
    int32_t h, w;
    int64_t row_bytes;
    double minv[6];
} fpd_warp_src_t;
typedef struct {
    int32_t B, H, W, _pad;         /* output [B,3,H,W] fp32 */
    const fpd_warp_src_t* src;     /* device table [B] */
    float mean[3], std[3];         /* ToTensor + Normalize in torch's fp32 order: (u8 / 255.f - mean) / std, per SOURCE channel */
    float* out;
} fpd_warp_t;
int fpd_warp_affine(const fpd_warp_t* a, fpd_stream_t stream);

/* ---- execution plan: a recorded list of the ops above, replayed with one call ---- */
enum {
    FPD_OP_CONV = 0, FPD_OP_WGRAD = 1, FPD_OP_STEM_FWD = 2, FPD_OP_STEM_WGRAD = 3, FPD_OP_EW = 4,
    FPD_OP_LOSS = 5, FPD_OP_ADAM = 6, FPD_OP_MEMSET = 7, FPD_OP_WPREP = 8, FPD_OP_BNUPD = 9, FPD_OP_WREDUCE = 10,
    FPD_OP_BNECK = 11, FPD_OP_BNECK_FOLD = 12, FPD_OP_CONV_PAIR = 13, FPD_OP_BNECK_PAIR = 14, FPD_OP_EW_PAIR = 15, FPD_OP_PCK = 16, FPD_OP_HEAD = 17,
    FPD_OP_HEAD_FOLD = 18, FPD_OP_NOP = 19, FPD_OP_AFFSUM = 20, FPD_OP_NCHW2NHWC = 21, FPD_OP_CONV_F8 = 22,
    FPD_OP_WQUANT = 23          /* args: fpd_table_t over fpd_wquant_entry_t */
};
typedef struct { void* ptr; int64_t bytes; } fpd_memset_t;                 /* zero-fill */
typedef struct { const void* table; int32_t n; int32_t dtype; int64_t max_elems; } fpd_table_t;

typedef struct fpd_plan fpd_plan;
fpd_plan* fpd_plan_create(void);
void fpd_plan_destroy(fpd_plan* p);
/* appends one op (args are copied); returns its index or <0 */
int fpd_plan_add(fpd_plan* p, int32_t op, const void* args, int64_t args_bytes);
int fpd_plan_size(const fpd_plan* p);
/* multi-lane schedule: op `op` is issued on lane `lane` (0 = the stream passed to run/capture, >0 = a stream owned
 * by the plan) after the ops listed in wait_ops[0..n_waits) (indices of ops on OTHER lanes) have completed; ops of
 * one lane execute in plan order.  Every fpd_plan_run/capture range forks the side lanes from `stream` at its start
 * and joins them back into `stream` at its end, so waits never need to cross a range boundary (such entries are
 * ignored).  Default for every op: lane 0, no waits = the plain in-order list. */
#define FPD_MAX_LANES 16
int fpd_plan_set_schedule(fpd_plan* p, int32_t op, int32_t lane, const int32_t* wait_ops, int32_t n_waits);
/* Hand-off to streams OUTSIDE the plan (the RCCL gradient all-reduce of one bucket, issued while the rest of the
 * backward still runs): fpd_plan_mark_event() asks fpd_plan_run() to record an event after op `op` each time it issues
 * it; fpd_plan_wait_op() makes `stream` wait for the most recent such record (hipStreamWaitEvent; no host sync).
 * FPD_OP_NOP (args: fpd_memset_t, ignored) launches nothing -- a pure ordering point in a lane. */
int fpd_plan_mark_event(fpd_plan* p, int32_t op);
int fpd_plan_wait_op(fpd_plan* p, int32_t op, fpd_stream_t stream);
/* launch ops [begin,end) on stream (and the plan's side-lane streams, see above) */
int fpd_plan_run(fpd_plan* p, int32_t begin, int32_t end, fpd_stream_t stream);
/* Issue the single op `op` on `stream` itself, ignoring its lane and waits (measurement aid: bench.py re-launches one
 * recorded op back to back between two events to time a kernel exactly as the step launches it). */
int fpd_plan_run_op(fpd_plan* p, int32_t op, fpd_stream_t stream);
/* capture ops [begin,end) into a hipGraph once, then replay it (graph id returned by capture) */
int fpd_plan_capture(fpd_plan* p, int32_t begin, int32_t end, fpd_stream_t stream);
int fpd_plan_replay(fpd_plan* p, int32_t graph_id, fpd_stream_t stream);

/* ---- misc ---- */
const char* fpd_last_error(void);
int fpd_set_backend(int32_t backend);         /* FPD_BACKEND_*; returns previous */
/* run-time knobs (process-wide; the defaults are what the product uses): "conv_pp" = 0 never / 1 launches with >= 256 pixel
 * tiles (default) / 2 whenever in its domain: use of the persistent convolution kernel for big maps (csrc/conv_pp.hip);
 * "conv_pp_blocks" = its persistent blocks per occupancy slot (default 256); "wgrad_tile_only" = 1: fpd_conv_wgrad() fails
 * instead of falling through to the generic kernels when the halo-tile kernel declines a shape (tests).  Returns the
 * previous value (>= 0; 0 for "wgrad_tile_only"), negative = unknown option. */
int fpd_set_option(const char* name, int32_t value);
int fpd_abi_sizeof(const char* struct_name);  /* sizeof of a struct above, -1 if unknown */
/* Version of everything sizeof cannot see (buffer encodings and sizes).  History: 1 = statistics as fp64 [R][2][C] (rounds 1-3)
 * and, mistakenly unchanged, the first integer-limb encoding of round 4; 2 = fpd_stat_t limbs [R][2][2][C] with hi in units of
 * 2^-20 (twice the bytes of version 1's buffers: a host built against version 1 must not run).  Hosts assert
 * fpd_abi_version() == FPD_ABI_VERSION and size statistics buffers with fpd_stats_words(). */
#define FPD_ABI_VERSION 2
int fpd_abi_version(void);
int64_t fpd_stats_words(int32_t channels);    /* 64-bit words of one statistics buffer over `channels` = FPD_STATS_WORDS(channels) */
/* in-stream timing helpers (HIP events on `stream`): returns elapsed ms of [start,stop) */
void* fpd_event_create(void);
int fpd_event_record(void* ev, fpd_stream_t stream);
float fpd_event_elapsed_ms(void* start, void* stop); /* synchronises on stop */
void fpd_event_destroy(void* ev);

#ifdef __cplusplus
}
#endif
#endif /* FPD_AMD_H */
