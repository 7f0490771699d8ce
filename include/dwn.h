/* libdwiseneuro_hip.so — C-ABI of the MI355X-native DwiseNeuro hot path.
 *
 * The reference (lRomul/sensorium) has no native code and no FFI: its hot path is
 * `DwiseNeuro.forward` (src/models/dwiseneuro.py:397-405) + `MicePoissonLoss` (src/losses.py:10-21) driven
 * by `MouseModel.train_step` (src/argus_models.py:43-71), all delegated to torch/cuDNN ops.  This header is
 * the boundary a maintainer binds instead (ctypes stub: INTEGRATION.md): one entry point per fused kernel
 * family plus block-level composites, each citing the reference lines it replaces.
 *
 * Conventions
 *  - plain `extern "C"`, raw device pointers + sizes; no torch types.  The caller (PyTorch) owns and
 *    allocates every buffer, including workspaces; the library never allocates device memory, keeps no
 *    mutable global state, never synchronises, and enqueues only on the passed `stream`
 *    (hipGraph-capturable).  Every call takes an explicit `device` because autograd calls backward from a
 *    worker thread whose current device is not the caller's.
 *  - return value: 0 ok; < 0 argument / unsupported-configuration error; > 0 a hipError_t.
 *    `dwn_last_error()` returns a thread-local message.
 *  - activations are channels-last matrices [rows = (b,t,h,w)][C] in `dtype` storage
 *    (DWN_F32 = fp32 parity mode, DWN_BF16 = bf16 storage / fp32 accumulate).  Parameters, BN
 *    coefficients, statistics and gradients of parameters are always fp32.
 *  - channel counts must be multiples of 8 (one 16-byte bf16 vector).
 */
#ifndef DWN_H_
#define DWN_H_
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DWN_ABI_VERSION 7
#define DWN_F32 0
#define DWN_BF16 1
/* How the dtype-f32 GEMMs of a block / cortex layer / readout multiply.  NATIVE: v_mfma_f32_16x16x4_f32.  SPLIT3: each operand
 * as bf16 hi + lo, three bf16 MFMAs (hi*hi + hi*lo + lo*hi, fp32 accumulate): 5.8e-7 relative L2 from NATIVE on the full-width
 * eval forward, a sixth of the matrix-core time.  AUTO: SPLIT3 in the eval-mode forward, NATIVE in training (where the
 * analytically-zero gradients' summation noise would otherwise exceed the parity tests' floors). */
#define DWN_F32_AUTO 0
#define DWN_F32_NATIVE 1
#define DWN_F32_SPLIT3 2
#define DWN_NREP 32 /* replicas of every cross-workgroup statistics buffer: double[DWN_NREP][2][C] */

/* operand loader kinds (how a kernel reads one 16-byte channel vector of an operand) */
#define DWN_LD_PLAIN 0   /* p                                                     */
#define DWN_LD_PE 1      /* p + pe_t[t] + pe_h[h] + pe_w[w]      (dwiseneuro.py:184-192) */
#define DWN_LD_BNACT 2   /* act(v1*p + v2) * gate[b]             (dwiseneuro.py:16-22, 38-43) */
#define DWN_LD_AFFINE2 3 /* v1*p + v2*q + v3                     (BatchNorm backward) */
#define DWN_LD_DY3 4     /* v1*((p*gate[b]+gate2[b])*silu'(v4*q+v5)) + v2*q + v3 (SE + BN3 backward) */
#define DWN_LD_GATE 5    /* p * gate[b]                          (SE gate on a materialised activation) */
#define DWN_LD_CAT1 6    /* columns [0, cat_c1): p (row stride ld); [cat_c1, cat_c1 + cat_c2): q (row stride ld2); beyond: 1.0
                          * — dwn_gemm_tn's P operand only: [dh1 | a0 | 1]^T a0 gives the three raw products of the conv_pw
                          * weight gradient in one pass (dwn_block_backward; csrc/dwn_elementwise.hip k_pw_wgrad_fold) */

typedef struct dwn_load_desc {
    const void* p;
    const void* q;
    long long ld;
    const float* v1;
    const float* v2;
    const float* v3;
    const float* v4;
    const float* v5;
    const float* gate;
    const float* gate2;
    int gate_ld;
    int rows_per_sample;
    int act;
    const float* pe_t;
    const float* pe_h;
    const float* pe_w;
    int pT, pH, pW;
    int pe_ld;
    long long ld2;          /* DWN_LD_CAT1: row stride of q */
    int cat_c1, cat_c2;     /* DWN_LD_CAT1: columns taken from p / from q */
} dwn_load_desc;

#define DWN_EPI_STORE 0
#define DWN_EPI_READOUT 1
#define DWN_EPI_DG 2
#define DWN_EPI_STORE_CAT 3   /* DWN_EPI_STORE + K-concatenated second A operand (a2, K1) + fp32 bias[N] */
#define DWN_EPI_DH3 4         /* see gate3 / dps3 / coef3 below */

/* C[M][N] = load(A)[M][K] . B[N][K]^T  (+ BN statistics / readout / SE-grad epilogues).
 * Replaces nn.Conv3d 1x1x1 (dwiseneuro.py:91,118) and grouped nn.Conv1d k=1 (:207,276). */
typedef struct dwn_gemm_nn_args {
    dwn_load_desc a;
    int a_kind;
    const void* b; long long ldb;
    void* c; long long ldc;
    int M, N, K;
    int groups;
    double* stats;
    int f32_split;      /* dtype f32 only: nonzero = products as bf16 hi/lo splits on the bf16 matrix cores (hi*hi + hi*lo + lo*hi, fp32
                         * accumulate; ~1e-5 relative) instead of the fp32 MFMA — the eval-mode forward sets it, training does not */
    int stat_nchan;
    int epi;
    const float* bias; float sp_beta; float* out_nct; int Tn; int n_valid;
    /* DWN_EPI_DG: y3 = the activated project-conv input z3 = SiLU(BN3(y3)) [M][N] (materialised by the forward);
     * dg[b][n] += sum_{m in sample b} C[m][n] * z3[m][n].  s3 / t3 are reserved and must be NULL. */
    const void* y3; long long ldy3; const float* s3; const float* t3; float* dg; int dg_ld; int rows_per_sample;
    /* optional K-concatenation: columns k >= K1 of the A operand come from a2[m][k - K1] (plain loads);
     * only with epi == DWN_EPI_STORE_CAT, which also adds `bias` (fp32 [N]) before rounding. */
    const void* a2; long long a2_ld; int K1;
    /* optional per-sample weights: rows [b*b_rows_per_sample, (b+1)*b_rows_per_sample) multiply the weight matrix at
     * b + b*b_sample_stride elements (e.g. W . diag(gate_b): the SE gate folded into conv_pwl).  Needs
     * b_rows_per_sample % 128 == 0, groups == 1 and K > one k-tile; 0 = one weight matrix for every row. */
    long long b_sample_stride; int b_rows_per_sample;
    /* DWN_EPI_DH3 (conv_pwl data gradient fused with the SE-gate / SiLU / BatchNorm-3 backward prologue,
     * dwiseneuro.py:113-120 backward): with du = the GEMM result rounded to `dtype`, the kernel stores
     *   dh3[m][n] = (du*gate3[b][n] + dps3[b][n]) * silu'(coef3[0][n]*y3[m][n] + coef3[1][n])
     * into c (y3 = the raw temporal-conv output; b = m / rows_per_sample; gate3/dps3 rows are dg_ld apart) and, when
     * `stats` is set, accumulates sum(dh3) and sum(dh3 * (y3 - coef3[2][n]) * coef3[3][n]) — the two BatchNorm-backward
     * sums.  coef3 is the [4][coef3_ld] table scale, shift, mean, invstd. */
    const float* gate3; const float* dps3; const float* coef3; int coef3_ld;
    /* kernel variant: DWN_NN_AUTO = chosen by shape; DWN_NN_XL128 / DWN_NN_XL256 = the 256-row LDS-DMA kernel with 128- / 256-column
     * tiles wherever its argument constraints hold (bf16, plain loader, store epilogue); DWN_NN_KD = the A-direct kernel (deep K,
     * N % 128 == 0, plain / gate loader, store / K-concat epilogue); DWN_NN_TILE128 = never those kernels (the tests compare them
     * with the 128-row kernel bit for bit) */
    int variant;
} dwn_gemm_nn_args;
#define DWN_NN_AUTO 0
#define DWN_NN_XL128 1
#define DWN_NN_XL256 2
#define DWN_NN_TILE128 3   /* never the 256-row / A-direct kernels: the 128-row persistent kernel (what the tests compare them with) */
#define DWN_NN_KD 4        /* the A-direct kernel (deep K, N % 128 == 0: only the weights pass through LDS) wherever its constraints hold */

/* dW[R][Cc] += sum_m load(P)[m][r] * load(Q)[m][c]   (fp32 atomics; dW must be zeroed by the caller) */
typedef struct dwn_gemm_tn_args {
    dwn_load_desc p; int p_kind;
    dwn_load_desc q; int q_kind;
    int M, R, Cc;
    float* dw; long long lddw;
    int groups;
    int rows_per_split; int nsplit;       /* nsplit <= 0: chosen by the library */
    int R_load;                           /* P columns per group incl. zero padding (>= R; 0 -> R) */
    /* per-sample products: with rows_per_sample > 0 (M % rows_per_sample == 0, groups == 1) the rows of sample b
     * accumulate into dw + b*dw_sample_stride — B separate [R][Cc] matrices P_b = load(P)_b^T load(Q)_b.
     * splits_per_sample is chosen by the library. */
    int rows_per_sample; int splits_per_sample; long long dw_sample_stride;
    /* 1: the caller wants dW = the product, not dW += : when the launch needs only one M-split the tiles are written with plain
     * stores (the 64-byte-segment fp32 atomics are the slow part of a weight gradient with a big output: 16 M floats for a
     * readout) and dW need not be zeroed; when it needs more, the library zeroes dW itself first.  Needs lddw == Cc and
     * rows_per_sample == 0 (the [groups * R][Cc] matrix is one contiguous block). */
    int overwrite;
    /* 1 (ABI 7): dw points to DOUBLE [groups * R][lddw] and the M-split partial tiles are added with fp64 atomics — for products
     * that are later differenced (the Gram pass of dwn_conv_pw_bn_stats: var = w^T (G / n - mu mu^T) w cancels mean^2 against
     * E[y^2], and several hundred fp32 atomic adds per element carry 1e-6 of the SUM).  Not with overwrite / per-sample mode. */
    int dw_f64;
} dwn_gemm_tn_args;

/* depth-wise (1,k,k) conv, stride (1,s,s), pad k/2 — dwiseneuro.py:96-100 */
typedef struct dwn_dw_spatial_fwd_args {
    dwn_load_desc in;   /* DWN_LD_BNACT over raw y1 */
    const float* w;     /* [k*k][C] fp32 tap-major */
    void* out;
    int planes, Hin, Win, Hout, Wout, C, stride, ks;
    double* stats;
    int rows_band;      /* output rows per chunk / band; <= 0: chosen by the library */
    int impl;           /* 0: the library's choice (chained row-walk kernels where they apply); 1: the pair / generic kernels, the
                         * second implementation the tests compare bit for bit */
    /* rebuilt-input mode (a0 != NULL; dwn_dw_spatial_fwd_rc_supported): in.p is NOT read — the kernel rebuilds the rows of
     * y1 = a0 . w1^T it needs on the matrix cores (dwiseneuro.py:90-91) from the block input a0 [rows][a0_ld] (Cin channels) and
     * w1 = conv_pw's weight [C][Cin] rounded to bf16, row-major, and applies in.v1 / in.v2 (BatchNorm-1 scale / shift) + SiLU to
     * the fp32 accumulators (since round 6 the product is not rounded to bf16 first: the result is the reference arithmetic with
     * one rounding fewer than conv_pw's stored output + the stored-input form).  conv_pw + spat_covn_dw in one pass over a 7x
     * narrower input. */
    const void* a0; long long a0_ld; const void* w1; int Cin;
} dwn_dw_spatial_fwd_args;

typedef struct dwn_dw_spatial_bwd_args {
    dwn_load_desc dy;   /* DWN_LD_AFFINE2 over (dh2, y2) */
    dwn_load_desc y1;   /* p = raw y1; v1..v4 = bn1 scale, shift, mean, invstd */
    const float* w;
    void* dh1;
    float* dw;          /* [C][k*k] fp32, accumulated */
    int planes, Hin, Win, Hout, Wout, C, stride, ks;
    double* stats;
    int rows_band;
    int impl;           /* as in dwn_dw_spatial_fwd_args */
    /* rebuilt-y1 mode (a0 != NULL; dwn_dw_spatial_bwd_rc_supported): y1.p is NOT read — the kernel rebuilds the y1 values it needs
     * as a0 . w1^T on the matrix cores (dwiseneuro.py:90-91; the fp32 accumulators, not rounded to bf16: the same values the
     * rebuilt-input forward activates) from the block input a0 [rows][a0_ld] (Cin channels, the positional encoding included)
     * and w1 = conv_pw's weight [C][Cin] rounded to bf16, row-major.  y1.v1..v4 (BatchNorm-1 coefficients) as usual. */
    const void* a0; long long a0_ld; const void* w1; int Cin;
} dwn_dw_spatial_bwd_args;

/* depth-wise (k,1,1) conv along T, pad k/2 — dwiseneuro.py:105-109 */
typedef struct dwn_dw_temporal_fwd_args {
    dwn_load_desc in;
    const float* w;     /* [k][C] */
    void* out;
    int B, T, HW, C, kt;
    double* stats;
    /* eval-mode epilogue (BatchNorm-3 coefficients are known before the pass): with z_scale != NULL the kernel stores
     * z3 = SiLU(z_scale * y3 + z_shift) instead of y3 (y3 rounded to the storage type first, as a stored y3 would read back)
     * and, with pooled != NULL, adds the per-(sample, channel) sums of the stored z3 into pooled [B][C] (zeroed by the
     * caller) — the SqueezeExcite pooling pass (dwiseneuro.py:38-39) without a separate read of y3.  The sums are 64-bit
     * fixed point in units of 2^-32 (sum = pooled * 2^-32): integer adds make them independent of the arrival order. */
    const float* z_scale; const float* z_shift; long long* pooled;
} dwn_dw_temporal_fwd_args;

typedef struct dwn_dw_temporal_bwd_args {
    dwn_load_desc dy; int dy_kind;   /* DWN_LD_AFFINE2: dy3 = v1*p + v2*q + v3 from (dh3, y3); DWN_LD_PLAIN: p = dh3 only and
                                      * y3 is recomputed from y2 inside the kernel (one E-wide pass less); DWN_LD_DY3 */
    dwn_load_desc y2;                /* p = raw y2; v1..v4 = bn2 scale, shift, mean, invstd */
    const float* w;
    void* dh2;
    float* dw;                       /* [C][k] */
    int B, T, HW, C, kt;
    double* stats;
} dwn_dw_temporal_bwd_args;

/* BatchNorm parameter bundle (BatchNormAct, dwiseneuro.py:9-22) */
typedef struct dwn_bn {
    const float* gamma; const float* beta;
    float* running_mean; float* running_var; long long* num_batches_tracked;
    float* coef;    /* [4][C]: scale, shift, mean, invstd — written by forward, read by backward */
    float* dgamma; float* dbeta;   /* backward outputs (may be null in forward) */
} dwn_bn;

/* stem: Conv3d(C_in->C0, 1x1x1) + BN on the NCDHW fp32 input — dwiseneuro.py:306-309 */
typedef struct dwn_stem_args {
    int dtype, training, B, Cin, C0; long long S;   /* S = T*H*W */
    float eps, momentum;
    const float* x;      /* [B][Cin][S] fp32 */
    const float* w;      /* [C0][Cin] */
    dwn_bn bn;
    const float *pe_t, *pe_h, *pe_w;   /* optional: positional encoding of the FIRST block, added to `out`  */
    int T, H, W;                       /* (dwiseneuro.py:184-192); S = T*H*W                                 */
    void* y0;            /* unused since ABI 2 (the raw conv output is never materialised); may be NULL */
    void* out;           /* BN output [B*S][C0] */
    const void* dout;    /* backward: grad wrt out */
    float* dw;           /* backward: [C0][Cin], written (not accumulated) */
    void* ws; size_t ws_bytes;
    double* xmom;        /* [8 + 8*8] input moments: sum x_k, sum x_k x_l — written by the training forward, read by the
                          * backward (y0 = W0 x is linear in the <= 8 input channels: BatchNorm statistics and the weight
                          * gradient follow from them; saved instead of y0) */
} dwn_stem_args;

/* one InvertedResidual3d preceded by its PositionalEncoding3d — dwiseneuro.py:136-144, 184-192 */
typedef struct dwn_block_args {
    int dtype, training;
    int B, T, Hin, Win, Hout, Wout, Cin, Cmid, Cout, stride, ks, kt, se_r;
    float eps, momentum;
    const void* x; void* out;
    void *y1, *y2, *y3, *y4;
    void* z3;                                /* silu(bn3(y3)), materialised once by the SE pooling pass (saved) */
    int x_has_pe;                            /* 1: x already contains this block's positional encoding          */
    void* a0;                                /* x_has_pe == 0: [M_in][Cin] buffer receiving x + PE (saved)      */
    const float *pe_t, *pe_h, *pe_w;         /* this block's PE tables [T][Cin], [Hin][Cin], [Win][Cin]         */
    const float *out_pe_t, *out_pe_h, *out_pe_w;  /* next block's PE tables ([T|Hout|Wout][Cout]) added to out, or null */
    const float *w_pw, *w_dws, *w_dwt, *w_pwl, *se_wr, *se_br, *se_we, *se_be;
    dwn_bn bn1, bn2, bn3, bn4, bnsc;
    const float* drop_scale;                 /* [B] DropPath factor mask/keep (dwiseneuro.py:46-54) or null */
    const int *hsrc, *wsrc, *hinv, *winv;    /* nearest-interpolation index maps (device int32) */
    float *se_pmean, *se_hidpre, *se_gate;   /* saved SE activations [B][Cmid], [B][se_r], [B][Cmid] */
    /* backward only */
    const void* dout; void* dx;
    void *buf_a, *buf_b;                     /* scratch [max(M_in,M_out)][Cmid], [M_out][Cmid] */
    void *dy4, *da0;                         /* scratch [M_out][Cout], [M_in][Cin] */
    float *dw_pw, *dw_dws, *dw_dwt, *dw_pwl, *dse_wr, *dse_br, *dse_we, *dse_be;   /* overwritten (16-byte aligned) */
    void* ws; size_t ws_bytes;
    /* backward, conv_pwl: 0 = the library chooses by shape between (1) per-sample products dy4_b^T z3_b + a GEMM that
     * recomputes du in its epilogue and (2) the materialised du = dy4 . W2 with a separate reduction pass; 1 / 2 force one
     * (the parity tests run both implementations against the oracle) */
    int pwl_bwd;
    int f32_products;                        /* DWN_F32_AUTO / _NATIVE / _SPLIT3: how dtype f32 multiplies (see below) */
    /* training, bf16: 0 = the library leaves y1 (conv_pw's output, the widest tensor of the block) unmaterialised where both
     * stencils can rebuild it from the block input on the matrix cores (BatchNorm-1 statistics from the Gram matrix of the
     * input) AND that is the faster path (64 input channels): y1 may then be NULL in forward and backward
     * (dwn_block_forward_writes bit 0 clear); 1 = always materialise y1; 2 = y1-free wherever it is built (64 / 128 input channels).
     * Forward and backward of one block must be called with the same value. */
    int y1_mode;
} dwn_block_args;

/* AdaptiveAvgPool3d((None,1,1)) — dwiseneuro.py:374,400 */
typedef struct dwn_pool_args {
    int dtype; long long BT; int HW, C;
    const void* x; void* out; const void* dout; void* dx;
} dwn_pool_args;

/* ShuffleLayer — dwiseneuro.py:228-234 */
typedef struct dwn_cortex_args {
    int dtype, training, B, T, Cin, C, groups;
    float eps, momentum;
    const void* x; void* out; void* y;      /* y: raw conv output [B*T][C] (saved) */
    const float* w;                         /* [C][Cin/groups] */
    dwn_bn bn, bnsc;
    const float* drop_scale;
    const void* dout; void* dx; float* dw;  /* backward; dw [C][Cin/groups] fp32 (16-byte aligned) is cleared and written by the call */
    const float* dout_mask; int dout_mask_ld; /* optional [B][C] multiplier on dout (readout Dropout1d backward) */
    void* ws; size_t ws_bytes;
    int f32_products;                         /* DWN_F32_AUTO / _NATIVE / _SPLIT3 */
} dwn_cortex_args;

/* Readout — dwiseneuro.py:283-287 */
typedef struct dwn_readout_args {
    int dtype, B, T, Cin, groups, n_out;    /* n_out = N (valid neurons); weight has ceil(N/g)*g rows */
    float softplus_beta;
    const void* x;                          /* [B*T][Cin] */
    const float* w; const float* bias;      /* [Npad][Cin/groups], [Npad] */
    const float* drop_mask;                 /* [B][Cin] Dropout1d factor or null */
    float* out;                             /* [B][N][T] fp32 */
    const float* dout; void* dx; float* dw; float* dbias;   /* backward: dw is overwritten (no zeroing needed), dbias is
                                                              * accumulated with atomics (zeroed by the caller) */
    void* ws; size_t ws_bytes;
    /* optional: the weight in the data gradient's operand layout, [groups][Cin/groups][Rp] in the compute type
     * (dwn_readout_wt_bytes).  Non-null in forward: written by the same pass that packs the forward layout (one read of the
     * fp32 weight for both).  Non-null in backward: used as is — the caller kept it from the forward of the same step —
     * instead of packing the weight again. */
    void* wt;
    int f32_products;                       /* DWN_F32_AUTO (= native here: the readout has no training flag) / _NATIVE / _SPLIT3 */
} dwn_readout_args;

typedef struct dwn_tensor_entry {
    float* param; const float* grad; float* exp_avg; float* exp_avg_sq; float* ema;
    long long numel; int is_int64; int pad_;
} dwn_tensor_entry;

/* ---- batch assembly on the device --------------------------------------------------------
 * A trial stays resident in HBM in the reference's on-disk layout (src/datasets.py:37-51, src/data.py:59-70):
 * video [H0][W0][L] (uint8 or float32), behavior [2][L], pupil_center [2][L], responses [N][L] (float32).
 * `dwn_clip_src` names one window of one trial: frame of window position t = frame_start + t*frame_step
 * (IndexesGenerator.make_indexes, src/indexes.py:23-30).  The caller guarantees the frames lie inside [0, length). */
#define DWN_VID_U8 0
#define DWN_VID_F32 1
typedef struct dwn_clip_src {
    const void* video; const float* behavior; const float* pupil_center; const float* responses;
    long long length;
    int video_dtype;
    int frame_start, frame_step;
    int valid;                       /* 0 = slot unused */
} dwn_clip_src;
/* One sample of the batch: `src`, optionally mixed with `mix` (same mouse).
 * mix_mode DWN_MIX_BOX (CutMix, mixers.py:52-67): rows [bbx1,bbx2) x columns [bby1,bby2) of all five input channels come
 * from `mix` — the reference pastes its "x" range onto the row axis, mixers.py:63 — and the target is
 * one_minus_lam*relu(src) + lam*relu(mix) with lam = box area / (H*W) (mixers.py:64-65).
 * mix_mode DWN_MIX_BLEND (Mixup, mixers.py:22-33): inputs AND target are one_minus_lam*src + lam*mix with the drawn lam.
 * Both factors are rounded to float by the caller (torch multiplies a float tensor by the scalar cast to float). */
#define DWN_MIX_BOX 0
#define DWN_MIX_BLEND 1
typedef struct dwn_clip_desc {
    dwn_clip_src src, mix;
    int bbx1, bby1, bbx2, bby2;
    float one_minus_lam, lam;
    int mouse;                       /* owner: index into the per-mouse target table */
    int mix_mode;
} dwn_clip_desc;

/* opt-in kernel-family timer: HIP events recorded on the launch stream around the block-level kernels.
 * Used by bench.py for the live roofline measurement; disabled (mask 0) by default. */
enum {
    DWN_FAM_PW_FWD = 0, DWN_FAM_DWS_FWD, DWN_FAM_DWT_FWD, DWN_FAM_SE_POOL, DWN_FAM_PWL_FWD, DWN_FAM_RESID_FWD,
    DWN_FAM_RESID_BWD, DWN_FAM_PWL_DGRAD, DWN_FAM_PWL_WGRAD, DWN_FAM_BN3_REDUCE, DWN_FAM_DWT_BWD, DWN_FAM_DWS_BWD,
    DWN_FAM_PW_DGRAD, DWN_FAM_PW_WGRAD, DWN_FAM_CORTEX_FWD, DWN_FAM_CORTEX_BWD, DWN_FAM_READOUT_FWD, DWN_FAM_READOUT_BWD,
    DWN_FAM_COUNT
};
/* family_mask bit DWN_PROF_ROCTX: additionally bracket every family launch with a roctx range named after the family
 * ("dwn:pw_fwd", ...) so that rocprofv3 --marker-trace timelines are labelled; the roctx library is dlopen'ed on first use
 * (librocprofiler-sdk-roctx.so, else libroctx64.so) and silently skipped when absent.  The range bit alone records no events. */
#define DWN_PROF_ROCTX (1ull << 63)
int dwn_profile_enable(unsigned long long family_mask, int device);
int dwn_profile_collect(int family, double* total_ms, long long* launches);

int dwn_abi_version(void);
const char* dwn_source_hash(void);         /* sha256[:16] of the source files the library was built from (csrc/Makefile HASH_SRCS) */
int dwn_sizeof(const char* struct_name);   /* sizeof of a struct of this header, -1 if unknown (binding self-check) */
const char* dwn_last_error(void);

/* kernel families */
int dwn_gemm_nn(const dwn_gemm_nn_args* a, int dtype, int device, void* stream);
int dwn_gemm_tn(const dwn_gemm_tn_args* a, int dtype, int device, void* stream);
int dwn_dw_spatial_fwd(const dwn_dw_spatial_fwd_args* a, int dtype, int device, void* stream);
int dwn_dw_spatial_bwd(const dwn_dw_spatial_bwd_args* a, int dtype, int device, void* stream);
/* 1 when dwn_dw_spatial_fwd can run these arguments in rebuilt-input mode (bf16, Cin 64 or 128, C % 64 == 0, the row-walk plane widths) */
int dwn_dw_spatial_fwd_rc_supported(const dwn_dw_spatial_fwd_args* a, int dtype);
/* 1 when dwn_dw_spatial_bwd can run these arguments in rebuilt-y1 mode (bf16, Cin 64 or 128, C % 64 == 0, the row-walk plane widths) */
int dwn_dw_spatial_bwd_rc_supported(const dwn_dw_spatial_bwd_args* a, int dtype);
int dwn_dw_temporal_fwd(const dwn_dw_temporal_fwd_args* a, int dtype, int device, void* stream);
int dwn_dw_temporal_bwd(const dwn_dw_temporal_bwd_args* a, int dtype, int device, void* stream);
int dwn_bn_finalize(const double* stats, int stat_c, double count, const dwn_bn* bn, int C, int training,
                    float momentum, float eps, int device, void* stream);
int dwn_bn_bwd_finalize(const double* stats, double count, const dwn_bn* bn, float* abc, int C, int device,
                        void* stream);
int dwn_pack_weight(const float* src, void* dst, int groups, int R, int C, int transpose, int Rd, int Cd,
                    int dtype, int device, void* stream);
/* BatchNorm-1 (conv_pw.1.bn, dwiseneuro.py:91-92) in training mode WITHOUT conv_pw's output: y1 = a0 . W1^T is linear in the block
 * input a0 [M][a0_ld] (Cin channels), so its batch mean / variance follow from the Gram matrix a0^T a0 and the column sums 1^T a0
 * (one Cin-wide pass): mean_e = w_e . mu, var_e = w_e^T (G / M - mu mu^T) w_e, with w_pw [E][Cin] rounded to `dtype` first (the
 * weights the matrix cores multiply with).  Writes bn->coef [4][E] = scale, shift, mean, invstd and updates the running statistics
 * and num_batches_tracked like nn.BatchNorm3d.  sc_stats (optional, double[DWN_NREP][2][Cin], zeroed by the caller): receives the
 * shortcut BatchNorm's sums of an identity-map block (sum a0, sum a0^2 per channel) in replica 0.  ws: caller-owned scratch. */
size_t dwn_conv_pw_bn_stats_workspace_bytes(int Cin);    /* (Cin + 8) * Cin doubles + alignment */
int dwn_conv_pw_bn_stats(const void* a0, long long a0_ld, long long M, const float* w_pw, int E, int Cin, const dwn_bn* bn,
                         float momentum, float eps, double* sc_stats, void* ws, size_t ws_bytes, int dtype, int device,
                         void* stream);

/* composites (forward / backward of one reference module each) */
size_t dwn_stem_workspace_bytes(const dwn_stem_args* a);
int dwn_stem_forward(const dwn_stem_args* a, int device, void* stream);
int dwn_stem_backward(const dwn_stem_args* a, int device, void* stream);
size_t dwn_block_workspace_bytes(const dwn_block_args* a, int backward);
int dwn_block_forward(const dwn_block_args* a, int device, void* stream);
int dwn_block_backward(const dwn_block_args* a, int device, void* stream);
/* bit 0: dwn_block_forward writes y1; bit 1: it writes y3 (eval mode skips them where the stencil rebuilds y1 / the temporal
 * pass emits z3 directly: the caller may leave those pointers NULL) */
int dwn_block_forward_writes(const dwn_block_args* a);
int dwn_pool_forward(const dwn_pool_args* a, int device, void* stream);
int dwn_pool_backward(const dwn_pool_args* a, int device, void* stream);
size_t dwn_cortex_workspace_bytes(const dwn_cortex_args* a, int backward);
int dwn_cortex_forward(const dwn_cortex_args* a, int device, void* stream);
int dwn_cortex_backward(const dwn_cortex_args* a, int device, void* stream);
size_t dwn_readout_workspace_bytes(const dwn_readout_args* a, int backward);
size_t dwn_readout_wt_bytes(const dwn_readout_args* a);
int dwn_readout_forward(const dwn_readout_args* a, int device, void* stream);
int dwn_readout_backward(const dwn_readout_args* a, int device, void* stream);

/* MicePoissonLoss — losses.py:10-21.  w = mice_weights[:, m] / sum(mice_weights) (normalised by caller).
 * forward accumulates into *loss_acc (double, zeroed by caller); backward writes dpred. */
int dwn_poisson_loss_forward(const float* pred, const float* target, const float* w, long long per_sample,
                             long long total, float eps, double* loss_acc, int device, void* stream);
int dwn_poisson_loss_backward(const float* pred, const float* target, const float* w, const float* gscale,
                              long long per_sample, long long total, float eps, float* dpred, int device,
                              void* stream);
int dwn_f64_to_f32(const double* src, float* dst, int n, int device, void* stream);

/* fused multi-tensor AdamW (+EMA) — torch.optim.AdamW (true_batch_001.py:45-48) + ModelEma.update (ema.py:47-55).
 * `list` is a DEVICE array of ntensors entries.  step >= 1. */
int dwn_adamw_ema_multi(const dwn_tensor_entry* list, int ntensors, int max_blocks, double lr, double beta1,
                        double beta2, double eps, double weight_decay, long long step, double ema_decay,
                        double grad_scale, int device, void* stream);
int dwn_ema_lerp_multi(const dwn_tensor_entry* list, int ntensors, int max_blocks, double decay, int device,
                       void* stream);


/* conv_pw backward (dwiseneuro.py:90-93 backward) WITHOUT reading y1.  With the BatchNorm-1 backward affine
 *   dy1 = abc[0]*dh1 + abc[1]*y1 + abc[2]   (per channel of E; abc is [3][E])   and   y1 = a0 . W1^T,
 * both products fold the y1 term into Cin x Cin matrices:
 *   da0[M][Cin] = dy1 . W1    = [dh1 | a0] . [diag(abc0) W1 ; W1^T diag(abc1) W1] + abc2 . W1
 *   dw[E][Cin]  = dy1^T . a0  = diag(abc0) (dh1^T a0) + diag(abc1) W1 (a0^T a0) + abc2 (1^T a0)     (fp32, overwritten)
 * so the E-wide traffic is one read of dh1 (dtype == DWN_BF16, Cin == 64, E == 448 or 384, M % 128 == 0 — see
 * dwn_pw_bwd_fused_supported — one kernel computes both products from a single pass) or two (dwn_gemm_nn with the
 * K-concatenated operand + dwn_gemm_tn with the DWN_LD_CAT1 loader).  w_pw = conv_pw.0.weight [E][Cin] fp32 (used as rounded
 * to `dtype`, the values the forward multiplied with).  ws: dwn_pw_backward_workspace_bytes(E, Cin, dtype) bytes, 256-aligned. */
typedef struct dwn_pw_bwd_args {
    const void* dh1; const void* a0; const float* w_pw; const float* abc;
    void* da0; float* dw;
    long long M; int E; int Cin;
    void* ws; size_t ws_bytes;
    /* optional: the gradient of a stride-1 block's shortcut branch (BatchNorm of the channel-tiled block input, dwiseneuro.py:125-134)
     * folded in, so that da0 becomes the block's input gradient:
     *   da0[m][c] += sum_{c' = c + j*Cin < res_C} (res_abc[0][c']*res[m][c'] + res_abc[1][c']*a0[m][c] + res_abc[2][c'])
     * res = the block's output gradient [M][res_C] (`dtype`), res_abc = [3][res_C]; res_C in {Cin, 2*Cin}.  Only where
     * dwn_pw_bwd_fused_supported (the one-pass kernel's epilogue adds it; on the two-GEMM path it measured slower than the
     * separate pass it replaces: -3 there).  NULL = off. */
    const void* res; const float* res_abc; int res_C;
    /* ... of a strided block: res_hinv [res_Hin] / res_winv [res_Win] = the inverse nearest maps (output row / column an input
     * row / column is sampled into, or -1), M = frames * res_Hin * res_Win; only rows m = (f, hi, wi) with both >= 0 get the sum
     * above, with res read at row (f * res_Hout + ho) * res_Wout + wo.  NULL = identity map (stride 1: res has M rows). */
    const int* res_hinv; const int* res_winv; int res_Hin, res_Win, res_Hout, res_Wout;
} dwn_pw_bwd_args;
int dwn_pw_bwd_fused_supported(int dtype, long long M, int E, int Cin);
size_t dwn_pw_backward_workspace_bytes(int E, int Cin, int dtype);
int dwn_pw_backward(const dwn_pw_bwd_args* a, int dtype, int device, void* stream);

/* ---- spat_covn_dw WITHOUT a materialised conv_pw output (dwiseneuro.py:90-102; bf16 storage, 3x3, stride 1 or 2,
 * Cin in {64, 128}, E % 64 == 0).  y1 = a0 . W1^T is the widest tensor at input resolution although it is a Cin-deep
 * product of a tensor E/Cin times narrower: the kernel rebuilds each y1 tile from the a0 tile with MFMAs while it
 * stages the stencil's LDS tile, so the forward reads a0 (once per tile, all E/64 channel slices are produced from
 * the LDS-resident copy) and writes y2.  `blob` holds, per 64-channel slice, the LDS images the kernel copies by
 * LDS-DMA: the W1 rows (bf16, bank-swizzled), the pair-packed bf16 stencil weights and the BatchNorm-1 scale / shift;
 * dwn_dw_spatial_rc_prep builds it (after the BatchNorm-1 coefficients are final).
 * round_y1 != 0 rounds the rebuilt y1 to bf16 before the BatchNorm (bit-identical to reading a stored bf16 y1). */
typedef struct dwn_dw_spatial_rc_fwd_args {
    const void* a0; long long a0_ld;   /* block input incl. its positional encoding [planes*Hin*Win][Cin] */
    const void* blob;                  /* dwn_dw_spatial_rc_blob_bytes(E, Cin) bytes */
    void* out;                         /* y2 [planes*Hout*Wout][E] */
    int planes, Hin, Win, Hout, Wout, Cin, E, stride;
    double* stats;                     /* double[DWN_NREP][2][E] sum / sum of squares of y2, or null */
    int rows_band;                     /* output rows per tile; <= 0: chosen by the library */
    int round_y1;
} dwn_dw_spatial_rc_fwd_args;
size_t dwn_dw_spatial_rc_blob_bytes(int E, int Cin);
/* w_pw [E][Cin] fp32 (conv_pw.0.weight), w_dws [9][E] fp32 tap-major, bn1_coef [>=2][E] scale, shift */
int dwn_dw_spatial_rc_prep(const float* w_pw, const float* w_dws, const float* bn1_coef, int E, int Cin, void* blob,
                           int device, void* stream);
int dwn_dw_spatial_rc_supported(int dtype, int Cin, int E, int ks, int stride, int Hin, int Win);
int dwn_dw_spatial_fwd_rc(const dwn_dw_spatial_rc_fwd_args* a, int device, void* stream);

/* StackInputsProcessor + CutMix on the inputs (inputs.py:15-36, mixers.py:52-63) for a whole batch:
 * x [B][5][T][H][W] fp32 = channel 0 the video centre-padded with pad_fill, channels 1-2 behavior, 3-4 pupil_center
 * broadcast over the frame.  `descs` is a DEVICE array of B entries; every video has the same H0 x W0. */
int dwn_assemble_inputs(const dwn_clip_desc* descs, int B, int T, int H0, int W0, int H, int W, float pad_fill,
                        float* x, int device, void* stream);
/* responses_to_tensor + CutMix target blend + construct_mice_sample + collate (responses.py:25-29, mixers.py:64-66,
 * datasets.py:172-187): targets[m] (DEVICE table of n_mice pointers) is [B][n_neurons[m]][T] fp32, fully overwritten —
 * the owner's rows with the target, every other mouse's rows of that sample with zeros; mice_weights [B][n_mice]
 * one-hot.  `max_neurons` = max over n_neurons (host copy, sizes the grid). */
int dwn_assemble_targets(const dwn_clip_desc* descs, int B, int T, float* const* targets, const int* n_neurons,
                         int n_mice, int max_neurons, float* mice_weights, int device, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* DWN_H_ */
