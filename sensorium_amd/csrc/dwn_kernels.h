// Prototypes of the glue-kernel launchers (dwn_elementwise.hip).  All return 0 or a negative/positive error code.
#pragma once
#include "dwn_internal.h"

struct ResGeom {
    int BT, T, Hin, Win, Hout, Wout, Cin, Cout;
    const int* hsrc; const int* wsrc;     // [Hout], [Wout]: nearest source index (dwiseneuro.py:127-129)
    const int* hinv; const int* winv;     // [Hin], [Win]: inverse map or -1
};

typedef dwn_tensor_entry TensorListEntry;

struct BnFinJob {
    const double* stats; int stat_c; double count; const float* gamma; const float* beta;
    float* running_mean; float* running_var; long long* nbt; float* coef; int C; int nblocks;
};
struct BnBwdJob { const double* stats; double count; const float* coef; float* dgamma; float* dbeta; float* abc; int C; int nblocks; };
int k_bn_finalize_train2(BnFinJob j0, BnFinJob j1, float momentum, float eps, hipStream_t s);
int k_bn_bwd_finalize2(BnBwdJob j0, BnBwdJob j1, hipStream_t s);
int k_bn_finalize_train(const double* stats, int stat_c, double count, const float* gamma, const float* beta,
                        float* rm, float* rv, long long* nbt, float momentum, float eps, float* coef, int C,
                        hipStream_t s);
int k_bn_finalize_eval(const float* gamma, const float* beta, const float* rm, const float* rv, float eps,
                       float* coef, int C, hipStream_t s);
int k_bn_bwd_finalize(const double* stats, double count, const float* coef, float* dgamma, float* dbeta, float* abc,
                      int C, hipStream_t s);
// BatchNorm-1 statistics from the Gram matrix of the block input (y1 never materialised; dwn_elementwise.hip)
int k_bn1_gram_finalize(const double* gram, const float* w1, int E, int Cin, double count, const float* gamma, const float* beta,
                        float* rm, float* rv, long long* nbt, float momentum, float eps, float* coef, double* sc_stats, int dtype,
                        hipStream_t s);
int k_ew_apply(const LoadDesc& d, int kind, void* out, i64 ldo, i64 rows, int C, int dtype, hipStream_t s);
int k_colstats(const LoadDesc& d, int kind, i64 rows, int C, double* stats, int dtype, hipStream_t s);
// stem through the input moments (no y0 round trip; dwn_elementwise.hip)
int k_stem_xmom(const float* x, int B, int Cin, i64 S, double* mom, hipStream_t s);
int k_stem_bn_finalize(const double* mom, double count, const float* w, const float* gamma, const float* beta, float* rm,
                       float* rv, long long* nbt, float momentum, float eps, float* coef, double* xmom, int C0, int Cin,
                       hipStream_t s);
int k_stem_out(const float* x, const float* w, const float* coef, const float* pe_t, const float* pe_h, const float* pe_w,
               int Tn, int H, int W, int B, int Cin, i64 S, int C0, void* out, int dtype, hipStream_t s);
int k_stem_bwd_acc(const void* dout, const float* x, const double* xmom, double count, int B, int Cin, i64 S, int C0,
                   double* acc, int dtype, hipStream_t s);
int k_stem_bwd_finalize(const double* acc, const double* xmom, const float* w, const float* coef, double count, float* dgamma,
                        float* dbeta, float* dw, int C0, int Cin, hipStream_t s);
int stem_moment_count();
int stem_acc_stride();
int k_shortcut_stats(const LoadDesc& xin, const ResGeom& gm, double* stats, int dtype, hipStream_t s);
int k_residual_fwd(const LoadDesc& xin, const void* y4, const float* coef4, const float* coefsc, const float* dscale,
                   const ResGeom& gm, const float* ope_t, const float* ope_h, const float* ope_w, void* out, int dtype,
                   hipStream_t s);
int k_residual_bwd_reduce(const LoadDesc& xin, const void* y4, const void* dout, const float* coef4,
                          const float* coefsc, const float* dscale, const ResGeom& gm, double* stats4,
                          double* statssc, int dtype, hipStream_t s);
int k_residual_bwd_dy4(const void* y4, const void* dout, const float* abc4, const float* dscale, const ResGeom& gm,
                       void* dy4, int dtype, hipStream_t s);
int k_residual_bwd_dx(const LoadDesc& xin, const void* da0, const void* dout, const float* abcsc, const ResGeom& gm,
                      void* dx, int dtype, hipStream_t s);
int k_se_pool(const LoadDesc& z3, int B, int C, int rows_per_sample, long long* pooled, void* z3out, int dtype, hipStream_t s);
int k_se_mlp_fwd(const long long* pooled_sum, float inv_s, const float* wr, const float* br, const float* we,
                 const float* be, int B, int C, int R, float* pmean, float* hid_pre, float* gate, const float* w2, void* wg, int N2,
                 int dtype, int* folded, hipStream_t s);
int k_se_mlp_bwd(const float* dg, const float* gate, const float* hid_pre, const float* pmean, const float* wr,
                 const float* we, int B, int C, int R, float inv_s, float* dgp, float* dhp, float* dps, float* dwr,
                 float* dbr, float* dwe, float* dbe, hipStream_t s);
int k_bn3_bwd_reduce(const LoadDesc& d, const float* coef3, i64 rows, int C, double* stats, void* dh_out, int dtype, hipStream_t s);
int k_pool_fwd(const void* x, void* out, i64 BT, int HW, int C, int dtype, hipStream_t s);
int k_pool_bwd(const void* dpool, void* dx, i64 BT, int HW, int C, int dtype, hipStream_t s);
int k_cortex_residual_fwd(const void* y, const void* x, const float* coef, const float* coefsc, const float* dscale,
                          int M, int Tn, int Cin, int C, int groups, void* out, int dtype, hipStream_t s);
int k_cortex_bwd_reduce(const void* y, const void* x, const void* dout, const float* gmask, int gmask_ld,
                        const float* coef, const float* coefsc, const float* dscale, int M, int Tn, int Cin, int C,
                        int groups, double* stats, double* statssc, int dtype, hipStream_t s);
int k_cortex_bwd_dy(const void* y, const void* dout, const float* gmask, int gmask_ld, const float* coef,
                    const float* abc, const float* dscale, int M, int Tn, int C, int groups, void* dy, int dtype,
                    hipStream_t s);
int k_cortex_bwd_dx(const void* dxmain, const void* x, const void* dout, const float* gmask, int gmask_ld,
                    const float* abcsc, int M, int Tn, int Cin, int C, void* dx, int dtype, hipStream_t s);
int k_pack_weight(const float* src, void* dst, int groups, int R, int C, int transpose, int Rd, int Cd, int dtype,
                  hipStream_t s);
int k_pack_dw(const float* src, float* dst, int C, int taps, hipStream_t s);
int k_gate_weights(const float* w, const float* gate, void* dst, int B, int N, int K, int dtype, hipStream_t s);
int k_readout_dz(const float* dout, const float* out, float beta, int B, int Tn, int n_valid, int Rg, int Rp,
                 int groups, void* dz, float* db, int dtype, hipStream_t s);
int k_poisson_fwd(const float* pred, const float* target, const float* w, i64 per_sample, i64 total, float eps,
                  double* loss, hipStream_t s);
int k_poisson_bwd(const float* pred, const float* target, const float* w, const float* gscale, i64 per_sample,
                  i64 total, float eps, float* dpred, hipStream_t s);
int k_f64_to_f32(const double* src, float* dst, int n, hipStream_t s);
int k_adamw_ema(const TensorListEntry* list, int ntensors, int max_blocks, float decay_w, float omb1, float beta2,
                float omb2, float eps, float step_size, float bc2_sqrt, float ema_decay, float ema_omd,
                float grad_scale, hipStream_t s);
int k_ema_lerp(const TensorListEntry* list, int ntensors, int max_blocks, float decay, float omd, hipStream_t s);
int k_fill_f32(float* p, float v, int n, hipStream_t s);
int k_pack_weight_dual(const float* src, void* plain, void* tr, int groups, int R, int C, int Rp, int ldp, int ldt, int dtype,
                       hipStream_t s);
int k_pwl_bwd_reduce(const float* P, const float* gate, const float* W, int B, int K, int N, float* dW, float* dg,
                     hipStream_t s);
int k_assemble_inputs(const dwn_clip_desc* descs, int B, int T, int H0, int W0, int H, int W, float pad, float* x,
                      hipStream_t s);
int k_assemble_targets(const dwn_clip_desc* descs, int B, int T, float* const* targets, const int* n_neurons,
                       int n_mice, int max_neurons, float* mice_weights, hipStream_t s);
int k_zero(void* p, size_t nbytes, hipStream_t s);
int k_pw_bwd_prep(const float* w1, const float* abc, int E, int C, void* bp, float* gacc, float* r3, int dtype,
                  const float* res_abc, int res_C, hipStream_t s);
// conv_pw weight gradient from the raw products (see pw_bwd_fused_kernel): tacc [(E + C + 8)][C] fp32 = rows T1 = dh1^T a0,
// Ga = a0^T a0, s = 1^T a0;  dw[e][c] = A1[e] T1[e][c] + A2[e] sum_c' W1[e][c'] Ga[c'][c] + A3[e] s[c]   (W1 as rounded to dtype)
size_t pw_wgrad_tacc_floats(int E, int C);
int k_pw_wgrad_fold(const float* tacc, const float* abc, const float* w1, int E, int C, float* dw, int dtype, hipStream_t s);

// ---- fused per-call preparation (k_prep): zero ranges, constant fills and weight packs in one launch
enum { PREP_ZERO = 0, PREP_FILL = 1, PREP_PACKW = 2, PREP_PACKDW = 3, PREP_BNEVAL = 4 };
#define PREP_MAX_OPS 16
struct PrepOp {
    int kind, nblocks;
    const float* src; void* dst;
    const float* src2; const float* src3; const float* src4;     // BNEVAL: beta, running_mean, running_var (src = gamma)
    long long n;                       // ZERO: 16-byte units; FILL / PACK* / BNEVAL: elements (channels)
    float v;
    int R, C, transpose, Rd, Cd;       // PACKW: src [groups][R][C] -> dst [groups][Rd][Cd]; PACKDW: C channels, R taps
};
struct PrepArgs {
    int nops;
    PrepOp op[PREP_MAX_OPS];
    PrepArgs() : nops(0) {}
    static int blocks_for(long long n, int cap) { long long b = (n + 255) / 256; return (int)(b > cap ? cap : b); }
    bool zero(void* p, size_t nbytes) {
        if (nbytes == 0) return true;
        if (((size_t)p & 15) || (nbytes & 15) || nops >= PREP_MAX_OPS) return false;
        PrepOp& q = op[nops++]; q = PrepOp(); q.kind = PREP_ZERO; q.dst = p; q.n = (long long)(nbytes / 16); q.nblocks = blocks_for(q.n, 256);
        return true;
    }
    bool fill(float* p, float v, int n) {
        if (n <= 0) return true;
        if (nops >= PREP_MAX_OPS) return false;
        PrepOp& q = op[nops++]; q = PrepOp(); q.kind = PREP_FILL; q.dst = p; q.v = v; q.n = n; q.nblocks = blocks_for(n, 1 << 30);
        return true;
    }
    bool packw(const float* src, void* dst, int groups, int R, int C, int transpose, int Rd, int Cd) {
        if (nops >= PREP_MAX_OPS) return false;
        PrepOp& q = op[nops++]; q = PrepOp(); q.kind = PREP_PACKW; q.src = src; q.dst = dst; q.R = R; q.C = C; q.transpose = transpose;
        q.Rd = Rd; q.Cd = Cd; q.n = (long long)groups * Rd * Cd; q.nblocks = blocks_for(q.n, 1 << 30);
        return true;
    }
    bool packdw(const float* src, float* dst, int C, int taps) {
        if (nops >= PREP_MAX_OPS) return false;
        PrepOp& q = op[nops++]; q = PrepOp(); q.kind = PREP_PACKDW; q.src = src; q.dst = dst; q.C = C; q.R = taps; q.n = (long long)C * taps;
        q.nblocks = blocks_for(q.n, 1 << 30);
        return true;
    }
    // eval-mode BatchNorm coefficients [4][C] = scale, shift, mean, invstd from the running statistics (v = eps)
    bool bneval(const dwn_bn& bn, int C, float eps) {
        if (nops >= PREP_MAX_OPS) return false;
        PrepOp& q = op[nops++]; q = PrepOp(); q.kind = PREP_BNEVAL; q.src = bn.gamma; q.src2 = bn.beta; q.src3 = bn.running_mean;
        q.src4 = bn.running_var; q.dst = bn.coef; q.v = eps; q.n = C; q.C = C; q.nblocks = blocks_for(C, 1 << 30);
        return true;
    }
};
int k_prep(const PrepArgs& pa, int dtype, hipStream_t s);
