// Memory-bound glue kernels: BatchNorm finalisation (forward / eval / backward coefficients), stem conv
// (K = 5 is too thin for MFMA), shortcut gather + residual add (+DropPath), H×W pooling, Squeeze-Excite
// pooling and MLP, cortex shuffle/residual, readout glue, Poisson loss, fused AdamW + EMA.
// Reference call sites are cited per kernel (paths relative to lRomul/sensorium).
#include "dwn_internal.h"
#include "dwn_kernels.h"

#define NCV 8
#define DISPATCH_T(dtype, CALL_BF, CALL_F) do { if ((dtype) == DWN_BF16) { CALL_BF; } else { CALL_F; } } while (0)

// every "sliced" kernel: thread -> (cv = tid % 8, pl = tid / 8), channel slice = blockIdx.y
#define SLICE_SETUP(Cval)                                   \
    constexpr int KC = TT<T>::KC;                           \
    const int tid = threadIdx.x;                            \
    const int cv = tid % NCV, pl = tid / NCV;               \
    const int c0 = blockIdx.y * NCV * KC;                   \
    const int chan = c0 + cv * KC;                          \
    const bool chan_ok = chan < (Cval);                     \
    (void)pl; (void)chan_ok;

template <int KC>
__device__ __forceinline__ void slice_stats_flush(float* lstat, const float* s0, const float* s1, int cv,
                                                 int c0, int C, double* stats, int rep) {
    // lanes l, l+8, l+16, ... of a wave hold the same channel vector (cv = tid % NCV, NCV = 8): combine them with
    // xor-shuffles so that one lane per vector and wave touches LDS
    static_assert(NCV == 8, "wave pre-reduction below assumes 8 channel vectors per slice");
    const int tid = threadIdx.x;
    DET_WAVES_BEGIN
#pragma unroll
    for (int i = 0; i < KC; ++i) {
        float a = s0[i], b = s1[i];
#pragma unroll
        for (int o = 8; o < 64; o <<= 1) { a += __shfl_xor(a, o); b += __shfl_xor(b, o); }
        if ((tid & 63) < NCV) {
            atomicAdd(&lstat[cv * KC + i], a);
            atomicAdd(&lstat[NCV * KC + cv * KC + i], b);
        }
    }
    DET_WAVES_END
    __syncthreads();
    DET_ENTER();
    if (tid < 2 * NCV * KC) {
        int which = tid / (NCV * KC), c = c0 + tid % (NCV * KC);
        if (c < C) stat_add(stats, rep, C, which, c, lstat[tid]);
    }
}

// persistent sizing for the streaming kernels that keep per-workgroup accumulators: exactly one resident wave of
// workgroups (occupancy query), so start-up / flush costs are paid once per CU slot and there is no ragged tail
template <typename K>
static dim3 resident_slice_grid(K kernel, i64 rows, int C, int KC, int max_bpc = 1 << 20) {
    int slices = (C + NCV * KC - 1) / (NCV * KC);
    int bpc = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&bpc, kernel, 256, 0) != hipSuccess || bpc < 1) { (void)hipGetLastError(); bpc = 4; }
    if (bpc > max_bpc) bpc = max_bpc;
    i64 bx = (256 * (i64)bpc) / slices;
    i64 need = (rows + 31) / 32;
    if (bx > need) bx = need;
    if (bx < 1) bx = 1;
    return dim3((unsigned)bx, (unsigned)slices);
}

static inline dim3 slice_grid(i64 rows, int C, int KC, int cap = 2048) {
    int slices = (C + NCV * KC - 1) / (NCV * KC);
    i64 bx = (rows + 31) / 32;
    int capx = (cap + slices - 1) / slices;
    if (bx > capx) bx = capx;
    if (bx < 1) bx = 1;
    return dim3((unsigned)bx, (unsigned)slices);
}

// one 1024-thread workgroup per CU in total (across the channel slices): the fewest end-of-kernel flushes per
// statistics address that still fills the chip
static inline dim3 fat_slice_grid(i64 rows, int C, int KC) {
    int slices = (C + NCV * KC - 1) / (NCV * KC);
    i64 bx = 256 / slices;
    i64 need = (rows + (1024 / NCV) - 1) / (1024 / NCV);
    if (bx > need) bx = need;
    if (bx < 1) bx = 1;
    return dim3((unsigned)bx, (unsigned)slices);
}

// ------------------------------------------------------------------------------------------------
// BatchNorm finalisation (BatchNormAct, dwiseneuro.py:9-22; torch BatchNorm semantics)
// coef layout: [4][C] = scale, shift, mean, invstd
// ------------------------------------------------------------------------------------------------
// One lane per (channel, replica): 32 consecutive lanes hold the 32 replicas of a channel's two fp64 sums and combine
// them with xor-shuffles — a single-thread loop over the replicas was 64 dependent-latency loads (7-8 us per launch,
// ~100 launches per training step).
static_assert(DWN_NREP == 32, "replica reduce below assumes 32 replicas");
__device__ __forceinline__ void rep_reduce(const double* stats, int stat_c, int sc, int r, bool ok, double& s, double& ss) {
    s = ok ? stats[((i64)r * 2 + 0) * stat_c + sc] : 0.0;
    ss = ok ? stats[((i64)r * 2 + 1) * stat_c + sc] : 0.0;
#pragma unroll
    for (int o = 16; o > 0; o >>= 1) { s += __shfl_xor(s, o); ss += __shfl_xor(ss, o); }
}

__device__ __forceinline__ void bn_finalize_train_body(const int blk, const double* stats, int stat_c, double count,
                                                       const float* gamma, const float* beta, float* running_mean,
                                                       float* running_var, long long* nbt, float momentum, float eps,
                                                       float* coef, int C) {
    const int r = threadIdx.x & 31;
    const int c = blk * 8 + (threadIdx.x >> 5);
    if (c == 0 && r == 0 && nbt) *nbt += 1;
    const bool ok = c < C;
    const int sc = ok ? c % stat_c : 0;      // shortcut BN: out channel c uses the stats of in channel c % C_in
    double s, ss;
    rep_reduce(stats, stat_c, sc, r, ok, s, ss);
    if (!ok || r != 0) return;
    double mean = s / count;
    double var = ss / count - mean * mean;
    if (var < 0) var = 0;
    float invstd = (float)(1.0 / sqrt(var + (double)eps));
    float scale = gamma[c] * invstd;
    coef[c] = scale;
    coef[C + c] = beta[c] - (float)mean * scale;
    coef[2 * C + c] = (float)mean;
    coef[3 * C + c] = invstd;
    if (running_mean) {
        double unbiased = count > 1 ? var * count / (count - 1) : var;
        running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * (float)mean;
        running_var[c] = (1.f - momentum) * running_var[c] + momentum * (float)unbiased;
    }
}
__global__ __launch_bounds__(256) void bn_finalize_train_kernel(const double* stats, int stat_c, double count,
                                                                const float* gamma, const float* beta,
                                                                float* running_mean, float* running_var, long long* nbt,
                                                                float momentum, float eps, float* coef, int C) {
    bn_finalize_train_body(blockIdx.x, stats, stat_c, count, gamma, beta, running_mean, running_var, nbt, momentum, eps, coef, C);
}
// two independent BatchNorms in one launch (conv_pwl.1.bn + bn_sc.bn; the cortex pair): every tiny launch costs ~4.5 us
// of GPU timeline on this part whatever its size
__global__ __launch_bounds__(256) void bn_finalize_train2_kernel(BnFinJob j0, BnFinJob j1, float momentum, float eps) {
    const bool first = (int)blockIdx.x < j0.nblocks;
    const BnFinJob& j = first ? j0 : j1;
    bn_finalize_train_body(first ? blockIdx.x : blockIdx.x - j0.nblocks, j.stats, j.stat_c, j.count, j.gamma, j.beta,
                           j.running_mean, j.running_var, j.nbt, momentum, eps, j.coef, j.C);
}

__global__ void bn_finalize_eval_kernel(const float* gamma, const float* beta, const float* running_mean,
                                        const float* running_var, float eps, float* coef, int C) {
    int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    float invstd = 1.0f / sqrtf(running_var[c] + eps);
    float scale = gamma[c] * invstd;
    coef[c] = scale;
    coef[C + c] = beta[c] - running_mean[c] * scale;
    coef[2 * C + c] = running_mean[c];
    coef[3 * C + c] = invstd;
}

// backward: stats = [NREP][2][C] with Σdh and Σdh·ŷ.  dy = A1*dh + A2*y + A3 (y raw), dgamma = Σdh·ŷ, dbeta = Σdh
__device__ __forceinline__ void bn_bwd_finalize_body(const int blk, const double* stats, double count, const float* coef,
                                                     float* dgamma, float* dbeta, float* abc, int C) {
    const int r = threadIdx.x & 31;
    const int c = blk * 8 + (threadIdx.x >> 5);
    const bool ok = c < C;
    double s1, s2;
    rep_reduce(stats, C, ok ? c : 0, r, ok, s1, s2);
    if (!ok || r != 0) return;
    float scale = coef[c], mean = coef[2 * C + c], invstd = coef[3 * C + c];
    if (dgamma) dgamma[c] = (float)s2;
    if (dbeta) dbeta[c] = (float)s1;
    double m1 = s1 / count, m2 = s2 / count;
    abc[c] = scale;
    abc[C + c] = (float)(-(double)scale * invstd * m2);
    abc[2 * C + c] = (float)((double)scale * (-m1 + (double)mean * invstd * m2));
}
__global__ __launch_bounds__(256) void bn_bwd_finalize_kernel(const double* stats, double count, const float* coef,
                                                              float* dgamma, float* dbeta, float* abc, int C) {
    bn_bwd_finalize_body(blockIdx.x, stats, count, coef, dgamma, dbeta, abc, C);
}
__global__ __launch_bounds__(256) void bn_bwd_finalize2_kernel(BnBwdJob j0, BnBwdJob j1) {
    const bool first = (int)blockIdx.x < j0.nblocks;
    const BnBwdJob& j = first ? j0 : j1;
    bn_bwd_finalize_body(first ? blockIdx.x : blockIdx.x - j0.nblocks, j.stats, j.count, j.coef, j.dgamma, j.dbeta, j.abc, j.C);
}

int k_bn_finalize_train(const double* stats, int stat_c, double count, const float* gamma, const float* beta,
                        float* rm, float* rv, long long* nbt, float momentum, float eps, float* coef, int C,
                        hipStream_t s) {
    hipLaunchKernelGGL(bn_finalize_train_kernel, dim3((C + 7) / 8), dim3(256), 0, s, stats, stat_c, count,
                       gamma, beta, rm, rv, nbt, momentum, eps, coef, C);
    DWN_CHECK_LAUNCH();
    return 0;
}
int k_bn_finalize_train2(BnFinJob j0, BnFinJob j1, float momentum, float eps, hipStream_t s) {
    j0.nblocks = (j0.C + 7) / 8; j1.nblocks = (j1.C + 7) / 8;
    hipLaunchKernelGGL(bn_finalize_train2_kernel, dim3(j0.nblocks + j1.nblocks), dim3(256), 0, s, j0, j1, momentum, eps);
    DWN_CHECK_LAUNCH();
    return 0;
}
int k_bn_bwd_finalize2(BnBwdJob j0, BnBwdJob j1, hipStream_t s) {
    j0.nblocks = (j0.C + 7) / 8; j1.nblocks = (j1.C + 7) / 8;
    hipLaunchKernelGGL(bn_bwd_finalize2_kernel, dim3(j0.nblocks + j1.nblocks), dim3(256), 0, s, j0, j1);
    DWN_CHECK_LAUNCH();
    return 0;
}
int k_bn_finalize_eval(const float* gamma, const float* beta, const float* rm, const float* rv, float eps,
                       float* coef, int C, hipStream_t s) {
    hipLaunchKernelGGL(bn_finalize_eval_kernel, dim3((C + 255) / 256), dim3(256), 0, s, gamma, beta, rm, rv, eps,
                       coef, C);
    DWN_CHECK_LAUNCH();
    return 0;
}
int k_bn_bwd_finalize(const double* stats, double count, const float* coef, float* dgamma, float* dbeta,
                      float* abc, int C, hipStream_t s) {
    hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3((C + 7) / 8), dim3(256), 0, s, stats, count, coef, dgamma,
                       dbeta, abc, C);
    DWN_CHECK_LAUNCH();
    return 0;
}

// ------------------------------------------------------------------------------------------------
// BatchNorm-1 batch statistics WITHOUT y1 (round 5).  y1 = a0 . W1^T is linear in the block input, so over the n rows
//   mean_e = w_e . mu,   var_e = w_e^T (G / n - mu mu^T) w_e     with G = a0^T a0, mu = (1^T a0) / n
// (gram = [(Cin + 8)][Cin] fp64: rows 0 .. Cin-1 = G, row Cin = 1^T a0 — the raw products a gemm_tn pass over a0 with the
// [a0 | 1] loader accumulates, per-workgroup fp32 MFMA tiles added with fp64 atomics (round 6: the difference G / n - mu mu^T
// amplifies the error of the SUM by (mean^2 + var) / var, and some hundred fp32 atomic adds carried 1e-6 of it);
// W1 as rounded to the compute type, i.e. the weights the MFMAs that rebuild y1 multiply with).
// These are the statistics of the UNROUNDED product; a stored bf16 y1 carries 2^-9 of unbiased rounding noise per element on
// top, which moves the mean by nothing and the variance by 3e-6 of itself.  One wave per channel, fp64 from the products on.
// sc_stats != NULL: the same raw products also are the shortcut BatchNorm's sums on an identity-map block (its input is a0):
// block 0 writes sum x = 1^T a0 and sum x^2 = diag(G) into replica 0 of that statistics buffer (zeroed by the caller).
// ------------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void bn1_gram_finalize_kernel(const double* __restrict__ gram, const float* __restrict__ w1,
                                                                int E, int Cin, double count, const float* gamma, const float* beta,
                                                                float* running_mean, float* running_var, long long* nbt,
                                                                float momentum, float eps, float* coef, double* sc_stats) {
    __shared__ float lw[4][512];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int e = blockIdx.x * 4 + wv;
    if (blockIdx.x == 0 && threadIdx.x == 0 && nbt) *nbt += 1;
    if (sc_stats && blockIdx.x == 0) {
        for (int c = threadIdx.x; c < Cin; c += 256) {
            sc_stats[c] = gram[(i64)Cin * Cin + c];
            sc_stats[Cin + c] = gram[(i64)c * Cin + c];
        }
    }
    if (e >= E) return;
    for (int j = lane; j < Cin; j += 64) lw[wv][j] = round_t<T>(w1[(i64)e * Cin + j]);
    __builtin_amdgcn_wave_barrier();
    const double inv_n = 1.0 / count;
    double quad = 0.0, lin = 0.0;
    for (int i = lane; i < Cin; i += 64) {
        const double* grow = gram + (i64)i * Cin;
        double r = 0.0;
        for (int j = 0; j < Cin; j += 2) {
            const double2 g2 = *reinterpret_cast<const double2*>(grow + j);
            r += g2.x * (double)lw[wv][j] + g2.y * (double)lw[wv][j + 1];
        }
        quad += (double)lw[wv][i] * r;
        lin += (double)lw[wv][i] * gram[(i64)Cin * Cin + i];
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { quad += __shfl_xor(quad, o); lin += __shfl_xor(lin, o); }
    if (lane != 0) return;
    const double mean = lin * inv_n;
    double var = quad * inv_n - mean * mean;
    if (var < 0) var = 0;
    const float invstd = (float)(1.0 / sqrt(var + (double)eps));
    const float scale = gamma[e] * invstd;
    coef[e] = scale;
    coef[E + e] = beta[e] - (float)mean * scale;
    coef[2 * E + e] = (float)mean;
    coef[3 * E + e] = invstd;
    if (running_mean) {
        const double unbiased = count > 1 ? var * count / (count - 1) : var;
        running_mean[e] = (1.f - momentum) * running_mean[e] + momentum * (float)mean;
        running_var[e] = (1.f - momentum) * running_var[e] + momentum * (float)unbiased;
    }
}
int k_bn1_gram_finalize(const double* gram, const float* w1, int E, int Cin, double count, const float* gamma, const float* beta,
                        float* rm, float* rv, long long* nbt, float momentum, float eps, float* coef, double* sc_stats, int dtype,
                        hipStream_t s) {
    if (Cin > 512 || Cin % 4) return dwn_set_error(-2, "bn1_gram_finalize: Cin must be a multiple of 4, at most 512");
    if (dtype == DWN_BF16)
        hipLaunchKernelGGL(bn1_gram_finalize_kernel<bf16_t>, dim3((E + 3) / 4), dim3(256), 0, s, gram, w1, E, Cin, count, gamma, beta,
                           rm, rv, nbt, momentum, eps, coef, sc_stats);
    else
        hipLaunchKernelGGL(bn1_gram_finalize_kernel<float>, dim3((E + 3) / 4), dim3(256), 0, s, gram, w1, E, Cin, count, gamma, beta,
                           rm, rv, nbt, momentum, eps, coef, sc_stats);
    DWN_CHECK_LAUNCH();
    return 0;
}

// ------------------------------------------------------------------------------------------------
// generic "materialise a loader" and per-channel statistics of a loader
// ------------------------------------------------------------------------------------------------
template <typename T, int KIND>
__global__ __launch_bounds__(256) void ew_apply_kernel(LoadDesc d, T* out, i64 ldo, i64 rows, int C) {
    SLICE_SETUP(C)
    if (!chan_ok) return;
    for (i64 row = (i64)blockIdx.x * 32 + pl; row < rows; row += (i64)gridDim.x * 32) {
        float v[KC];
        load_op<KIND, T>(d, row, chan, v);
        st_vec<T>(out + row * ldo + chan, v);
    }
}

template <typename T, int KIND>
__global__ __launch_bounds__(256) void colstats_kernel(LoadDesc d, i64 rows, int C, double* stats) {
    SLICE_SETUP(C)
    __shared__ float lstat[2 * NCV * KC];
    if (tid < 2 * NCV * KC) lstat[tid] = 0.f;
    __syncthreads();
    float s0[KC], s1[KC];
#pragma unroll
    for (int i = 0; i < KC; ++i) { s0[i] = 0.f; s1[i] = 0.f; }
    if (chan_ok)
        for (i64 row = (i64)blockIdx.x * 32 + pl; row < rows; row += (i64)gridDim.x * 32) {
            float v[KC];
            load_op<KIND, T>(d, row, chan, v);
#pragma unroll
            for (int i = 0; i < KC; ++i) { s0[i] += v[i]; s1[i] += v[i] * v[i]; }
        }
    slice_stats_flush<KC>(lstat, s0, s1, cv, c0, C, stats, blockIdx.x % DWN_NREP);
    DET_EXIT();
}

template <typename T>
static int ew_apply_t(const LoadDesc& d, int kind, void* out, i64 ldo, i64 rows, int C, hipStream_t s) {
    dim3 grid = slice_grid(rows, C, TT<T>::KC);
    T* o = reinterpret_cast<T*>(out);
    switch (kind) {
        case LD_PLAIN: hipLaunchKernelGGL((ew_apply_kernel<T, LD_PLAIN>), grid, dim3(256), 0, s, d, o, ldo, rows, C); break;
        case LD_PE: hipLaunchKernelGGL((ew_apply_kernel<T, LD_PE>), grid, dim3(256), 0, s, d, o, ldo, rows, C); break;
        case LD_BNACT: hipLaunchKernelGGL((ew_apply_kernel<T, LD_BNACT>), grid, dim3(256), 0, s, d, o, ldo, rows, C); break;
        case LD_AFFINE2: hipLaunchKernelGGL((ew_apply_kernel<T, LD_AFFINE2>), grid, dim3(256), 0, s, d, o, ldo, rows, C); break;
        case LD_DY3: hipLaunchKernelGGL((ew_apply_kernel<T, LD_DY3>), grid, dim3(256), 0, s, d, o, ldo, rows, C); break;
        case LD_GATE: hipLaunchKernelGGL((ew_apply_kernel<T, LD_GATE>), grid, dim3(256), 0, s, d, o, ldo, rows, C); break;
        default: return dwn_set_error(-3, "ew_apply: bad loader kind");
    }
    DWN_CHECK_LAUNCH();
    return 0;
}
int k_ew_apply(const LoadDesc& d, int kind, void* out, i64 ldo, i64 rows, int C, int dtype, hipStream_t s) {
    return dtype == DWN_BF16 ? ew_apply_t<bf16_t>(d, kind, out, ldo, rows, C, s)
                             : ew_apply_t<float>(d, kind, out, ldo, rows, C, s);
}
template <typename T>
static int colstats_t(const LoadDesc& d, int kind, i64 rows, int C, double* stats, hipStream_t s) {
    dim3 grid = slice_grid(rows, C, TT<T>::KC, 1024);
    switch (kind) {
        case LD_PLAIN: hipLaunchKernelGGL((colstats_kernel<T, LD_PLAIN>), grid, dim3(256), 0, s, d, rows, C, stats); break;
        case LD_PE: hipLaunchKernelGGL((colstats_kernel<T, LD_PE>), grid, dim3(256), 0, s, d, rows, C, stats); break;
        case LD_BNACT: hipLaunchKernelGGL((colstats_kernel<T, LD_BNACT>), grid, dim3(256), 0, s, d, rows, C, stats); break;
        default: return dwn_set_error(-3, "colstats: bad loader kind");
    }
    DWN_CHECK_LAUNCH();
    return 0;
}
int k_colstats(const LoadDesc& d, int kind, i64 rows, int C, double* stats, int dtype, hipStream_t s) {
    return dtype == DWN_BF16 ? colstats_t<bf16_t>(d, kind, rows, C, stats, s) : colstats_t<float>(d, kind, rows, C, stats, s);
}

// ------------------------------------------------------------------------------------------------
// stem, round 3: BatchNorm through the input moments.  y0 = W0 x is LINEAR in the 5 input channels, so the batch statistics
// of every output channel follow from Σx (Cin values) and Σ x x^T (Cin x Cin) — a pass over the 47 MB input instead of a
// write + read of the 300 MB y0 — and the backward needs only Σ dout (x - x̄)^T (C0 x Cin) and Σ dout:
//   mean_c = w_c . x̄,  var_c = w_c^T Cov w_c;   dW_c = A1 Σ dout_c (x - x̄) + A2 n Cov w_c   (Σ dy0 = 0 kills the A3 term)
// y0 is never materialised: forward = moments pass + one pass x -> out, backward = one pass over (dout, x).
// xmom (saved): [Cin] Σx, [Cin][Cin] Σ x x^T as doubles.
// ------------------------------------------------------------------------------------------------
constexpr int STEM_MAXCIN = 8;
constexpr int STEM_NM = STEM_MAXCIN + STEM_MAXCIN * STEM_MAXCIN;          // moments per replica (padded to the maximum Cin)

__global__ __launch_bounds__(256) void stem_xmom_kernel(const float* x, int B, int Cin, i64 S, double* mom) {
    __shared__ double lsum[STEM_NM];
    const int tid = threadIdx.x;
    if (tid < STEM_NM) lsum[tid] = 0.0;
    __syncthreads();
    float sx[STEM_MAXCIN], xx[STEM_MAXCIN][STEM_MAXCIN];
#pragma unroll
    for (int a_ = 0; a_ < STEM_MAXCIN; ++a_) {
        sx[a_] = 0.f;
#pragma unroll
        for (int b_ = 0; b_ < STEM_MAXCIN; ++b_) xx[a_][b_] = 0.f;
    }
    const i64 rows = (i64)B * S;
    for (i64 row = (i64)blockIdx.x * 256 + tid; row < rows; row += (i64)gridDim.x * 256) {
        const i64 b = row / S, sp = row - b * S;
        float xv[STEM_MAXCIN];
#pragma unroll
        for (int c = 0; c < STEM_MAXCIN; ++c) xv[c] = c < Cin ? x[(b * Cin + c) * S + sp] : 0.f;
#pragma unroll
        for (int a_ = 0; a_ < STEM_MAXCIN; ++a_) {
            sx[a_] += xv[a_];
#pragma unroll
            for (int b_ = a_; b_ < STEM_MAXCIN; ++b_) xx[a_][b_] = fmaf(xv[a_], xv[b_], xx[a_][b_]);
        }
    }
    // wave reduction in double, then one LDS add per wave and value
    DET_WAVES_BEGIN
#pragma unroll
    for (int a_ = 0; a_ < STEM_MAXCIN; ++a_) {
        if (a_ >= Cin) break;
        double v = (double)sx[a_];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
        if ((tid & 63) == 0) atomicAdd(&lsum[a_], v);
#pragma unroll
        for (int b_ = a_; b_ < STEM_MAXCIN; ++b_) {
            if (b_ >= Cin) break;
            double q = (double)xx[a_][b_];
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) q += __shfl_xor(q, o);
            if ((tid & 63) == 0) atomicAdd(&lsum[STEM_MAXCIN + a_ * STEM_MAXCIN + b_], q);
        }
    }
    DET_WAVES_END
    __syncthreads();
    DET_ENTER();
    if (tid < STEM_NM) {
        const double v = lsum[tid];
        if (v != 0.0) atomicAdd(mom + (i64)(blockIdx.x % DWN_NREP) * STEM_NM + tid, v);
    }
    DET_EXIT();
}

// one workgroup: replica reduce of the moments -> xmom, then per output channel the BatchNorm coefficients + running stats
__global__ __launch_bounds__(256) void stem_bn_finalize_kernel(const double* mom, double count, const float* w, const float* gamma,
                                                               const float* beta, float* running_mean, float* running_var,
                                                               long long* nbt, float momentum, float eps, float* coef,
                                                               double* xmom, int C0, int Cin) {
    __shared__ double lm[STEM_NM];
    const int tid = threadIdx.x;
    if (tid < STEM_NM) {
        double v = 0.0;
        for (int r = 0; r < DWN_NREP; ++r) v += mom[(i64)r * STEM_NM + tid];
        lm[tid] = v;
    }
    __syncthreads();
    if (tid < STEM_NM) {                    // upper triangle was accumulated: mirror it, save Σx and Σxx^T for the backward
        const int a_ = tid < STEM_MAXCIN ? 0 : (tid - STEM_MAXCIN) / STEM_MAXCIN, b_ = tid < STEM_MAXCIN ? 0 : (tid - STEM_MAXCIN) % STEM_MAXCIN;
        double v = lm[tid];
        if (tid >= STEM_MAXCIN && b_ < a_) v = lm[STEM_MAXCIN + b_ * STEM_MAXCIN + a_];
        xmom[tid] = v;
    }
    if (tid == 0 && nbt) *nbt += 1;
    for (int c = tid; c < C0; c += blockDim.x) {
        double mean = 0.0, var = 0.0;
        for (int a_ = 0; a_ < Cin; ++a_) mean += (double)w[c * Cin + a_] * (lm[a_] / count);
        for (int a_ = 0; a_ < Cin; ++a_)
            for (int b_ = 0; b_ < Cin; ++b_) {
                const double sxx = a_ <= b_ ? lm[STEM_MAXCIN + a_ * STEM_MAXCIN + b_] : lm[STEM_MAXCIN + b_ * STEM_MAXCIN + a_];
                const double cov = sxx / count - (lm[a_] / count) * (lm[b_] / count);
                var += (double)w[c * Cin + a_] * (double)w[c * Cin + b_] * cov;
            }
        if (var < 0) var = 0;
        const float invstd = (float)(1.0 / sqrt(var + (double)eps));
        const float scale = gamma[c] * invstd;
        coef[c] = scale;
        coef[C0 + c] = beta[c] - (float)mean * scale;
        coef[2 * C0 + c] = (float)mean;
        coef[3 * C0 + c] = invstd;
        if (running_mean) {
            const double unbiased = count > 1 ? var * count / (count - 1) : var;
            running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * (float)mean;
            running_var[c] = (1.f - momentum) * running_var[c] + momentum * (float)unbiased;
        }
    }
}

// out[row][c] = BN(W0 x) (+ positional encoding of the first block), straight from the NCDHW input
// CINT: compile-time bound of the input-channel loops (5 = the reference's inputs, 8 = generic): with the generic bound the
// weight / accumulator arrays cost 60 more VGPRs and a wave per SIMD
template <typename T, int CINT>
__global__ __launch_bounds__(256) void stem_out_kernel(const float* x, const float* w, const float* coef, const float* pe_t,
                                                       const float* pe_h, const float* pe_w, int Tn, int H, int W, int B,
                                                       int Cin, i64 S, int C0, T* out) {
    SLICE_SETUP(C0)
    constexpr int STEM_MAXCIN = CINT;
    const int chs_ = chan_ok ? chan : 0;                   // channel-tail lanes keep running: they fetch x for their row group
    float wr[STEM_MAXCIN][KC], sc[KC], sh[KC];
#pragma unroll
    for (int c = 0; c < STEM_MAXCIN; ++c)
#pragma unroll
        for (int i = 0; i < KC; ++i) wr[c][i] = (c < Cin && chan_ok) ? w[(chan + i) * Cin + c] : 0.f;
    ld_coef<KC>(coef + chs_, sc);
    ld_coef<KC>(coef + C0 + chs_, sh);
    const RasterIdx ro(H, W);
    const UDiv32 dT((unsigned)Tn), dS((unsigned)S);
    constexpr int RU = 2;
    const unsigned nrows = (unsigned)((i64)B * S), stride = gridDim.x * 32u;
    for (unsigned row0 = blockIdx.x * 32u + pl; row0 < nrows; row0 += RU * stride) {
        // the 8 lanes of a row share its Cin input values: lane cv fetches channel cv (one load per lane and row instead of
        // Cin eight-fold redundant ones), the group then broadcasts them with lane shuffles
        float xm[RU], xv[RU][STEM_MAXCIN];
#pragma unroll
        for (int u = 0; u < RU; ++u) {
            const unsigned row = row0 + u * stride < nrows ? row0 + u * stride : row0;
            const unsigned b = dS.div(row), sp = row - b * (unsigned)S;
            xm[u] = cv < Cin ? x[((i64)b * Cin + cv) * S + sp] : 0.f;
        }
#pragma unroll
        for (int u = 0; u < RU; ++u)
#pragma unroll
            for (int c = 0; c < STEM_MAXCIN; ++c) xv[u][c] = __shfl(xm[u], (tid & 56) | c);
#pragma unroll
        for (int u = 0; u < RU; ++u) {
            const unsigned row = row0 + u * stride;
            if (row >= nrows || !chan_ok) break;
            float v[KC];
#pragma unroll
            for (int i = 0; i < KC; ++i) v[i] = 0.f;
#pragma unroll
            for (int c = 0; c < STEM_MAXCIN; ++c)
#pragma unroll
                for (int i = 0; i < KC; ++i) v[i] = fmaf(wr[c][i], xv[u][c], v[i]);
#pragma unroll
            for (int i = 0; i < KC; ++i) v[i] = fmaf(v[i], sc[i], sh[i]);
            if (pe_t) {
                unsigned f; int h, wq;
                ro.decode(row, f, h, wq);
                const unsigned t = f - dT.div(f) * (unsigned)Tn;
                float pa[KC], pb[KC], pc[KC];
                ld_coef<KC>(pe_t + (i64)t * C0 + chan, pa);
                ld_coef<KC>(pe_h + (i64)h * C0 + chan, pb);
                ld_coef<KC>(pe_w + (i64)wq * C0 + chan, pc);
#pragma unroll
                for (int i = 0; i < KC; ++i) v[i] = round_t<T>(v[i]) + ((pa[i] + pb[i]) + pc[i]);
            }
            st_vec<T>(out + (i64)row * C0 + chan, v);
        }
    }
}

// backward accumulation: acc[rep][c][0..Cin-1] = Σ dout_c (x_k - x̄_k),  acc[rep][c][STEM_MAXCIN] = Σ dout_c
template <typename T, int CINT>
__global__ __launch_bounds__(256) void stem_bwd_acc_kernel(const T* dout, const float* x, const double* xmom, double count,
                                                           int B, int Cin, i64 S, int C0, double* acc) {
    SLICE_SETUP(C0)
    constexpr int NVG = ::STEM_MAXCIN + 1;                  // row stride of acc (fixed layout)
    constexpr int STEM_MAXCIN = CINT;
    constexpr int NV = STEM_MAXCIN + 1;
    __shared__ float lacc[NV * NCV * KC];
    for (int i = tid; i < NV * NCV * KC; i += 256) lacc[i] = 0.f;
    __syncthreads();
    float a_[NV][KC];
#pragma unroll
    for (int c = 0; c < NV; ++c)
#pragma unroll
        for (int i = 0; i < KC; ++i) a_[c][i] = 0.f;
    const float xbar_mine = cv < Cin ? (float)(xmom[cv] / count) : 0.f;
    const unsigned rows = (unsigned)((i64)B * S);
    const UDiv32 dS((unsigned)S);
    const int chs_ = chan_ok ? chan : 0;                   // channel-tail lanes keep running: they fetch x for their row group
    {
        constexpr int RU = 4;
        const unsigned stride = gridDim.x * 32u;
        for (unsigned row0 = blockIdx.x * 32u + pl; row0 < rows; row0 += RU * stride) {
            float g[RU][KC], xm[RU], xv[RU][STEM_MAXCIN];
#pragma unroll
            for (int u = 0; u < RU; ++u) {
                const unsigned row = row0 + u * stride < rows ? row0 + u * stride : row0;
                const unsigned b = dS.div(row), sp = row - b * (unsigned)S;
                ld_vec<T>(dout + (i64)row * C0 + chs_, g[u]);
                xm[u] = cv < Cin ? x[((i64)b * Cin + cv) * S + sp] - xbar_mine : 0.f;      // lane cv: channel cv of its row
            }
#pragma unroll
            for (int u = 0; u < RU; ++u)
#pragma unroll
                for (int c = 0; c < STEM_MAXCIN; ++c) xv[u][c] = __shfl(xm[u], (tid & 56) | c);
#pragma unroll
            for (int u = 0; u < RU; ++u) {
                if (row0 + u * stride >= rows || !chan_ok) break;
#pragma unroll
                for (int c = 0; c < STEM_MAXCIN; ++c)
#pragma unroll
                    for (int i = 0; i < KC; ++i) a_[c][i] = fmaf(g[u][i], xv[u][c], a_[c][i]);
#pragma unroll
                for (int i = 0; i < KC; ++i) a_[STEM_MAXCIN][i] += g[u][i];
            }
        }
    }
    // lanes l, l+8, ... of a wave share the channel vector: xor-shuffle them together, one LDS add per wave and value
    DET_WAVES_BEGIN
#pragma unroll
    for (int c = 0; c < NV; ++c) {
        if (c < STEM_MAXCIN && c >= Cin) continue;
#pragma unroll
        for (int i = 0; i < KC; ++i) {
            float v = a_[c][i];
#pragma unroll
            for (int o = 8; o < 64; o <<= 1) v += __shfl_xor(v, o);
            if ((tid & 63) < NCV) atomicAdd(&lacc[c * NCV * KC + cv * KC + i], v);
        }
    }
    DET_WAVES_END
    __syncthreads();
    DET_ENTER();
    for (int i = tid; i < NV * NCV * KC; i += 256) {
        const int c = i / (NCV * KC), o = c0 + i % (NCV * KC);
        if ((c == STEM_MAXCIN || c < Cin) && o < C0)
            atomicAdd(acc + ((i64)(blockIdx.x % DWN_NREP) * C0 + o) * NVG + (c == STEM_MAXCIN ? ::STEM_MAXCIN : c), (double)lacc[i]);
    }
    DET_EXIT();
}

// one workgroup: dgamma, dbeta, dW from the accumulated sums (see the header of this section)
__global__ __launch_bounds__(256) void stem_bwd_finalize_kernel(const double* acc, const double* xmom, const float* w,
                                                                const float* coef, double count, float* dgamma, float* dbeta,
                                                                float* dw, int C0, int Cin) {
    constexpr int NV = STEM_MAXCIN + 1;
    // four lanes per channel, eight replicas each (a single-thread loop over 32 replicas is a chain of dependent loads)
    const int part = threadIdx.x & 3;
    for (int cb = 0; cb < C0; cb += blockDim.x / 4) {
        const int c = cb + (threadIdx.x >> 2);
        const bool ok = c < C0;
        double dx[STEM_MAXCIN], s1 = 0.0;
        for (int k = 0; k < STEM_MAXCIN; ++k) dx[k] = 0.0;
        for (int r = part * (DWN_NREP / 4); r < (part + 1) * (DWN_NREP / 4); ++r) {
            const double* p = acc + ((i64)r * C0 + (ok ? c : 0)) * NV;
            for (int k = 0; k < Cin; ++k) dx[k] += p[k];
            s1 += p[STEM_MAXCIN];
        }
        for (int o = 1; o < 4; o <<= 1) {
            for (int k = 0; k < STEM_MAXCIN; ++k) dx[k] += __shfl_xor(dx[k], o);
            s1 += __shfl_xor(s1, o);
        }
        if (!ok || part != 0) continue;
        const double scale = coef[c], invstd = coef[3 * C0 + c];
        double dyc = 0.0;                                   // Σ dout (y0 - mean)
        for (int k = 0; k < Cin; ++k) dyc += (double)w[c * Cin + k] * dx[k];
        const double s2 = invstd * dyc;                     // Σ dout ŷ
        if (dgamma) dgamma[c] = (float)s2;
        if (dbeta) dbeta[c] = (float)s1;
        const double a2 = -scale * invstd * (s2 / count);
        for (int k = 0; k < Cin; ++k) {
            double cw = 0.0;                                // n (Cov w)_k = Σ_l w_l (Σ x_l x_k - n x̄_l x̄_k)
            for (int l = 0; l < Cin; ++l)
                cw += (double)w[c * Cin + l] * (xmom[STEM_MAXCIN + l * STEM_MAXCIN + k] - xmom[l] * xmom[k] / count);
            dw[c * Cin + k] = (float)(scale * dx[k] + a2 * cw);
        }
    }
}

int k_stem_xmom(const float* x, int B, int Cin, i64 S, double* mom, hipStream_t s) {
    if (Cin > STEM_MAXCIN) return dwn_set_error(-4, "stem: in_channels > 8 not built");
    i64 bx = ((i64)B * S + 255) / 256;
    if (bx > 1024) bx = 1024;
    hipLaunchKernelGGL(stem_xmom_kernel, dim3((unsigned)bx), dim3(256), 0, s, x, B, Cin, S, mom);
    DWN_CHECK_LAUNCH();
    return 0;
}
int k_stem_bn_finalize(const double* mom, double count, const float* w, const float* gamma, const float* beta, float* rm,
                       float* rv, long long* nbt, float momentum, float eps, float* coef, double* xmom, int C0, int Cin,
                       hipStream_t s) {
    hipLaunchKernelGGL(stem_bn_finalize_kernel, dim3(1), dim3(256), 0, s, mom, count, w, gamma, beta, rm, rv, nbt, momentum, eps,
                       coef, xmom, C0, Cin);
    DWN_CHECK_LAUNCH();
    return 0;
}
template <typename T, int CINT>
static void stem_out_launch(const float* x, const float* w, const float* coef, const float* pe_t, const float* pe_h, const float* pe_w,
                            int Tn, int H, int W, int B, int Cin, i64 S, int C0, void* out, hipStream_t s) {
    auto kern = stem_out_kernel<T, CINT>;
    hipLaunchKernelGGL(kern, resident_slice_grid(kern, (i64)B * S, C0, TT<T>::KC), dim3(256), 0, s, x, w, coef, pe_t, pe_h, pe_w, Tn, H, W,
                       B, Cin, S, C0, (T*)out);
}
int k_stem_out(const float* x, const float* w, const float* coef, const float* pe_t, const float* pe_h, const float* pe_w,
               int Tn, int H, int W, int B, int Cin, i64 S, int C0, void* out, int dtype, hipStream_t s) {
    if (Cin > STEM_MAXCIN) return dwn_set_error(-4, "stem: in_channels > 8 not built");
    if ((i64)B * S >= (1ll << 31)) return dwn_set_error(-4, "stem: more than 2^31 rows");
    if (Cin <= 5) {
        DISPATCH_T(dtype, (stem_out_launch<bf16_t, 5>(x, w, coef, pe_t, pe_h, pe_w, Tn, H, W, B, Cin, S, C0, out, s)),
                   (stem_out_launch<float, 5>(x, w, coef, pe_t, pe_h, pe_w, Tn, H, W, B, Cin, S, C0, out, s)));
    } else {
        DISPATCH_T(dtype, (stem_out_launch<bf16_t, 8>(x, w, coef, pe_t, pe_h, pe_w, Tn, H, W, B, Cin, S, C0, out, s)),
                   (stem_out_launch<float, 8>(x, w, coef, pe_t, pe_h, pe_w, Tn, H, W, B, Cin, S, C0, out, s)));
    }
    DWN_CHECK_LAUNCH();
    return 0;
}
template <typename T, int CINT>
static void stem_bwd_acc_launch(const void* dout, const float* x, const double* xmom, double count, int B, int Cin, i64 S, int C0,
                                double* acc, hipStream_t s) {
    auto kern = stem_bwd_acc_kernel<T, CINT>;
    // exactly one resident round of workgroups (accumulators are flushed once per workgroup; no ragged second round)
    hipLaunchKernelGGL(kern, resident_slice_grid(kern, (i64)B * S, C0, TT<T>::KC), dim3(256), 0, s, (const T*)dout, x, xmom, count, B, Cin,
                       S, C0, acc);
}
int k_stem_bwd_acc(const void* dout, const float* x, const double* xmom, double count, int B, int Cin, i64 S, int C0,
                   double* acc, int dtype, hipStream_t s) {
    if (Cin > STEM_MAXCIN) return dwn_set_error(-4, "stem: in_channels > 8 not built");
    if ((i64)B * S >= (1ll << 31)) return dwn_set_error(-4, "stem: more than 2^31 rows");
    if (Cin <= 5) {
        DISPATCH_T(dtype, (stem_bwd_acc_launch<bf16_t, 5>(dout, x, xmom, count, B, Cin, S, C0, acc, s)),
                   (stem_bwd_acc_launch<float, 5>(dout, x, xmom, count, B, Cin, S, C0, acc, s)));
    } else {
        DISPATCH_T(dtype, (stem_bwd_acc_launch<bf16_t, 8>(dout, x, xmom, count, B, Cin, S, C0, acc, s)),
                   (stem_bwd_acc_launch<float, 8>(dout, x, xmom, count, B, Cin, S, C0, acc, s)));
    }
    DWN_CHECK_LAUNCH();
    return 0;
}
int k_stem_bwd_finalize(const double* acc, const double* xmom, const float* w, const float* coef, double count, float* dgamma,
                        float* dbeta, float* dw, int C0, int Cin, hipStream_t s) {
    hipLaunchKernelGGL(stem_bwd_finalize_kernel, dim3(1), dim3(256), 0, s, acc, xmom, w, coef, count, dgamma, dbeta, dw, C0, Cin);
    DWN_CHECK_LAUNCH();
    return 0;
}
int stem_moment_count() { return STEM_NM; }
int stem_acc_stride() { return STEM_MAXCIN + 1; }


// ------------------------------------------------------------------------------------------------
// shortcut (interpolate_shortcut, dwiseneuro.py:125-134) + residual (:143)
// geometry of one block's shortcut gather
// ------------------------------------------------------------------------------------------------
// All kernels of this section walk output (or input) rows with 32-bit indices (rows < 2^31 is checked by the
// launchers), decode (frame, y, x) with UDiv32 and keep RU rows in flight per thread.
#define RES_RU 2
// The nearest-neighbour index tables (<= RES_TAB entries each) are staged in LDS once per workgroup: reading them from
// global memory put two dependent memory latencies in front of every gathered row.
#define RES_TAB 256
#define RES_STAGE_TABLES(hptr, hn, wptr, wn)                                             \
    __shared__ int s_ht[RES_TAB], s_wt[RES_TAB];                                         \
    const bool tab_lds = (hn) <= RES_TAB && (wn) <= RES_TAB;                             \
    if (tab_lds) {                                                                       \
        for (int i_ = threadIdx.x; i_ < (hn); i_ += blockDim.x) s_ht[i_] = (hptr)[i_];          \
        for (int i_ = threadIdx.x; i_ < (wn); i_ += blockDim.x) s_wt[i_] = (wptr)[i_];          \
    }                                                                                    \
    __syncthreads();                                                                     \
    const int* const ht_ = tab_lds ? s_ht : (hptr);                                      \
    const int* const wt_ = tab_lds ? s_wt : (wptr);
// Few, fat workgroups (2 per CU, SC_RU rows in flight per thread): with one or two channel slices a fully resident
// grid put 64 workgroups on every replica address of the fp64 statistics and their end-of-kernel atomics serialised at
// the memory side (~45 us fixed, more than the data pass itself).
#define SC_RU 4
#define STAT_NT 1024        // threads per workgroup of the small-C statistics kernels: one workgroup per CU
template <typename T>
__global__ __launch_bounds__(STAT_NT) void shortcut_stats_kernel(LoadDesc xin, ResGeom gm, double* stats) {
    // Σ, Σ² of (x + PE) at the gathered (nearest) positions, per *input* channel
    SLICE_SETUP(gm.Cin)
    __shared__ float lstat[2 * NCV * KC];
    if (tid < 2 * NCV * KC) lstat[tid] = 0.f;
    __syncthreads();
    float s0[KC], s1[KC];
#pragma unroll
    for (int i = 0; i < KC; ++i) { s0[i] = 0.f; s1[i] = 0.f; }
    const unsigned rows = (unsigned)gm.BT * gm.Hout * gm.Wout;
    const RasterIdx ro(gm.Hout, gm.Wout);
    const T* xp = reinterpret_cast<const T*>(xin.p);
    RES_STAGE_TABLES(gm.hsrc, gm.Hout, gm.wsrc, gm.Wout)
    if (chan_ok) {
        const unsigned stride = gridDim.x * (unsigned)(STAT_NT / NCV);
        for (unsigned row0 = blockIdx.x * (unsigned)(STAT_NT / NCV) + pl; row0 < rows; row0 += SC_RU * stride) {
            uint4 raw[SC_RU];
#pragma unroll
            for (int u = 0; u < SC_RU; ++u) {
                const unsigned row = row0 + u * stride < rows ? row0 + u * stride : row0;
                unsigned bt; int ho, wo;
                ro.decode(row, bt, ho, wo);
                const i64 rin = ((i64)bt * gm.Hin + ht_[ho]) * gm.Win + wt_[wo];
                raw[u] = *reinterpret_cast<const uint4*>(xp + rin * xin.ld + chan);
            }
#pragma unroll
            for (int u = 0; u < SC_RU; ++u) {
                if (row0 + u * stride >= rows) break;
                float v[KC];
                unpack16<T>(raw[u], v);
#pragma unroll
                for (int i = 0; i < KC; ++i) { s0[i] += v[i]; s1[i] += v[i] * v[i]; }
            }
        }
    }
    slice_stats_flush<KC>(lstat, s0, s1, cv, c0, gm.Cin, stats, blockIdx.x % DWN_NREP);
    DET_EXIT();
}

// out[m_out][c'] = d[b] * (s4*y4 + t4) + ssc*s + tsc,  s = (x+PE)[m_in][c' % Cin]
template <typename T>
__global__ __launch_bounds__(256) void residual_fwd_kernel(LoadDesc xin, const T* y4, const float* coef4,
                                                           const float* coefsc, const float* dscale, ResGeom gm,
                                                           const float* ope_t, const float* ope_h, const float* ope_w,
                                                           T* out) {
    SLICE_SETUP(gm.Cout)
    RES_STAGE_TABLES(gm.hsrc, gm.Hout, gm.wsrc, gm.Wout)
    if (!chan_ok) return;
    float s4[KC], t4[KC], ss[KC], ts[KC];
    ld_coef<KC>(coef4 + chan, s4); ld_coef<KC>(coef4 + gm.Cout + chan, t4);
    ld_coef<KC>(coefsc + chan, ss); ld_coef<KC>(coefsc + gm.Cout + chan, ts);
    const int csrc = chan % gm.Cin;
    const unsigned rows = (unsigned)gm.BT * gm.Hout * gm.Wout;
    const RasterIdx ro(gm.Hout, gm.Wout);
    const UDiv32 dT((unsigned)gm.T);
    const T* xp = reinterpret_cast<const T*>(xin.p);
    const unsigned stride = gridDim.x * 32u;
    for (unsigned row0 = blockIdx.x * 32u + pl; row0 < rows; row0 += RES_RU * stride) {
        uint4 rs[RES_RU], ry[RES_RU];
        unsigned btv[RES_RU]; int hov[RES_RU], wov[RES_RU];
#pragma unroll
        for (int u = 0; u < RES_RU; ++u) {
            const unsigned row = row0 + u * stride < rows ? row0 + u * stride : row0;
            ro.decode(row, btv[u], hov[u], wov[u]);
            const i64 rin = ((i64)btv[u] * gm.Hin + ht_[hov[u]]) * gm.Win + wt_[wov[u]];
            rs[u] = *reinterpret_cast<const uint4*>(xp + rin * xin.ld + csrc);
            ry[u] = *reinterpret_cast<const uint4*>(y4 + (i64)row * gm.Cout + chan);
        }
#pragma unroll
        for (int u = 0; u < RES_RU; ++u) {
            const unsigned row = row0 + u * stride;
            if (row >= rows) break;
            float sv[KC], yv[KC], o[KC];
            unpack16<T>(rs[u], sv);
            unpack16<T>(ry[u], yv);
            const unsigned b = dT.div(btv[u]);
            const float d = dscale ? dscale[b] : 1.0f;
#pragma unroll
            for (int i = 0; i < KC; ++i) o[i] = d * fmaf(yv[i], s4[i], t4[i]) + fmaf(sv[i], ss[i], ts[i]);
            if (ope_t) {       // the NEXT block's positional encoding is folded into this block's output
                float pa[KC], pb[KC], pc[KC];
                ld_coef<KC>(ope_t + (i64)(btv[u] - b * (unsigned)gm.T) * gm.Cout + chan, pa);
                ld_coef<KC>(ope_h + (i64)hov[u] * gm.Cout + chan, pb);
                ld_coef<KC>(ope_w + (i64)wov[u] * gm.Cout + chan, pc);
#pragma unroll
                for (int i = 0; i < KC; ++i) o[i] = round_t<T>(o[i]) + ((pa[i] + pb[i]) + pc[i]);
            }
            st_vec<T>(out + (i64)row * gm.Cout + chan, o);
        }
    }
}

// Σ d·dout, Σ d·dout·ŷ4 -> stats4 ; Σ dout, Σ dout·ŝ -> statssc   (per out channel)
#define RR_RU 2
template <typename T>
__global__ __launch_bounds__(STAT_NT) void residual_bwd_reduce_kernel(LoadDesc xin, const T* y4, const T* dout,
                                                                  const float* coef4, const float* coefsc,
                                                                  const float* dscale, ResGeom gm, double* stats4,
                                                                  double* statssc) {
    SLICE_SETUP(gm.Cout)
    __shared__ float lstat[2 * NCV * KC];
    __shared__ float lstat2[2 * NCV * KC];
    if (tid < 2 * NCV * KC) { lstat[tid] = 0.f; lstat2[tid] = 0.f; }
    RES_STAGE_TABLES(gm.hsrc, gm.Hout, gm.wsrc, gm.Wout)
    float a0[KC], a1[KC], b0[KC], b1[KC];
#pragma unroll
    for (int i = 0; i < KC; ++i) { a0[i] = a1[i] = b0[i] = b1[i] = 0.f; }
    if (chan_ok) {
        float m4[KC], i4[KC], ms[KC], is[KC];
        ld_coef<KC>(coef4 + 2 * gm.Cout + chan, m4); ld_coef<KC>(coef4 + 3 * gm.Cout + chan, i4);
        ld_coef<KC>(coefsc + 2 * gm.Cout + chan, ms); ld_coef<KC>(coefsc + 3 * gm.Cout + chan, is);
        const int csrc = chan % gm.Cin;
        const unsigned rows = (unsigned)gm.BT * gm.Hout * gm.Wout;
        const RasterIdx ro(gm.Hout, gm.Wout);
        const UDiv32 dT((unsigned)gm.T);
        const T* xp = reinterpret_cast<const T*>(xin.p);
        const unsigned stride = gridDim.x * (unsigned)(STAT_NT / NCV);
        for (unsigned row0 = blockIdx.x * (unsigned)(STAT_NT / NCV) + pl; row0 < rows; row0 += RR_RU * stride) {
            uint4 rs[RR_RU], ry[RR_RU], rg[RR_RU];
            unsigned btv[RR_RU];
#pragma unroll
            for (int u = 0; u < RR_RU; ++u) {
                const unsigned row = row0 + u * stride < rows ? row0 + u * stride : row0;
                int ho, wo;
                ro.decode(row, btv[u], ho, wo);
                const i64 rin = ((i64)btv[u] * gm.Hin + ht_[ho]) * gm.Win + wt_[wo];
                rs[u] = *reinterpret_cast<const uint4*>(xp + rin * xin.ld + csrc);
                ry[u] = *reinterpret_cast<const uint4*>(y4 + (i64)row * gm.Cout + chan);
                rg[u] = *reinterpret_cast<const uint4*>(dout + (i64)row * gm.Cout + chan);
            }
#pragma unroll
            for (int u = 0; u < RR_RU; ++u) {
                if (row0 + u * stride >= rows) break;
                float sv[KC], yv[KC], g[KC];
                unpack16<T>(rs[u], sv);
                unpack16<T>(ry[u], yv);
                unpack16<T>(rg[u], g);
                const float d = dscale ? dscale[dT.div(btv[u])] : 1.0f;
#pragma unroll
                for (int i = 0; i < KC; ++i) {
                    float gd = g[i] * d;
                    a0[i] += gd;
                    a1[i] += gd * (yv[i] - m4[i]) * i4[i];
                    b0[i] += g[i];
                    b1[i] += g[i] * (sv[i] - ms[i]) * is[i];
                }
            }
        }
    }
    slice_stats_flush<KC>(lstat, a0, a1, cv, c0, gm.Cout, stats4, blockIdx.x % DWN_NREP);
    __syncthreads();
    slice_stats_flush<KC>(lstat2, b0, b1, cv, c0, gm.Cout, statssc, blockIdx.x % DWN_NREP);
    DET_EXIT();
}

// dy4[m][c'] = A1*(d*dout) + A2*y4 + A3
template <typename T>
__global__ __launch_bounds__(256) void residual_bwd_dy4_kernel(const T* y4, const T* dout, const float* abc4,
                                                               const float* dscale, ResGeom gm, T* dy4) {
    SLICE_SETUP(gm.Cout)
    if (!chan_ok) return;
    float a1[KC], a2[KC], a3[KC];
    ld_coef<KC>(abc4 + chan, a1); ld_coef<KC>(abc4 + gm.Cout + chan, a2); ld_coef<KC>(abc4 + 2 * gm.Cout + chan, a3);
    const unsigned rows = (unsigned)gm.BT * gm.Hout * gm.Wout;
    const UDiv32 dS((unsigned)(gm.T * gm.Hout * gm.Wout));
    const unsigned stride = gridDim.x * 32u;
    for (unsigned row0 = blockIdx.x * 32u + pl; row0 < rows; row0 += RES_RU * stride) {
        uint4 ry[RES_RU], rg[RES_RU];
#pragma unroll
        for (int u = 0; u < RES_RU; ++u) {
            const unsigned row = row0 + u * stride < rows ? row0 + u * stride : row0;
            ry[u] = *reinterpret_cast<const uint4*>(y4 + (i64)row * gm.Cout + chan);
            rg[u] = *reinterpret_cast<const uint4*>(dout + (i64)row * gm.Cout + chan);
        }
#pragma unroll
        for (int u = 0; u < RES_RU; ++u) {
            const unsigned row = row0 + u * stride;
            if (row >= rows) break;
            float yv[KC], g[KC], o[KC];
            unpack16<T>(ry[u], yv);
            unpack16<T>(rg[u], g);
            const float d = dscale ? dscale[dS.div(row)] : 1.0f;
#pragma unroll
            for (int i = 0; i < KC; ++i) o[i] = fmaf(a1[i], g[i] * d, fmaf(a2[i], yv[i], a3[i]));
            st_vec<T>(dy4 + (i64)row * gm.Cout + chan, o);
        }
    }
}

// dx[m_in][c] = da0[m_in][c] + [m_in gathered] sum_{c' = c + j*Cin < Cout} (A1sc*dout[m_out][c'] + A2sc*s + A3sc)
template <typename T>
__global__ __launch_bounds__(256) void residual_bwd_dx_kernel(LoadDesc xin, const T* da0, const T* dout,
                                                              const float* abcsc, ResGeom gm, T* dx) {
    SLICE_SETUP(gm.Cin)
    RES_STAGE_TABLES(gm.hinv, gm.Hin, gm.winv, gm.Win)
    if (!chan_ok) return;
    const unsigned rows = (unsigned)gm.BT * gm.Hin * gm.Win;
    const RasterIdx ri(gm.Hin, gm.Win);
    const T* xp = reinterpret_cast<const T*>(xin.p);
    const unsigned stride = gridDim.x * 32u;
    for (unsigned row0 = blockIdx.x * 32u + pl; row0 < rows; row0 += RES_RU * stride) {
        uint4 ra[RES_RU], rx[RES_RU];
        i64 routv[RES_RU];
#pragma unroll
        for (int u = 0; u < RES_RU; ++u) {
            const unsigned row = row0 + u * stride < rows ? row0 + u * stride : row0;
            unsigned bt; int hi, wi;
            ri.decode(row, bt, hi, wi);
            const int ho = ht_[hi], wo = wt_[wi];
            routv[u] = (ho >= 0 && wo >= 0) ? ((i64)bt * gm.Hout + ho) * gm.Wout + wo : -1;
            ra[u] = *reinterpret_cast<const uint4*>(da0 + (i64)row * gm.Cin + chan);
            // x only feeds the shortcut's BatchNorm backward: rows the nearest map skips (3 of 4 on a stride-2 block) do not read it
            rx[u] = make_uint4(0, 0, 0, 0);
            if (routv[u] >= 0) rx[u] = *reinterpret_cast<const uint4*>(xp + (i64)row * xin.ld + chan);
        }
#pragma unroll
        for (int u = 0; u < RES_RU; ++u) {
            const unsigned row = row0 + u * stride;
            if (row >= rows) break;
            float o[KC];
            unpack16<T>(ra[u], o);
            if (routv[u] >= 0) {
                float sv[KC];
                unpack16<T>(rx[u], sv);
                for (int cc = chan; cc < gm.Cout; cc += gm.Cin) {
                    float g[KC], a1[KC], a2[KC], a3[KC];
                    ld_vec<T>(dout + routv[u] * gm.Cout + cc, g);
                    ld_coef<KC>(abcsc + cc, a1); ld_coef<KC>(abcsc + gm.Cout + cc, a2);
                    ld_coef<KC>(abcsc + 2 * gm.Cout + cc, a3);
#pragma unroll
                    for (int i = 0; i < KC; ++i) o[i] += fmaf(a1[i], g[i], fmaf(a2[i], sv[i], a3[i]));
                }
            }
            st_vec<T>(dx + (i64)row * gm.Cin + chan, o);
        }
    }
}


int k_shortcut_stats(const LoadDesc& xin, const ResGeom& gm, double* stats, int dtype, hipStream_t s) {
    i64 rows = (i64)gm.BT * gm.Hout * gm.Wout;
    DISPATCH_T(dtype,
        hipLaunchKernelGGL((shortcut_stats_kernel<bf16_t>), fat_slice_grid(rows, gm.Cin, 8), dim3(STAT_NT), 0, s, xin, gm, stats),
        hipLaunchKernelGGL((shortcut_stats_kernel<float>), fat_slice_grid(rows, gm.Cin, 4), dim3(STAT_NT), 0, s, xin, gm, stats));
    DWN_CHECK_LAUNCH();
    return 0;
}
int k_residual_fwd(const LoadDesc& xin, const void* y4, const float* coef4, const float* coefsc, const float* dscale,
                   const ResGeom& gm, const float* ope_t, const float* ope_h, const float* ope_w, void* out, int dtype,
                   hipStream_t s) {
    i64 rows = (i64)gm.BT * gm.Hout * gm.Wout;
    DISPATCH_T(dtype,
        hipLaunchKernelGGL((residual_fwd_kernel<bf16_t>), slice_grid(rows, gm.Cout, 8), dim3(256), 0, s, xin, (const bf16_t*)y4, coef4, coefsc, dscale, gm, ope_t, ope_h, ope_w, (bf16_t*)out),
        hipLaunchKernelGGL((residual_fwd_kernel<float>), slice_grid(rows, gm.Cout, 4), dim3(256), 0, s, xin, (const float*)y4, coef4, coefsc, dscale, gm, ope_t, ope_h, ope_w, (float*)out));
    DWN_CHECK_LAUNCH();
    return 0;
}
int k_residual_bwd_reduce(const LoadDesc& xin, const void* y4, const void* dout, const float* coef4,
                          const float* coefsc, const float* dscale, const ResGeom& gm, double* stats4,
                          double* statssc, int dtype, hipStream_t s) {
    i64 rows = (i64)gm.BT * gm.Hout * gm.Wout;
    DISPATCH_T(dtype,
        hipLaunchKernelGGL((residual_bwd_reduce_kernel<bf16_t>), fat_slice_grid(rows, gm.Cout, 8), dim3(STAT_NT), 0, s, xin, (const bf16_t*)y4, (const bf16_t*)dout, coef4, coefsc, dscale, gm, stats4, statssc),
        hipLaunchKernelGGL((residual_bwd_reduce_kernel<float>), fat_slice_grid(rows, gm.Cout, 4), dim3(STAT_NT), 0, s, xin, (const float*)y4, (const float*)dout, coef4, coefsc, dscale, gm, stats4, statssc));
    DWN_CHECK_LAUNCH();
    return 0;
}
int k_residual_bwd_dy4(const void* y4, const void* dout, const float* abc4, const float* dscale, const ResGeom& gm,
                       void* dy4, int dtype, hipStream_t s) {
    i64 rows = (i64)gm.BT * gm.Hout * gm.Wout;
    DISPATCH_T(dtype,
        hipLaunchKernelGGL((residual_bwd_dy4_kernel<bf16_t>), slice_grid(rows, gm.Cout, 8), dim3(256), 0, s, (const bf16_t*)y4, (const bf16_t*)dout, abc4, dscale, gm, (bf16_t*)dy4),
        hipLaunchKernelGGL((residual_bwd_dy4_kernel<float>), slice_grid(rows, gm.Cout, 4), dim3(256), 0, s, (const float*)y4, (const float*)dout, abc4, dscale, gm, (float*)dy4));
    DWN_CHECK_LAUNCH();
    return 0;
}
int k_residual_bwd_dx(const LoadDesc& xin, const void* da0, const void* dout, const float* abcsc, const ResGeom& gm,
                      void* dx, int dtype, hipStream_t s) {
    i64 rows = (i64)gm.BT * gm.Hin * gm.Win;
    DISPATCH_T(dtype,
        hipLaunchKernelGGL((residual_bwd_dx_kernel<bf16_t>), slice_grid(rows, gm.Cin, 8), dim3(256), 0, s, xin, (const bf16_t*)da0, (const bf16_t*)dout, abcsc, gm, (bf16_t*)dx),
        hipLaunchKernelGGL((residual_bwd_dx_kernel<float>), slice_grid(rows, gm.Cin, 4), dim3(256), 0, s, xin, (const float*)da0, (const float*)dout, abcsc, gm, (float*)dx));
    DWN_CHECK_LAUNCH();
    return 0;
}

// ------------------------------------------------------------------------------------------------
// Squeeze-Excite (SqueezeExcite3d, dwiseneuro.py:38-43)
// ------------------------------------------------------------------------------------------------
// pooled[b][c] += sum over a chunk of the sample's rows of silu(bn3(y3)), in 64-bit fixed point (pool_fix, dwn_common.h):
// a thread's own sum runs over a fixed set of rows in a fixed order, and everything after it is integer addition, so the
// pooled sums — and with them the gate and the whole forward pass — do not depend on the order in which waves and
// workgroups arrive (no ordering needed in the deterministic build either)
template <typename T>
__global__ __launch_bounds__(256) void se_pool_kernel(LoadDesc z3, int C, int rows_per_sample, int chunks,
                                                      long long* pooled, T* z3out) {
    SLICE_SETUP(C)
    __shared__ unsigned long long lacc[NCV * KC];
    if (tid < NCV * KC) lacc[tid] = 0ull;
    __syncthreads();
    const int b = blockIdx.x / chunks, chunk = blockIdx.x % chunks;
    const int per = (rows_per_sample + chunks - 1) / chunks;
    const int r_beg = chunk * per;
    const int r_end = (r_beg + per < rows_per_sample) ? r_beg + per : rows_per_sample;
    float acc[KC];
#pragma unroll
    for (int i = 0; i < KC; ++i) acc[i] = 0.f;
    if (chan_ok) {
        float sc[KC], sh[KC];
        ld_coef<KC>(z3.v1 + chan, sc);
        ld_coef<KC>(z3.v2 + chan, sh);
        const T* yp = reinterpret_cast<const T*>(z3.p);
        for (int r0 = r_beg + pl; r0 < r_end; r0 += 4 * 32) {
            uint4 raw[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                int r = r0 + 32 * u;
                raw[u] = *reinterpret_cast<const uint4*>(yp + ((i64)b * rows_per_sample + (r < r_end ? r : r_beg)) * z3.ld + chan);
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                int r = r0 + 32 * u;
                if (r >= r_end) break;
                float v[KC];
                unpack16<T>(raw[u], v);
#pragma unroll
                for (int i = 0; i < KC; ++i) v[i] = fmaf(v[i], sc[i], sh[i]);
                silu_n<KC>(v);
                if (z3out) st_vec<T>(z3out + ((i64)b * rows_per_sample + r) * C + chan, v);
#pragma unroll
                for (int i = 0; i < KC; ++i) acc[i] += round_t<T>(v[i]);
            }
        }
    }
#pragma unroll
    for (int i = 0; i < KC; ++i) atomicAdd(&lacc[cv * KC + i], (unsigned long long)pool_fix(acc[i]));
    __syncthreads();
    if (tid < NCV * KC && c0 + tid < C)
        atomicAdd(reinterpret_cast<unsigned long long*>(pooled) + (i64)b * C + c0 + tid, lacc[tid]);
}

int k_se_pool(const LoadDesc& z3, int B, int C, int rows_per_sample, long long* pooled, void* z3out, int dtype, hipStream_t s) {
    int KCv = dtype == DWN_BF16 ? 8 : 4;
    int slices = (C + NCV * KCv - 1) / (NCV * KCv);
    int chunks = (rows_per_sample + 255) / 256;
    int maxchunks = (2048 + B * slices - 1) / (B * slices);
    if (chunks > maxchunks) chunks = maxchunks;
    if (chunks < 1) chunks = 1;
    dim3 grid(B * chunks, slices);
    DISPATCH_T(dtype,
        hipLaunchKernelGGL((se_pool_kernel<bf16_t>), grid, dim3(256), 0, s, z3, C, rows_per_sample, chunks, pooled, (bf16_t*)z3out),
        hipLaunchKernelGGL((se_pool_kernel<float>), grid, dim3(256), 0, s, z3, C, rows_per_sample, chunks, pooled, (float*)z3out));
    DWN_CHECK_LAUNCH();
    return 0;
}

// SE MLP forward.  grid = (B, SPLIT): every workgroup recomputes the tiny hidden layer (R x C MACs) of its sample
// and produces a 1/SPLIT slice of the C gates, so the launch fills the chip instead of B = 32 workgroups.
__global__ __launch_bounds__(256) void se_mlp_fwd_kernel(const long long* pooled_sum, float inv_s, const float* wr,
                                                         const float* br, const float* we, const float* be, int C,
                                                         int R, float* pmean, float* hid_pre, float* gate) {
    extern __shared__ float sh[];          // [C] mean + [R] hid
    float* pm = sh;
    float* hid = sh + C;
    const int b = blockIdx.x, tid = threadIdx.x;
    const int split = gridDim.y, part = blockIdx.y;
    for (int c = tid; c < C; c += 256) {
        float v = pool_unfix(pooled_sum[(i64)b * C + c]) * inv_s;
        pm[c] = v;
        if (part == 0) pmean[(i64)b * C + c] = v;
    }
    __syncthreads();
    const int wave = tid >> 6, lane = tid & 63;
    for (int r = wave; r < R; r += 4) {
        float acc = 0.f;
        for (int c = lane; c < C; c += 64) acc = fmaf(wr[(i64)r * C + c], pm[c], acc);
        for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o);
        if (lane == 0) {
            float h = acc + br[r];
            if (part == 0) hid_pre[(i64)b * R + r] = h;
            hid[r] = siluf_(h);
        }
    }
    __syncthreads();
    const int per = (C + split - 1) / split;
    const int c_end = (part + 1) * per < C ? (part + 1) * per : C;
    for (int c = part * per + tid; c < c_end; c += 256) {
        float acc = be[c];
        for (int r = 0; r < R; ++r) acc = fmaf(we[(i64)c * R + r], hid[r], acc);
        gate[(i64)b * C + c] = sigmoidf_(acc);
    }
}

// SE MLP backward, data path.  grid = (B, SPLIT): dgp = dg*g*(1-g); dhid = We^T dgp (recomputed per workgroup);
// dhp = dhid*silu'(hid_pre); dpS = (Wr^T dhp)/S for a 1/SPLIT slice of the channels.
__global__ __launch_bounds__(256) void se_mlp_bwd_kernel(const float* dg, const float* gate, const float* hid_pre,
                                                         const float* wr, const float* we, int C, int R, float inv_s,
                                                         float* dgp_out, float* dhp_out, float* dps) {
    extern __shared__ float sh[];          // [C] dgp + [R] dhp
    float* dgp = sh;
    float* dhp = sh + C;
    const int b = blockIdx.x, tid = threadIdx.x;
    const int split = gridDim.y, part = blockIdx.y;
    for (int c = tid; c < C; c += 256) {
        float g = gate[(i64)b * C + c];
        float v = dg[(i64)b * C + c] * g * (1.f - g);
        dgp[c] = v;
        if (part == 0) dgp_out[(i64)b * C + c] = v;
    }
    __syncthreads();
    const int wave = tid >> 6, lane = tid & 63;
    for (int r = wave; r < R; r += 4) {
        float acc = 0.f;
        for (int c = lane; c < C; c += 64) acc = fmaf(we[(i64)c * R + r], dgp[c], acc);
        for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o);
        if (lane == 0) {
            float v = acc * silu_gradf_(hid_pre[(i64)b * R + r]);
            dhp[r] = v;
            if (part == 0) dhp_out[(i64)b * R + r] = v;
        }
    }
    __syncthreads();
    const int per = (C + split - 1) / split;
    const int c_end = (part + 1) * per < C ? (part + 1) * per : C;
    for (int c = part * per + tid; c < c_end; c += 256) {
        float acc = 0.f;
        for (int r = 0; r < R; ++r) acc = fmaf(wr[(i64)r * C + c], dhp[r], acc);
        dps[(i64)b * C + c] = acc * inv_s;
    }
}

// ---- latency-oriented variants for R <= SE_RT hidden units (the model's R = Cmid / 32 <= 56).
// The generic kernels above walk the hidden units one after another per wave: ~R/4 dependent rounds of L2 latency
// (40-50 us per launch); they remain for R > SE_RT and odd R.
#define SE_RT 64
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

// The squeeze layer by ROWS.  Wave w owns hidden units w, w+4, ... (RP/4 of them, all unrolled) and walks
// the channels with 16-byte loads (lane -> 4 consecutive channels, 256 channels per wave step): every load independent,
// one xor-shuffle sum per owned unit and NO LDS atomics (a hidden unit has exactly one owner wave — nothing to order in the deterministic build either).
// Requires C % 4 == 0 (16-byte rows of `wr`).
// w2 != NULL: the kernel also writes the gate folded into conv_pwl's weight for its sample and channel range,
// wg[b][n][c] = w2[n][c] * gate[b][c] in the compute type (dwiseneuro.py:40-43,117-120; what k_gate_weights did in a launch of its
// own, 9 launches per step)
template <int RP>
__global__ __launch_bounds__(256) void se_mlp_fwd_rows_kernel(const long long* pooled_sum, float inv_s, const float* wr,
                                                              const float* br, const float* we, const float* be, int C,
                                                              int R, float* pmean, float* hid_pre, float* gate,
                                                              const float* __restrict__ w2, void* __restrict__ wg, int N2, int wg_bf16) {
    constexpr int RPW = RP / 4;
    __shared__ float hid[RP];
    __shared__ float gl[256];
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int split = gridDim.y, part = blockIdx.y;
    const int C4 = C >> 2;                                  // float4 per row
    float acc[RPW];
#pragma unroll
    for (int q = 0; q < RPW; ++q) acc[q] = 0.f;
    const longlong2* ps2 = reinterpret_cast<const longlong2*>(pooled_sum + (i64)b * C);
    float4* pm4 = reinterpret_cast<float4*>(pmean + (i64)b * C);
    for (int c4 = lane; c4 < C4; c4 += 64) {
        const longlong2 p0 = ps2[2 * c4], p1 = ps2[2 * c4 + 1];
        float4 v;
        v.x = pool_unfix(p0.x) * inv_s; v.y = pool_unfix(p0.y) * inv_s; v.z = pool_unfix(p1.x) * inv_s; v.w = pool_unfix(p1.y) * inv_s;
        if (part == 0 && wave == 0) pm4[c4] = v;
        float4 w[RPW];
#pragma unroll
        for (int q = 0; q < RPW; ++q) {
            const int r = wave + 4 * q;
            w[q] = reinterpret_cast<const float4*>(wr + (i64)(r < R ? r : R - 1) * C)[c4];
        }
#pragma unroll
        for (int q = 0; q < RPW; ++q)
            acc[q] = fmaf(w[q].x, v.x, fmaf(w[q].y, v.y, fmaf(w[q].z, v.z, fmaf(w[q].w, v.w, acc[q]))));
    }
#pragma unroll
    for (int q = 0; q < RPW; ++q) {
        const float t = wave_sum(acc[q]);
        const int r = wave + 4 * q;
        if (lane == 0 && r < R) {
            const float h = t + br[r];
            if (part == 0) hid_pre[(i64)b * R + r] = h;
            hid[r] = siluf_(h);
        }
    }
    __syncthreads();
    const int per = (C + split - 1) / split;
    const int c_end = (part + 1) * per < C ? (part + 1) * per : C;
    for (int c = part * per + tid; c < c_end; c += 256) {
        float a = be[c];
        const float* wrow = we + (i64)c * R;
#pragma unroll 1
        for (int r0 = 0; r0 < RP; r0 += 16) {
            if (r0 >= R) break;
            float wv[16];
            if ((R & 3) == 0) {
                const float4* w4 = reinterpret_cast<const float4*>(wrow);
                const int nq = R >> 2;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int qi = (r0 >> 2) + q;
                    const float4 t = w4[qi < nq ? qi : nq - 1];
                    wv[4 * q] = t.x; wv[4 * q + 1] = t.y; wv[4 * q + 2] = t.z; wv[4 * q + 3] = t.w;
                }
            } else {
#pragma unroll
                for (int r = 0; r < 16; ++r) wv[r] = wrow[r0 + r < R ? r0 + r : R - 1];
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) a = fmaf(wv[r], r0 + r < R ? hid[r0 + r] : 0.f, a);
        }
        const float gv = sigmoidf_(a);
        gate[(i64)b * C + c] = gv;
        if (w2 && c - part * per < 256) gl[c - part * per] = gv;
    }
    if (w2) {                                   // (uniform: per <= 256 is checked by the launcher)
        __syncthreads();
        const int c0 = part * per, nc = c_end - c0;
        if (nc > 0) {
            const int tot = N2 * nc;
            for (int i = tid; i < tot; i += 256) {
                const int n = i / nc, cc = i - n * nc;
                const float v = w2[(i64)n * C + c0 + cc] * gl[cc];
                const i64 o = ((i64)b * N2 + n) * C + c0 + cc;
                if (wg_bf16) reinterpret_cast<bf16_t*>(wg)[o] = f2bf(v);
                else reinterpret_cast<float*>(wg)[o] = v;
            }
        }
    }
}

// Folding wave reduction: every lane holds N partial sums (N a power of two <= 64); afterwards lane l holds the wave total
// of value l % N.  At distance d a lane keeps the half of its live values whose index bit matches its own lane bit and
// adds the partner's copy of them: N - 1 + (plain steps) shuffles instead of 6 N for N separate butterfly sums.
template <int N, int LIVE, int D> struct WaveFold {
    static __device__ __forceinline__ void run(float (&v)[N], int lane) {
        if constexpr (D > 0) {
            if constexpr (LIVE > D) {
                constexpr int H = LIVE / 2;
                const bool up = (lane & D) != 0;
#pragma unroll
                for (int i = 0; i < H; ++i) {
                    const float send = up ? v[i] : v[i + H];
                    const float keep = up ? v[i + H] : v[i];
                    v[i] = keep + __shfl_xor(send, D);
                }
                WaveFold<N, H, D / 2>::run(v, lane);
            } else {
#pragma unroll
                for (int i = 0; i < LIVE; ++i) v[i] += __shfl_xor(v[i], D);
                WaveFold<N, LIVE, D / 2>::run(v, lane);
            }
        }
    }
};
template <int N>
__device__ __forceinline__ float wave_fold(float (&v)[N], int lane) {
    WaveFold<N, N, 32>::run(v, lane);
    return v[0];
}

// Backward data path: all RP partial sums of the excite layer's transpose product in one pass (16-/8-byte loads of the
// `we` rows), one folding reduction per wave, the four wave totals combined through LDS slots — no LDS atomics (nothing to
// order in the deterministic build).  Requires R even.
template <int RP, int VW>
__global__ __launch_bounds__(256) void se_mlp_bwd_fold_kernel(const float* dg, const float* gate, const float* hid_pre,
                                                              const float* wr, const float* we, int C, int R,
                                                              float inv_s, float* dgp_out, float* dhp_out, float* dps) {
    __shared__ float part4[4][RP];
    __shared__ float dhp[RP];
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int split = gridDim.y, part = blockIdx.y;
    float acc[RP];
#pragma unroll
    for (int r = 0; r < RP; ++r) acc[r] = 0.f;
    for (int c = tid; c < C; c += 256) {
        const float g = gate[(i64)b * C + c];
        const float v = dg[(i64)b * C + c] * g * (1.f - g);
        if (part == 0) dgp_out[(i64)b * C + c] = v;
        const float* wrow = we + (i64)c * R;
        if constexpr (VW == 4) {
            const float4* w4 = reinterpret_cast<const float4*>(wrow);
            const int nq = R >> 2;
            float4 t[RP / 4];
#pragma unroll
            for (int q = 0; q < RP / 4; ++q) t[q] = w4[q < nq ? q : nq - 1];
#pragma unroll
            for (int q = 0; q < RP / 4; ++q) {
                acc[4 * q] = fmaf(t[q].x, v, acc[4 * q]); acc[4 * q + 1] = fmaf(t[q].y, v, acc[4 * q + 1]);
                acc[4 * q + 2] = fmaf(t[q].z, v, acc[4 * q + 2]); acc[4 * q + 3] = fmaf(t[q].w, v, acc[4 * q + 3]);
            }
        } else {
            const float2* w2 = reinterpret_cast<const float2*>(wrow);
            const int nq = R >> 1;
            float2 t[RP / 2];
#pragma unroll
            for (int q = 0; q < RP / 2; ++q) t[q] = w2[q < nq ? q : nq - 1];
#pragma unroll
            for (int q = 0; q < RP / 2; ++q) {
                acc[2 * q] = fmaf(t[q].x, v, acc[2 * q]); acc[2 * q + 1] = fmaf(t[q].y, v, acc[2 * q + 1]);
            }
        }
    }
    const float tot = wave_fold<RP>(acc, lane);              // lane l: this wave's total of unit l % RP (units >= R: ignored)
    if (lane < RP) part4[wave][lane] = tot;
    __syncthreads();
    if (tid < RP) {
        const float sum = (part4[0][tid] + part4[1][tid]) + (part4[2][tid] + part4[3][tid]);
        float v = 0.f;
        if (tid < R) {
            v = sum * silu_gradf_(hid_pre[(i64)b * R + tid]);
            if (part == 0) dhp_out[(i64)b * R + tid] = v;
        }
        dhp[tid] = v;
    }
    __syncthreads();
    const int per = (C + split - 1) / split;
    const int c_end = (part + 1) * per < C ? (part + 1) * per : C;
    for (int c = part * per + tid; c < c_end; c += 256) {
        float a = 0.f;
#pragma unroll 1
        for (int r0 = 0; r0 < RP; r0 += 16) {
            if (r0 >= R) break;
            float wv[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) wv[r] = wr[(i64)(r0 + r < R ? r0 + r : R - 1) * C + c];
#pragma unroll
            for (int r = 0; r < 16; ++r) a = fmaf(wv[r], r0 + r < R ? dhp[r0 + r] : 0.f, a);
        }
        dps[(i64)b * C + c] = a * inv_s;
    }
}

// parameter grads of the SE MLP.  grid.x = ceil(C/256), grid.y = R: thread (c, r) reduces over the batch.
__global__ __launch_bounds__(256) void se_mlp_wgrad_kernel(const float* dgp, const float* dhp, const float* pmean,
                                                           const float* hid_pre, int B, int C, int R, float* dwr,
                                                           float* dbr, float* dwe, float* dbe) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    const int r = blockIdx.y;
    if (c < C) {
        float a = 0.f, a2 = 0.f, sb = 0.f;
#pragma unroll 8
        for (int b = 0; b < B; ++b) {
            float g = dgp[(i64)b * C + c];
            a = fmaf(g, siluf_(hid_pre[(i64)b * R + r]), a);
            a2 = fmaf(dhp[(i64)b * R + r], pmean[(i64)b * C + c], a2);
            sb += g;
        }
        dwe[(i64)c * R + r] = a;
        dwr[(i64)r * C + c] = a2;
        if (r == 0) dbe[c] = sb;
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        float sb = 0.f;
        for (int b = 0; b < B; ++b) sb += dhp[(i64)b * R + r];
        dbr[r] = sb;
    }
}

// w2 / wg / N2 (optional): fold the gate into conv_pwl's weight in the same launch (rows kernel only; returns 1 in *folded)
int k_se_mlp_fwd(const long long* pooled_sum, float inv_s, const float* wr, const float* br, const float* we,
                 const float* be, int B, int C, int R, float* pmean, float* hid_pre, float* gate, const float* w2, void* wg, int N2,
                 int dtype, int* folded, hipStream_t s) {
    if (folded) *folded = 0;
    if ((C & 3) == 0 && R <= SE_RT && !(((size_t)pooled_sum | (size_t)wr | (size_t)pmean) & 15)) {
        const bool fold = w2 && wg && (C + 7) / 8 <= 256;
        const float* w2f = fold ? w2 : nullptr;
        const int bf = dtype == DWN_BF16 ? 1 : 0;
        if (R <= 16)
            hipLaunchKernelGGL(se_mlp_fwd_rows_kernel<16>, dim3(B, 8), dim3(256), 0, s, pooled_sum, inv_s, wr, br, we, be, C, R,
                               pmean, hid_pre, gate, w2f, wg, N2, bf);
        else if (R <= 32)
            hipLaunchKernelGGL(se_mlp_fwd_rows_kernel<32>, dim3(B, 8), dim3(256), 0, s, pooled_sum, inv_s, wr, br, we, be, C, R,
                               pmean, hid_pre, gate, w2f, wg, N2, bf);
        else
            hipLaunchKernelGGL(se_mlp_fwd_rows_kernel<SE_RT>, dim3(B, 8), dim3(256), 0, s, pooled_sum, inv_s, wr, br, we, be, C,
                               R, pmean, hid_pre, gate, w2f, wg, N2, bf);
        if (folded && fold) *folded = 1;
    } else {
        hipLaunchKernelGGL(se_mlp_fwd_kernel, dim3(B, 8), dim3(256), (C + R) * sizeof(float), s, pooled_sum, inv_s, wr, br,
                           we, be, C, R, pmean, hid_pre, gate);
    }
    DWN_CHECK_LAUNCH();
    return 0;
}
int k_se_mlp_bwd(const float* dg, const float* gate, const float* hid_pre, const float* pmean, const float* wr,
                 const float* we, int B, int C, int R, float inv_s, float* dgp, float* dhp, float* dps, float* dwr,
                 float* dbr, float* dwe, float* dbe, hipStream_t s) {
    if ((R & 1) == 0 && R <= SE_RT && !((size_t)we & 15)) {
#define SE_BWD_FOLD(RP_) do { \
        if ((R & 3) == 0) hipLaunchKernelGGL((se_mlp_bwd_fold_kernel<RP_, 4>), dim3(B, 8), dim3(256), 0, s, dg, gate, hid_pre, wr, we, C, R, inv_s, dgp, dhp, dps); \
        else hipLaunchKernelGGL((se_mlp_bwd_fold_kernel<RP_, 2>), dim3(B, 8), dim3(256), 0, s, dg, gate, hid_pre, wr, we, C, R, inv_s, dgp, dhp, dps); } while (0)
        if (R <= 16) SE_BWD_FOLD(16);
        else if (R <= 32) SE_BWD_FOLD(32);
        else SE_BWD_FOLD(SE_RT);
#undef SE_BWD_FOLD
    } else {
        hipLaunchKernelGGL(se_mlp_bwd_kernel, dim3(B, 8), dim3(256), (C + R) * sizeof(float), s, dg, gate, hid_pre, wr, we,
                           C, R, inv_s, dgp, dhp, dps);
    }
    DWN_CHECK_LAUNCH();
    hipLaunchKernelGGL(se_mlp_wgrad_kernel, dim3((C + 255) / 256, R), dim3(256), 0, s, dgp, dhp, pmean, hid_pre, B, C, R,
                       dwr, dbr, dwe, dbe);
    DWN_CHECK_LAUNCH();
    return 0;
}

// Σdh3, Σdh3·ŷ3 where dh3 = (du*gate + dpS) * silu'(bn3(y3))   (LD_DY3 with A1=1, A2=A3=0 gives dh3)
template <typename T>
__global__ __launch_bounds__(256) void bn3_bwd_reduce_kernel(LoadDesc d, const float* coef3, i64 rows, int C,
                                                             double* stats, T* dh_out) {
    SLICE_SETUP(C)
    __shared__ float lstat[2 * NCV * KC];
    if (tid < 2 * NCV * KC) lstat[tid] = 0.f;
    __syncthreads();
    float s0[KC], s1[KC];
#pragma unroll
    for (int i = 0; i < KC; ++i) { s0[i] = 0.f; s1[i] = 0.f; }
    if (chan_ok) {
        float m3[KC], i3[KC];
        ld_coef<KC>(coef3 + 2 * C + chan, m3);
        ld_coef<KC>(coef3 + 3 * C + chan, i3);
        const T* yp = reinterpret_cast<const T*>(d.q);
        const T* pp = reinterpret_cast<const T*>(d.p);
        float sc[KC], sh[KC], g[KC], g2[KC];
        ld_coef<KC>(d.v4 + chan, sc);
        ld_coef<KC>(d.v5 + chan, sh);
        int gb = -1;
        const i64 stride = (i64)gridDim.x * 32;
        for (i64 row0 = (i64)blockIdx.x * 32 + pl; row0 < rows; row0 += 2 * stride) {
            uint4 rp[2], ry[2];
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                i64 row = row0 + u * stride;
                if (row >= rows) row = row0;
                rp[u] = *reinterpret_cast<const uint4*>(pp + row * d.ld + chan);
                ry[u] = *reinterpret_cast<const uint4*>(yp + row * d.ld + chan);
            }
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                i64 row = row0 + u * stride;
                if (row >= rows) break;
                int b = (int)((unsigned)row / (unsigned)d.rows_per_sample);
                if (b != gb) {
                    gb = b;
                    ld_coef<KC>(d.gate + (i64)b * d.gate_ld + chan, g);
                    ld_coef<KC>(d.gate2 + (i64)b * d.gate_ld + chan, g2);
                }
                float du[KC], y[KC], dh[KC];
                unpack16<T>(rp[u], du);
                unpack16<T>(ry[u], y);
                float h[KC], sp[KC];
#pragma unroll
                for (int i = 0; i < KC; ++i) h[i] = fmaf(y[i], sc[i], sh[i]);
                silu_grad_n<KC>(h, sp);
#pragma unroll
                for (int i = 0; i < KC; ++i) dh[i] = fmaf(du[i], g[i], g2[i]) * sp[i];
                if (dh_out) st_vec<T>(dh_out + row * d.ld + chan, dh);      // may alias d.p (element-wise in place)
#pragma unroll
                for (int i = 0; i < KC; ++i) { float r = round_t<T>(dh[i]); s0[i] += r; s1[i] += r * (y[i] - m3[i]) * i3[i]; }
            }
        }
    }
    slice_stats_flush<KC>(lstat, s0, s1, cv, c0, C, stats, blockIdx.x % DWN_NREP);
    DET_EXIT();
}
int k_bn3_bwd_reduce(const LoadDesc& d, const float* coef3, i64 rows, int C, double* stats, void* dh_out, int dtype, hipStream_t s) {
    DISPATCH_T(dtype,
        hipLaunchKernelGGL((bn3_bwd_reduce_kernel<bf16_t>), resident_slice_grid(bn3_bwd_reduce_kernel<bf16_t>, rows, C, 8), dim3(256), 0, s, d, coef3, rows, C, stats, (bf16_t*)dh_out),
        hipLaunchKernelGGL((bn3_bwd_reduce_kernel<float>), resident_slice_grid(bn3_bwd_reduce_kernel<float>, rows, C, 4), dim3(256), 0, s, d, coef3, rows, C, stats, (float*)dh_out));
    DWN_CHECK_LAUNCH();
    return 0;
}

// conv_pwl backward from per-sample products P_b = dy4_b^T z3_b ([B][K = Cout][N = Cmid], gemm_tn per-sample mode):
//   dW2[k][n] += sum_b gate[b][n] * P_b[k][n]        (the SE gate is constant over a sample, so it factors out)
//   dg[b][n]   = sum_k W2[k][n] * P_b[k][n]          (= sum_m du[m][n] * z3[m][n] with du = dy4 . W2)
// which replaces one full read of z3 and the write + read of du.  grid (N / 64, B / 4); thread = (column, k-lane).
__global__ __launch_bounds__(256) void pwl_bwd_reduce_kernel(const float* __restrict__ P, const float* __restrict__ gate,
                                                             const float* __restrict__ W, int B, int K, int N,
                                                             float* __restrict__ dW, float* __restrict__ dg) {
    // thread = (4 consecutive columns, one of 16 k-lanes): 16-byte loads, K/16 iterations with 4 samples in flight each
    // (the first version walked K/4 rows with scalar loads: 34 us for the 29 MB of block 6)
    __shared__ float red[4][16][64 + 4];
    const int nq = threadIdx.x & 15, kl = threadIdx.x >> 4;
    const int n = blockIdx.x * 64 + nq * 4, b0 = blockIdx.y * 4;
    float dgp[4][4], gt[4][4];
#pragma unroll
    for (int bb = 0; bb < 4; ++bb)
#pragma unroll
        for (int j = 0; j < 4; ++j) { dgp[bb][j] = 0.f; gt[bb][j] = 0.f; }
    const bool vec = (N & 3) == 0 && n + 3 < N;         // whole 16-byte group inside the row (N % 4 == 0 keeps it aligned)
    const bool nok = n < N;
    DET_ENTER();                 // dW receives this workgroup's sums from inside the k loop
    if (nok) {
#pragma unroll
        for (int bb = 0; bb < 4; ++bb)
            if (b0 + bb < B)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    if (n + j < N) gt[bb][j] = gate[(i64)(b0 + bb) * N + n + j];
        for (int k = kl; k < K; k += 16) {
            float w[4], dwa[4] = {0.f, 0.f, 0.f, 0.f}, p[4][4];
            if (vec) {
                const float4 w4 = *reinterpret_cast<const float4*>(W + (i64)k * N + n);
                w[0] = w4.x; w[1] = w4.y; w[2] = w4.z; w[3] = w4.w;
#pragma unroll
                for (int bb = 0; bb < 4; ++bb) {
                    const int b = b0 + bb < B ? b0 + bb : B - 1;
                    const float4 p4 = *reinterpret_cast<const float4*>(P + ((i64)b * K + k) * N + n);
                    p[bb][0] = p4.x; p[bb][1] = p4.y; p[bb][2] = p4.z; p[bb][3] = p4.w;
                }
            } else {
#pragma unroll
                for (int j = 0; j < 4; ++j) w[j] = n + j < N ? W[(i64)k * N + n + j] : 0.f;
#pragma unroll
                for (int bb = 0; bb < 4; ++bb) {
                    const int b = b0 + bb < B ? b0 + bb : B - 1;
#pragma unroll
                    for (int j = 0; j < 4; ++j) p[bb][j] = n + j < N ? P[((i64)b * K + k) * N + n + j] : 0.f;
                }
            }
#pragma unroll
            for (int bb = 0; bb < 4; ++bb)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    dgp[bb][j] = fmaf(w[j], p[bb][j], dgp[bb][j]);
                    dwa[j] = fmaf(gt[bb][j], p[bb][j], dwa[j]);          // gt = 0 for the clamped duplicates
                }
#pragma unroll
            for (int j = 0; j < 4; ++j)
                if (n + j < N) atomicAdd(dW + (i64)k * N + n + j, dwa[j]);
        }
    }
#pragma unroll
    for (int bb = 0; bb < 4; ++bb)
#pragma unroll
        for (int j = 0; j < 4; ++j) red[bb][kl][nq * 4 + j] = dgp[bb][j];
    __syncthreads();
    {   // 256 threads finish 4 samples x 64 columns: sum over the 16 k-lanes in a fixed order
        const int bb = threadIdx.x >> 6, nl = threadIdx.x & 63;
        const int nn = blockIdx.x * 64 + nl;
        if (nn < N && b0 + bb < B) {
            float t = 0.f;
#pragma unroll
            for (int q = 0; q < 16; ++q) t += red[bb][q][nl];
            dg[(i64)(b0 + bb) * N + nn] = t;
        }
    }
    DET_EXIT();
}
int k_pwl_bwd_reduce(const float* P, const float* gate, const float* W, int B, int K, int N, float* dW, float* dg,
                     hipStream_t s) {
    hipLaunchKernelGGL(pwl_bwd_reduce_kernel, dim3((N + 63) / 64, (B + 3) / 4), dim3(256), 0, s, P, gate, W, B, K, N, dW, dg);
    DWN_CHECK_LAUNCH();
    return 0;
}

// ------------------------------------------------------------------------------------------------
// H×W average pool (AdaptiveAvgPool3d((None,1,1)), dwiseneuro.py:374,400) and its backward
// ------------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void pool_fwd_kernel(const T* x, T* out, i64 BT, int HW, int C) {
    SLICE_SETUP(C)
    if (!chan_ok) return;
    const float inv = 1.0f / HW;
    for (i64 bt = (i64)blockIdx.x * 32 + pl; bt < BT; bt += (i64)gridDim.x * 32) {
        float acc[KC];
#pragma unroll
        for (int i = 0; i < KC; ++i) acc[i] = 0.f;
        for (int p = 0; p < HW; ++p) {
            float v[KC];
            ld_vec<T>(x + (bt * HW + p) * C + chan, v);
#pragma unroll
            for (int i = 0; i < KC; ++i) acc[i] += v[i];
        }
#pragma unroll
        for (int i = 0; i < KC; ++i) acc[i] *= inv;
        st_vec<T>(out + bt * C + chan, acc);
    }
}
template <typename T>
__global__ __launch_bounds__(256) void pool_bwd_kernel(const T* dpool, T* dx, i64 BT, int HW, int C) {
    SLICE_SETUP(C)
    if (!chan_ok) return;
    const float inv = 1.0f / HW;
    const i64 rows = BT * HW;
    for (i64 row = (i64)blockIdx.x * 32 + pl; row < rows; row += (i64)gridDim.x * 32) {
        float v[KC];
        ld_vec<T>(dpool + (row / HW) * C + chan, v);
#pragma unroll
        for (int i = 0; i < KC; ++i) v[i] *= inv;
        st_vec<T>(dx + row * C + chan, v);
    }
}
int k_pool_fwd(const void* x, void* out, i64 BT, int HW, int C, int dtype, hipStream_t s) {
    DISPATCH_T(dtype,
        hipLaunchKernelGGL((pool_fwd_kernel<bf16_t>), slice_grid(BT, C, 8), dim3(256), 0, s, (const bf16_t*)x, (bf16_t*)out, BT, HW, C),
        hipLaunchKernelGGL((pool_fwd_kernel<float>), slice_grid(BT, C, 4), dim3(256), 0, s, (const float*)x, (float*)out, BT, HW, C));
    DWN_CHECK_LAUNCH();
    return 0;
}
int k_pool_bwd(const void* dpool, void* dx, i64 BT, int HW, int C, int dtype, hipStream_t s) {
    DISPATCH_T(dtype,
        hipLaunchKernelGGL((pool_bwd_kernel<bf16_t>), slice_grid(BT * HW, C, 8), dim3(256), 0, s, (const bf16_t*)dpool, (bf16_t*)dx, BT, HW, C),
        hipLaunchKernelGGL((pool_bwd_kernel<float>), slice_grid(BT * HW, C, 4), dim3(256), 0, s, (const float*)dpool, (float*)dx, BT, HW, C));
    DWN_CHECK_LAUNCH();
    return 0;
}

// ------------------------------------------------------------------------------------------------
// cortex ShuffleLayer glue (dwiseneuro.py:212-234).  Tiny tensors ([B*T][<=4096]): scalar kernels.
//   out[m][j] = d[b]*silu(s[o]*y[m][o]+t[o]) + ssc[j]*x[m][j % Cin] + tsc[j],  o = (j % g)*(C/g) + j / g
// ------------------------------------------------------------------------------------------------
template <typename T>
__global__ void cortex_residual_fwd_kernel(const T* y, const T* x, const float* coef, const float* coefsc,
                                           const float* dscale, int M, int Tn, int Cin, int C, int groups, T* out) {
    i64 idx = (i64)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (i64)M * C) return;
    int j = (int)(idx % C);
    i64 m = idx / C;
    int o = (j % groups) * (C / groups) + j / groups;
    float h = fmaf(to_f<T>(y[m * C + o]), coef[o], coef[C + o]);
    float z = siluf_(h);
    float d = dscale ? dscale[m / Tn] : 1.0f;
    float sc = fmaf(to_f<T>(x[m * Cin + j % Cin]), coefsc[j], coefsc[C + j]);
    out[idx] = from_f<T>(d * z + sc);
}

// per-channel backward sums: main BN (indexed by o): Σdh, Σdh·ŷ ; shortcut BN (indexed by j): Σdout, Σdout·ŝ
// one thread per channel, looping over rows (M = B*T is ~1k)
template <typename T>
__global__ void cortex_bwd_reduce_kernel(const T* y, const T* x, const T* dout, const float* gmask, int gmask_ld,
                                         const float* coef, const float* coefsc, const float* dscale, int M, int Tn,
                                         int Cin, int C, int groups, int rows_per_chunk, double* stats,
                                         double* statssc) {
    int ch = blockIdx.x * blockDim.x + threadIdx.x;
    if (ch >= C) return;
    int m_beg = blockIdx.y * rows_per_chunk;
    int m_end = m_beg + rows_per_chunk < M ? m_beg + rows_per_chunk : M;
    // main path for o = ch : shuffled position j(o) = (o % (C/g))*g + o / (C/g)
    int o = ch;
    int j = (o % (C / groups)) * groups + o / (C / groups);
    float s = coef[o], t = coef[C + o], mean = coef[2 * C + o], invstd = coef[3 * C + o];
    float a0 = 0.f, a1 = 0.f;
#pragma unroll 8
    for (int m = m_beg; m < m_end; ++m) {
        float yv = to_f<T>(y[(i64)m * C + o]);
        float g = to_f<T>(dout[(i64)m * C + j]);
        if (gmask) g *= gmask[(i64)(m / Tn) * gmask_ld + j];
        float d = dscale ? dscale[m / Tn] : 1.0f;
        float dh = g * d * silu_gradf_(fmaf(yv, s, t));
        a0 += dh;
        a1 += dh * (yv - mean) * invstd;
    }
    int rep = blockIdx.y % DWN_NREP;
    stat_add(stats, rep, C, 0, o, a0);
    stat_add(stats, rep, C, 1, o, a1);
    // shortcut path for j = ch
    j = ch;
    float ms = coefsc[2 * C + j], is = coefsc[3 * C + j];
    float b0 = 0.f, b1 = 0.f;
#pragma unroll 8
    for (int m = m_beg; m < m_end; ++m) {
        float g = to_f<T>(dout[(i64)m * C + j]);
        if (gmask) g *= gmask[(i64)(m / Tn) * gmask_ld + j];
        float xv = to_f<T>(x[(i64)m * Cin + j % Cin]);
        b0 += g;
        b1 += g * (xv - ms) * is;
    }
    stat_add(statssc, rep, C, 0, j, b0);
    stat_add(statssc, rep, C, 1, j, b1);
}

// dy[m][o] = A1*dh + A2*y + A3 (main path, materialised for the GEMMs)
template <typename T>
__global__ void cortex_bwd_dy_kernel(const T* y, const T* dout, const float* gmask, int gmask_ld, const float* coef,
                                     const float* abc, const float* dscale, int M, int Tn, int C, int groups, T* dy) {
    i64 idx = (i64)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (i64)M * C) return;
    int o = (int)(idx % C);
    i64 m = idx / C;
    int j = (o % (C / groups)) * groups + o / (C / groups);
    float yv = to_f<T>(y[idx]);
    float g = to_f<T>(dout[m * C + j]);
    if (gmask) g *= gmask[(m / Tn) * gmask_ld + j];
    float d = dscale ? dscale[m / Tn] : 1.0f;
    float dh = g * d * silu_gradf_(fmaf(yv, coef[o], coef[C + o]));
    dy[idx] = from_f<T>(fmaf(abc[o], dh, fmaf(abc[C + o], yv, abc[2 * C + o])));
}

// dx[m][c] = dxmain[m][c] + sum_{j = c + k*Cin < C} (A1sc[j]*dout[m][j] + A2sc[j]*x[m][c] + A3sc[j])
template <typename T>
__global__ void cortex_bwd_dx_kernel(const T* dxmain, const T* x, const T* dout, const float* gmask, int gmask_ld,
                                     const float* abcsc, int M, int Tn, int Cin, int C, T* dx) {
    i64 idx = (i64)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (i64)M * Cin) return;
    int c = (int)(idx % Cin);
    i64 m = idx / Cin;
    float acc = to_f<T>(dxmain[idx]);
    float xv = to_f<T>(x[idx]);
    for (int j = c; j < C; j += Cin) {
        float g = to_f<T>(dout[m * C + j]);
        if (gmask) g *= gmask[(m / Tn) * gmask_ld + j];
        acc += fmaf(abcsc[j], g, fmaf(abcsc[C + j], xv, abcsc[2 * C + j]));
    }
    dx[idx] = from_f<T>(acc);
}

int k_cortex_residual_fwd(const void* y, const void* x, const float* coef, const float* coefsc, const float* dscale,
                          int M, int Tn, int Cin, int C, int groups, void* out, int dtype, hipStream_t s) {
    i64 n = (i64)M * C;
    dim3 grid((unsigned)((n + 255) / 256));
    DISPATCH_T(dtype,
        hipLaunchKernelGGL((cortex_residual_fwd_kernel<bf16_t>), grid, dim3(256), 0, s, (const bf16_t*)y, (const bf16_t*)x, coef, coefsc, dscale, M, Tn, Cin, C, groups, (bf16_t*)out),
        hipLaunchKernelGGL((cortex_residual_fwd_kernel<float>), grid, dim3(256), 0, s, (const float*)y, (const float*)x, coef, coefsc, dscale, M, Tn, Cin, C, groups, (float*)out));
    DWN_CHECK_LAUNCH();
    return 0;
}
int k_cortex_bwd_reduce(const void* y, const void* x, const void* dout, const float* gmask, int gmask_ld,
                        const float* coef, const float* coefsc, const float* dscale, int M, int Tn, int Cin, int C,
                        int groups, double* stats, double* statssc, int dtype, hipStream_t s) {
    int rows_per_chunk = 16;           // short per-thread row loops: the kernel is a latency chain, not a byte mover
    dim3 grid((C + 127) / 128, (M + rows_per_chunk - 1) / rows_per_chunk);
    DISPATCH_T(dtype,
        hipLaunchKernelGGL((cortex_bwd_reduce_kernel<bf16_t>), grid, dim3(128), 0, s, (const bf16_t*)y, (const bf16_t*)x, (const bf16_t*)dout, gmask, gmask_ld, coef, coefsc, dscale, M, Tn, Cin, C, groups, rows_per_chunk, stats, statssc),
        hipLaunchKernelGGL((cortex_bwd_reduce_kernel<float>), grid, dim3(128), 0, s, (const float*)y, (const float*)x, (const float*)dout, gmask, gmask_ld, coef, coefsc, dscale, M, Tn, Cin, C, groups, rows_per_chunk, stats, statssc));
    DWN_CHECK_LAUNCH();
    return 0;
}
int k_cortex_bwd_dy(const void* y, const void* dout, const float* gmask, int gmask_ld, const float* coef,
                    const float* abc, const float* dscale, int M, int Tn, int C, int groups, void* dy, int dtype,
                    hipStream_t s) {
    i64 n = (i64)M * C;
    dim3 grid((unsigned)((n + 255) / 256));
    DISPATCH_T(dtype,
        hipLaunchKernelGGL((cortex_bwd_dy_kernel<bf16_t>), grid, dim3(256), 0, s, (const bf16_t*)y, (const bf16_t*)dout, gmask, gmask_ld, coef, abc, dscale, M, Tn, C, groups, (bf16_t*)dy),
        hipLaunchKernelGGL((cortex_bwd_dy_kernel<float>), grid, dim3(256), 0, s, (const float*)y, (const float*)dout, gmask, gmask_ld, coef, abc, dscale, M, Tn, C, groups, (float*)dy));
    DWN_CHECK_LAUNCH();
    return 0;
}
int k_cortex_bwd_dx(const void* dxmain, const void* x, const void* dout, const float* gmask, int gmask_ld,
                    const float* abcsc, int M, int Tn, int Cin, int C, void* dx, int dtype, hipStream_t s) {
    i64 n = (i64)M * Cin;
    dim3 grid((unsigned)((n + 255) / 256));
    DISPATCH_T(dtype,
        hipLaunchKernelGGL((cortex_bwd_dx_kernel<bf16_t>), grid, dim3(256), 0, s, (const bf16_t*)dxmain, (const bf16_t*)x, (const bf16_t*)dout, gmask, gmask_ld, abcsc, M, Tn, Cin, C, (bf16_t*)dx),
        hipLaunchKernelGGL((cortex_bwd_dx_kernel<float>), grid, dim3(256), 0, s, (const float*)dxmain, (const float*)x, (const float*)dout, gmask, gmask_ld, abcsc, M, Tn, Cin, C, (float*)dx));
    DWN_CHECK_LAUNCH();
    return 0;
}

// ------------------------------------------------------------------------------------------------
// weight packing: fp32 parameter -> T, optionally transposed and zero padded
// dst[g][r][c] (rows R_dst, cols C_dst per group) = src[g][..] ; transpose: dst[g][c][r] = src[g][r][c]
// ------------------------------------------------------------------------------------------------
template <typename T>
__global__ void pack_weight_kernel(const float* src, T* dst, int groups, int R, int C, int transpose, int Rd, int Cd) {
    i64 idx = (i64)blockIdx.x * blockDim.x + threadIdx.x;
    i64 per = (i64)Rd * Cd;
    if (idx >= per * groups) return;
    int g = (int)(idx / per);
    i64 rem = idx % per;
    int rd = (int)(rem / Cd), cd = (int)(rem % Cd);
    float v = 0.f;
    if (!transpose) { if (rd < R && cd < C) v = src[((i64)g * R + rd) * C + cd]; }
    else { if (cd < R && rd < C) v = src[((i64)g * R + cd) * C + rd]; }
    dst[idx] = from_f<T>(v);
}
int k_pack_weight(const float* src, void* dst, int groups, int R, int C, int transpose, int Rd, int Cd, int dtype,
                  hipStream_t s) {
    i64 n = (i64)groups * Rd * Cd;
    dim3 grid((unsigned)((n + 255) / 256));
    DISPATCH_T(dtype,
        hipLaunchKernelGGL((pack_weight_kernel<bf16_t>), grid, dim3(256), 0, s, src, (bf16_t*)dst, groups, R, C, transpose, Rd, Cd),
        hipLaunchKernelGGL((pack_weight_kernel<float>), grid, dim3(256), 0, s, src, (float*)dst, groups, R, C, transpose, Rd, Cd));
    DWN_CHECK_LAUNCH();
    return 0;
}
// One pass over an fp32 weight [groups*R][C] for both operand layouts the readout needs (dwiseneuro.py:276-281):
//   plain[groups*R][C]   (forward: B operand, K = C contiguous) and
//   tr[g][C][Rp]         (data gradient: B operand, K = Rp contiguous; rows r >= R are zero)
// with row strides ldp / ldt (elements)
// 64 x 64 tiles: coalesced float4 reads, the transposed copy goes through LDS and leaves as 16-byte row segments.
// (The per-element pack_weight_kernel above read the transposed layout with a stride of C floats between neighbouring lanes:
// 53 us per 16 M-element readout weight and layout, twice per step.)
template <typename T>
__global__ __launch_bounds__(256) void pack_weight_dual_kernel(const float* src, T* plain, T* tr, int R, int C, int Rp, int ldp, int ldt) {
    constexpr int KC = TT<T>::KC;
    __shared__ float tile[64][65];
    const int tid = threadIdx.x;
    const int c0 = blockIdx.x * 64, r0 = blockIdx.y * 64, g = blockIdx.z;
    const int c4 = (tid & 15) * 4, rr = tid >> 4;
#pragma unroll
    for (int it = 0; it < 4; ++it) {
        const int r = r0 + rr + 16 * it, c = c0 + c4;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (r < R && c < C) {                             // C % 4 == 0 (checked by the launcher)
            v = *reinterpret_cast<const float4*>(src + ((i64)g * R + r) * C + c);
            if (plain) {
                T* d = plain + ((i64)g * R + r) * ldp + c;
                if constexpr (TT<T>::IS_BF16) {
                    uint2 o;
                    o.x = (unsigned)f2bf(v.x) | ((unsigned)f2bf(v.y) << 16);
                    o.y = (unsigned)f2bf(v.z) | ((unsigned)f2bf(v.w) << 16);
                    *reinterpret_cast<uint2*>(d) = o;
                } else {
                    *reinterpret_cast<float4*>(d) = v;
                }
            }
        }
        tile[rr + 16 * it][c4 + 0] = v.x; tile[rr + 16 * it][c4 + 1] = v.y;
        tile[rr + 16 * it][c4 + 2] = v.z; tile[rr + 16 * it][c4 + 3] = v.w;
    }
    if (!tr) return;
    __syncthreads();
    // transposed copy: thread -> (column c of the tile, KC consecutive r); Rp % 64 == 0 so whole segments are in range
    constexpr int SEG = 64 / KC;                          // 16-byte segments per transposed row of the tile
    for (int i = tid; i < 64 * SEG; i += 256) {
        const int cl = i / SEG, sg = i % SEG;
        const int c = c0 + cl;
        if (c >= C) continue;
        float v[KC];
#pragma unroll
        for (int k = 0; k < KC; ++k) v[k] = tile[sg * KC + k][cl];
        *reinterpret_cast<uint4*>(tr + ((i64)g * C + c) * ldt + r0 + sg * KC) = pack16<T>(v);
    }
}
int k_pack_weight_dual(const float* src, void* plain, void* tr, int groups, int R, int C, int Rp, int ldp, int ldt, int dtype,
                       hipStream_t s) {
    if (C % 4 || (tr && Rp % 64) || ldp % 8 || ldt % 8 || ldp < C || ldt < Rp) return dwn_set_error(-2, "pack_weight_dual: C % 4 == 0 and Rp % 64 == 0 required");
    const int rows = tr ? Rp : R;
    dim3 grid((C + 63) / 64, (rows + 63) / 64, groups);
    DISPATCH_T(dtype,
        hipLaunchKernelGGL((pack_weight_dual_kernel<bf16_t>), grid, dim3(256), 0, s, src, (bf16_t*)plain, (bf16_t*)tr, R, C, Rp, ldp, ldt),
        hipLaunchKernelGGL((pack_weight_dual_kernel<float>), grid, dim3(256), 0, s, src, (float*)plain, (float*)tr, R, C, Rp, ldp, ldt));
    DWN_CHECK_LAUNCH();
    return 0;
}
// per-sample gated weights: dst[b][n][k] = T(w[n][k] * gate[b][k])  — the SE gate folded into conv_pwl's weights so
// that its GEMM streams the activated tensor z3 with a plain loader (dwiseneuro.py:40-43,117-120)
template <typename T>
__global__ __launch_bounds__(256) void gate_weights_kernel(const float* w, const float* gate, T* dst, int N, int K) {
    const int b = blockIdx.y;
    const i64 per = (i64)N * K;
    for (i64 i = (i64)blockIdx.x * 256 + threadIdx.x; i < per; i += (i64)gridDim.x * 256) {
        const int k = (int)(i % K);
        dst[(i64)b * per + i] = from_f<T>(w[i] * gate[(i64)b * K + k]);
    }
}
int k_gate_weights(const float* w, const float* gate, void* dst, int B, int N, int K, int dtype, hipStream_t s) {
    i64 per = (i64)N * K;
    unsigned bx = (unsigned)((per + 255) / 256);
    if (bx > 64) bx = 64;
    dim3 grid(bx, (unsigned)B);
    DISPATCH_T(dtype,
        hipLaunchKernelGGL((gate_weights_kernel<bf16_t>), grid, dim3(256), 0, s, w, gate, (bf16_t*)dst, N, K),
        hipLaunchKernelGGL((gate_weights_kernel<float>), grid, dim3(256), 0, s, w, gate, (float*)dst, N, K));
    DWN_CHECK_LAUNCH();
    return 0;
}
// depth-wise weights: reference layout [C][taps] -> tap-major [taps][C] fp32
__global__ void pack_dw_kernel(const float* src, float* dst, int C, int taps) {
    int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= C * taps) return;
    int k = idx / C, c = idx % C;
    dst[idx] = src[c * taps + k];
}
int k_pack_dw(const float* src, float* dst, int C, int taps, hipStream_t s) {
    hipLaunchKernelGGL(pack_dw_kernel, dim3((C * taps + 255) / 256), dim3(256), 0, s, src, dst, C, taps);
    DWN_CHECK_LAUNCH();
    return 0;
}

// ------------------------------------------------------------------------------------------------
// readout glue (Readout, dwiseneuro.py:283-287) + Poisson loss (losses.py:10-21)
// ------------------------------------------------------------------------------------------------
// dz[m][g*Rp + r] = dout[b][n][t] * (1 - exp(-beta*out[b][n][t])),  n = g*Rg + r  (zero in the padding),
// db[n] += sum_m dz.  One workgroup per (b, 64-neuron tile): LDS transpose (t-contiguous -> n-contiguous).
template <typename T>
__global__ __launch_bounds__(256) void readout_dz_kernel(const float* dout, const float* out, float beta, int Tn,
                                                         int n_valid, int Rg, int Rp, int groups, T* dz, float* db) {
    extern __shared__ float tile[];        // [64][Tn+1]
    const int b = blockIdx.y, n0 = blockIdx.x * 64, tid = threadIdx.x;
    const int npad_total = groups * Rp;
    for (int i = tid; i < 64 * Tn; i += 256) {
        int nl = i / Tn, t = i % Tn;
        int np = n0 + nl;                  // index in the padded [groups][Rp] space
        int g = np / Rp, r = np % Rp;
        int n = g * Rg + r;
        float v = 0.f;
        if (np < npad_total && r < Rg && n < n_valid) {
            i64 off = ((i64)b * n_valid + n) * Tn + t;
            v = dout[off] * (1.0f - __expf(-beta * out[off]));
        }
        tile[nl * (Tn + 1) + t] = v;
    }
    __syncthreads();
    for (int i = tid; i < 64 * Tn; i += 256) {
        int t = i / 64, nl = i % 64;
        int np = n0 + nl;
        if (np < npad_total) dz[((i64)b * Tn + t) * npad_total + np] = from_f<T>(tile[nl * (Tn + 1) + t]);
    }
    DET_ENTER();
    if (tid < 64) {
        int np = n0 + tid;
        int g = np / Rp, r = np % Rp;
        int n = g * Rg + r;
        if (np < npad_total && r < Rg && n < groups * Rg) {
            float sum = 0.f;
            for (int t = 0; t < Tn; ++t) sum += round_t<T>(tile[tid * (Tn + 1) + t]);
            atomicAdd(db + n, sum);
        }
    }
    DET_EXIT();
}
int k_readout_dz(const float* dout, const float* out, float beta, int B, int Tn, int n_valid, int Rg, int Rp,
                 int groups, void* dz, float* db, int dtype, hipStream_t s) {
    dim3 grid((groups * Rp + 63) / 64, B);
    size_t lds = (size_t)64 * (Tn + 1) * sizeof(float);
    DISPATCH_T(dtype,
        hipLaunchKernelGGL((readout_dz_kernel<bf16_t>), grid, dim3(256), lds, s, dout, out, beta, Tn, n_valid, Rg, Rp, groups, (bf16_t*)dz, db),
        hipLaunchKernelGGL((readout_dz_kernel<float>), grid, dim3(256), lds, s, dout, out, beta, Tn, n_valid, Rg, Rp, groups, (float*)dz, db));
    DWN_CHECK_LAUNCH();
    return 0;
}

// loss += sum_{b,n,t} w[b] * (x - y*log(x + eps));  dpred = gscale * w[b] * (1 - y/(x+eps))
// grid = (chunks, B): a workgroup works inside ONE sample, so the weight is a scalar and a sample whose weight is zero (every
// sample of the other mice with one-hot mouse weights, src/datasets.py:185-186) costs no reads at all; 16-byte accesses when the
// sample length allows (VEC = 4).
template <int VEC>
__global__ __launch_bounds__(256) void poisson_fwd_kernel(const float* pred, const float* target, const float* w,
                                                          i64 per_sample, float eps, double* loss) {
    const float wb = w[blockIdx.y];
#ifndef DWN_DETERMINISTIC
    if (wb == 0.f) return;                               // (uniform; the ordered build must still pass its ticket below)
#endif
    const float* x = pred + (i64)blockIdx.y * per_sample;
    const float* y = target + (i64)blockIdx.y * per_sample;
    float acc = 0.f;
    const i64 stride = (i64)gridDim.x * 256 * VEC;
    i64 i = ((i64)blockIdx.x * 256 + threadIdx.x) * VEC;
    if constexpr (VEC == 4) {
        // four 16-byte pairs in flight per thread (the grid is small: one fp64 atomic per workgroup on ONE address is what this
        // kernel used to spend its time on — 4096 of them, 54 us; now at most 512)
        for (; wb != 0.f && i + 3 * stride < per_sample; i += 4 * stride) {
            float4 xv[4], yv[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) { xv[u] = *reinterpret_cast<const float4*>(x + i + u * stride); yv[u] = *reinterpret_cast<const float4*>(y + i + u * stride); }
#pragma unroll
            for (int u = 0; u < 4; ++u)
                acc += (xv[u].x - yv[u].x * __logf(xv[u].x + eps)) + (xv[u].y - yv[u].y * __logf(xv[u].y + eps)) +
                       (xv[u].z - yv[u].z * __logf(xv[u].z + eps)) + (xv[u].w - yv[u].w * __logf(xv[u].w + eps));
        }
    }
    for (; wb != 0.f && i < per_sample; i += stride) {
        if constexpr (VEC == 4) {
            const float4 xv = *reinterpret_cast<const float4*>(x + i), yv = *reinterpret_cast<const float4*>(y + i);
            acc += (xv.x - yv.x * __logf(xv.x + eps)) + (xv.y - yv.y * __logf(xv.y + eps)) +
                   (xv.z - yv.z * __logf(xv.z + eps)) + (xv.w - yv.w * __logf(xv.w + eps));
        } else {
            acc += x[i] - y[i] * __logf(x[i] + eps);
        }
    }
    double accd = (double)(wb * acc);
    for (int o = 32; o > 0; o >>= 1) accd += __shfl_xor(accd, o);
    __shared__ double part[4];
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = accd;
    __syncthreads();
    DET_ENTER();
    if (threadIdx.x == 0) atomicAdd(loss, part[0] + part[1] + part[2] + part[3]);
    DET_EXIT();
}
template <int VEC>
__global__ __launch_bounds__(256) void poisson_bwd_kernel(const float* pred, const float* target, const float* w,
                                                          const float* gscale, i64 per_sample, float eps, float* dpred) {
    const float wb = w[blockIdx.y];
    const float g = (gscale ? *gscale : 1.0f) * wb;
    const float* x = pred + (i64)blockIdx.y * per_sample;
    const float* y = target + (i64)blockIdx.y * per_sample;
    float* d = dpred + (i64)blockIdx.y * per_sample;
    for (i64 i = ((i64)blockIdx.x * 256 + threadIdx.x) * VEC; i < per_sample; i += (i64)gridDim.x * 256 * VEC) {
        if constexpr (VEC == 4) {
            float4 o = make_float4(0.f, 0.f, 0.f, 0.f);
            if (wb != 0.f) {
                const float4 xv = *reinterpret_cast<const float4*>(x + i), yv = *reinterpret_cast<const float4*>(y + i);
                o = make_float4(g * (1.0f - yv.x / (xv.x + eps)), g * (1.0f - yv.y / (xv.y + eps)),
                                g * (1.0f - yv.z / (xv.z + eps)), g * (1.0f - yv.w / (xv.w + eps)));
            }
            *reinterpret_cast<float4*>(d + i) = o;
        } else {
            d[i] = wb != 0.f ? g * (1.0f - y[i] / (x[i] + eps)) : 0.f;
        }
    }
}
__global__ void f64_to_f32_kernel(const double* src, float* dst, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dst[i] = (float)src[i];
}
// per_sample = elements of one sample, total = B * per_sample
static inline bool poisson_vec4(const float* a, const float* b, const float* c, i64 per_sample) {
    return (per_sample & 3) == 0 && !(((size_t)a | (size_t)b | (size_t)c) & 15);
}
int k_poisson_fwd(const float* pred, const float* target, const float* w, i64 per_sample, i64 total, float eps,
                  double* loss, hipStream_t s) {
    const i64 B = per_sample > 0 ? total / per_sample : 0;
    if (B <= 0) return 0;
    if (B > 65535) return dwn_set_error(-2, "poisson loss: batch is a grid dimension (<= 65535)");
    i64 chunks = (per_sample + 1023) / 1024;
    const i64 cap = (512 + B - 1) / B;                   // every workgroup ends in one fp64 atomic on the same address: keep them few
    if (chunks > cap) chunks = cap;
    dim3 grid((unsigned)chunks, (unsigned)B);
    if (poisson_vec4(pred, target, pred, per_sample)) hipLaunchKernelGGL(poisson_fwd_kernel<4>, grid, dim3(256), 0, s, pred, target, w, per_sample, eps, loss);
    else hipLaunchKernelGGL(poisson_fwd_kernel<1>, grid, dim3(256), 0, s, pred, target, w, per_sample, eps, loss);
    DWN_CHECK_LAUNCH();
    return 0;
}
int k_poisson_bwd(const float* pred, const float* target, const float* w, const float* gscale, i64 per_sample,
                  i64 total, float eps, float* dpred, hipStream_t s) {
    const i64 B = per_sample > 0 ? total / per_sample : 0;
    if (B <= 0) return 0;
    if (B > 65535) return dwn_set_error(-2, "poisson loss: batch is a grid dimension (<= 65535)");
    i64 chunks = (per_sample + 1023) / 1024;
    const i64 cap = (4096 + B - 1) / B;
    if (chunks > cap) chunks = cap;
    dim3 grid((unsigned)chunks, (unsigned)B);
    if (poisson_vec4(pred, target, dpred, per_sample)) hipLaunchKernelGGL(poisson_bwd_kernel<4>, grid, dim3(256), 0, s, pred, target, w, gscale, per_sample, eps, dpred);
    else hipLaunchKernelGGL(poisson_bwd_kernel<1>, grid, dim3(256), 0, s, pred, target, w, gscale, per_sample, eps, dpred);
    DWN_CHECK_LAUNCH();
    return 0;
}
int k_f64_to_f32(const double* src, float* dst, int n, hipStream_t s) {
    hipLaunchKernelGGL(f64_to_f32_kernel, dim3((n + 63) / 64), dim3(64), 0, s, src, dst, n);
    DWN_CHECK_LAUNCH();
    return 0;
}

// ------------------------------------------------------------------------------------------------
// layout conversion at the API boundary: T channels-last [B*T][C] <-> fp32 NCT [B][C][T]  (cortex in/out)
// ------------------------------------------------------------------------------------------------

// ------------------------------------------------------------------------------------------------
// fused multi-tensor AdamW (+ EMA of the parameters) — torch.optim.AdamW semantics
// (true_batch_001.py:45-48) and ModelEma.update (ema.py:47-55) in one pass over the parameters.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void adamw_ema_kernel(const TensorListEntry* list, int ntensors, float decay_w,
                                                        float omb1, float beta2, float omb2, float eps,
                                                        float step_size, float bc2_sqrt, float ema_decay,
                                                        float ema_omd, float grad_scale) {
    // blockIdx.y = tensor, blockIdx.x strides over its elements.  Scalars are computed on the host in
    // double exactly as torch does: decay_w = 1 - lr*wd, omb = 1 - beta, step_size = lr / (1 - beta1^t).
    const TensorListEntry e = list[blockIdx.y];
    float* p = e.param; const float* g = e.grad; float* m = e.exp_avg; float* v = e.exp_avg_sq; float* ema = e.ema;
    for (i64 i = (i64)blockIdx.x * 256 + threadIdx.x; i < e.numel; i += (i64)gridDim.x * 256) {
        float gi = g[i] * grad_scale;
        float pi = p[i] * decay_w;
        float mi = m[i] + omb1 * (gi - m[i]);                    // exp_avg.lerp_(grad, 1 - beta1)
        float vi = beta2 * v[i] + omb2 * gi * gi;                // mul_(beta2).addcmul_(g, g, 1 - beta2)
        float denom = sqrtf(vi) / bc2_sqrt + eps;
        pi = pi - step_size * (mi / denom);
        p[i] = pi; m[i] = mi; v[i] = vi;
        if (ema) ema[i] = ema_decay * ema[i] + ema_omd * pi;
    }
}
// EMA of float buffers (running stats) and int64 counters (num_batches_tracked: float result truncated, ema.py:52)
__global__ __launch_bounds__(256) void ema_lerp_kernel(const TensorListEntry* list, int ntensors, float decay, float omd) {
    const TensorListEntry e = list[blockIdx.y];
    if (e.is_int64) {
        if (blockIdx.x == 0 && threadIdx.x == 0) {
            long long* d = reinterpret_cast<long long*>(e.ema);
            const long long* sp = reinterpret_cast<const long long*>(e.param);
            for (i64 i = 0; i < e.numel; ++i) d[i] = (long long)(decay * (float)d[i] + omd * (float)sp[i]);
        }
        return;
    }
    for (i64 i = (i64)blockIdx.x * 256 + threadIdx.x; i < e.numel; i += (i64)gridDim.x * 256)
        e.ema[i] = decay * e.ema[i] + omd * e.param[i];
}
int k_adamw_ema(const TensorListEntry* list, int ntensors, int max_blocks, float decay_w, float omb1, float beta2,
                float omb2, float eps, float step_size, float bc2_sqrt, float ema_decay, float ema_omd,
                float grad_scale, hipStream_t s) {
    if (ntensors <= 0) return 0;
    hipLaunchKernelGGL(adamw_ema_kernel, dim3(max_blocks, ntensors), dim3(256), 0, s, list, ntensors, decay_w, omb1,
                       beta2, omb2, eps, step_size, bc2_sqrt, ema_decay, ema_omd, grad_scale);
    DWN_CHECK_LAUNCH();
    return 0;
}
int k_ema_lerp(const TensorListEntry* list, int ntensors, int max_blocks, float decay, float omd, hipStream_t s) {
    if (ntensors <= 0) return 0;
    hipLaunchKernelGGL(ema_lerp_kernel, dim3(max_blocks, ntensors), dim3(256), 0, s, list, ntensors, decay, omd);
    DWN_CHECK_LAUNCH();
    return 0;
}

// ------------------------------------------------------------------------------------------------
// small utilities
// ------------------------------------------------------------------------------------------------
// zero-fill as a kernel (not hipMemsetAsync): always replayed by a captured hipGraph, 16 bytes per lane
__global__ __launch_bounds__(256) void zero_bytes_kernel(uint4* p, size_t n16) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (size_t)gridDim.x * 256) p[i] = make_uint4(0, 0, 0, 0);
}
int k_zero(void* p, size_t nbytes, hipStream_t s) {
    if (nbytes == 0) return 0;
    if (((size_t)p & 15) || (nbytes & 15)) return dwn_set_error(-2, "k_zero: pointer and size must be 16-byte aligned");
    size_t n16 = nbytes / 16;
    size_t blocks = (n16 + 255) / 256;
    if (blocks > 1024) blocks = 1024;
    hipLaunchKernelGGL(zero_bytes_kernel, dim3((unsigned)blocks), dim3(256), 0, s, (uint4*)p, n16);
    DWN_CHECK_LAUNCH();
    return 0;
}

__global__ void fill_f32_kernel(float* p, float v, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = v;
}
int k_fill_f32(float* p, float v, int n, hipStream_t s) {
    hipLaunchKernelGGL(fill_f32_kernel, dim3((n + 255) / 256), dim3(256), 0, s, p, v, n);
    DWN_CHECK_LAUNCH();
    return 0;
}

// ---- one launch for the per-call preparation work of a composite (zero the statistics arena, constant vectors,
// weight packs): these were 5-6 separate 3-10 us launches in front of every block forward / backward.
template <typename T>
__global__ __launch_bounds__(256) void prep_kernel(const PrepArgs pa) {
    int blk = blockIdx.x, o = 0;
    while (o < pa.nops - 1 && blk >= pa.op[o].nblocks) { blk -= pa.op[o].nblocks; ++o; }
    const PrepOp& q = pa.op[o];
    const i64 idx = (i64)blk * 256 + threadIdx.x;
    if (q.kind == PREP_ZERO) {
        uint4* p = reinterpret_cast<uint4*>(q.dst);
        for (i64 i = idx; i < q.n; i += (i64)q.nblocks * 256) p[i] = make_uint4(0, 0, 0, 0);
    } else if (q.kind == PREP_FILL) {
        if (idx < q.n) reinterpret_cast<float*>(q.dst)[idx] = q.v;
    } else if (q.kind == PREP_BNEVAL) {            // bn_finalize_eval_kernel, folded into this launch
        if (idx < q.n) {
            const int c = (int)idx, C = q.C;
            const float invstd = 1.0f / sqrtf(q.src4[c] + q.v);
            const float scale = q.src[c] * invstd;
            float* coef = reinterpret_cast<float*>(q.dst);
            coef[c] = scale;
            coef[C + c] = q.src2[c] - q.src3[c] * scale;
            coef[2 * C + c] = q.src3[c];
            coef[3 * C + c] = invstd;
        }
    } else if (q.kind == PREP_PACKDW) {            // [C][taps] -> [taps][C] fp32
        if (idx < q.n) {
            const int C = q.C, k = (int)(idx / C), c = (int)(idx % C);
            reinterpret_cast<float*>(q.dst)[idx] = q.src[(i64)c * q.R + k];
        }
    } else {                                        // PREP_PACKW: see pack_weight_kernel
        if (idx < q.n) {
            const i64 per = (i64)q.Rd * q.Cd;
            const int g = (int)(idx / per);
            const i64 rem = idx % per;
            const int rd = (int)(rem / q.Cd), cd = (int)(rem % q.Cd);
            float v = 0.f;
            if (!q.transpose) { if (rd < q.R && cd < q.C) v = q.src[((i64)g * q.R + rd) * q.C + cd]; }
            else { if (cd < q.R && rd < q.C) v = q.src[((i64)g * q.R + cd) * q.C + rd]; }
            reinterpret_cast<T*>(q.dst)[idx] = from_f<T>(v);
        }
    }
}
int k_prep(const PrepArgs& pa, int dtype, hipStream_t s) {
    int total = 0;
    for (int i = 0; i < pa.nops; ++i) total += pa.op[i].nblocks;
    if (total == 0) return 0;
    DISPATCH_T(dtype,
        hipLaunchKernelGGL((prep_kernel<bf16_t>), dim3(total), dim3(256), 0, s, pa),
        hipLaunchKernelGGL((prep_kernel<float>), dim3(total), dim3(256), 0, s, pa));
    DWN_CHECK_LAUNCH();
    return 0;
}



// ------------------------------------------------------------------------------------------------
// conv_pw data-gradient without re-reading y1.  The BatchNorm-backward affine dy1 = A1*dh1 + A2*y1 + A3 is linear and
// y1 = a0 . W1^T, so the A2*y1 term folds into a Cin x Cin matrix (reference math: backward of dwiseneuro.py:90-93):
//   da0 = dh1 . (diag(A1) W1) + a0 . G + r3,    G = W1^T diag(A2) W1,    r3 = A3 . W1
// The GEMM then reads dh1 (and the small a0) only: half the HBM traffic of reading (dh1, y1).
// Bp[n][k] (T, ld = E + C): k < E: A1[k]*W1[k][n];  k = E + c': G[c'][n].   W1 is used as rounded to T (forward's values).
// ------------------------------------------------------------------------------------------------
// One launch: blocks [0, nscale) write the diag(A1) W1 part of Bp; the others accumulate 64-row chunks of E into
// gacc[C][C] / r3[C] with fp32 atomics (zeroed by the caller's prep launch); a second launch converts G into Bp.  Was five dependent tiny launches on the dws_bwd -> pw_dgrad critical path.
template <typename T>
__global__ __launch_bounds__(256) void pw_bwd_prep_kernel(const float* w1, const float* abc, int E, int C, T* bp, float* gacc,
                                                          float* r3, int nscale, int gx, int gy, int gr) {
    const i64 ld = (i64)E + C;
    int bid = blockIdx.x;
    __shared__ float sA[64][64 + 4], sB[64][64 + 4];
    if (bid < nscale) {
        // Bp[n][k] = A1[k] * W1[k][n]: a 64 x 64 tile transposed through LDS (reads contiguous along n, writes contiguous along
        // k; the first version wrote 2-byte elements at a stride of (E + C) elements)
        const int nkt = (E + 63) / 64;
        const int k0 = (bid % nkt) * 64, n0 = (bid / nkt) * 64;
        for (int i = threadIdx.x; i < 64 * 64; i += 256) {
            const int kl = i >> 6, nl = i & 63;
            const int k = k0 + kl, n = n0 + nl;
            sA[kl][nl] = (k < E && n < C) ? abc[k] * round_t<T>(w1[(i64)k * C + n]) : 0.f;
        }
        __syncthreads();
        for (int i = threadIdx.x; i < 64 * 64; i += 256) {
            const int nl = i >> 6, kl = i & 63;
            const int k = k0 + kl, n = n0 + nl;
            if (k < E && n < C) bp[n * ld + k] = from_f<T>(sA[kl][nl]);
        }
        DET_EXIT();
        return;
    }
    bid -= nscale;
    const int GR = gr;                            // rows of E per Gram workgroup (<= 64)
    // Gram part: a 64 x 64 tile of G over a 64-row chunk of E, operands staged through LDS (the first version read both W1
    // columns straight from global memory: 0.9 GB of L1 traffic for 117 MFLOP at C = 256, 63 us)
    // sA[e][c'] = A2[e]*W1[e][c'], sB[e][c] = W1[e][c]
    const int bx = bid % gx, by = (bid / gx) % gy, bz = bid / (gx * gy);
    const int e0 = bz * GR;
    const float* A2 = abc + E;
    const float* A3 = abc + 2 * E;
    for (int i = threadIdx.x; i < GR * 64; i += 256) {
        const int el = i >> 6, cl = i & 63;
        const int e = e0 + el;
        const int cpg = bx * 64 + cl, cg = by * 64 + cl;
        const bool eok = e < E;
        sA[el][cl] = (eok && cpg < C) ? A2[e] * round_t<T>(w1[(i64)e * C + cpg]) : 0.f;
        sB[el][cl] = (eok && cg < C) ? round_t<T>(w1[(i64)e * C + cg]) : 0.f;
    }
    __syncthreads();
    const int ty = threadIdx.x >> 4, tx = threadIdx.x & 15;
    float acc[4][4], acc3[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        acc3[i] = 0.f;
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = 0.f;
    }
    const bool do_r3 = bx == 0 && ty == 0;
#pragma unroll 8
    for (int el = 0; el < GR; ++el) {             // ascending e with fmaf
        const float4 a4 = *reinterpret_cast<const float4*>(&sA[el][ty * 4]);
        const float4 b4 = *reinterpret_cast<const float4*>(&sB[el][tx * 4]);
        const float a[4] = {a4.x, a4.y, a4.z, a4.w}, b[4] = {b4.x, b4.y, b4.z, b4.w};
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = fmaf(a[i], b[j], acc[i][j]);
        if (do_r3) {
            const float a3 = e0 + el < E ? A3[e0 + el] : 0.f;
#pragma unroll
            for (int j = 0; j < 4; ++j) acc3[j] = fmaf(a3, b[j], acc3[j]);
        }
    }
    DET_ENTER();
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int cp = bx * 64 + ty * 4 + i;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int c = by * 64 + tx * 4 + j;
            if (cp < C && c < C) atomicAdd(gacc + (i64)cp * C + c, acc[i][j]);
        }
    }
    if (do_r3) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int c = by * 64 + tx * 4 + j;
            if (c < C) atomicAdd(r3 + c, acc3[j]);
        }
    }
    DET_EXIT();
}
// (a "last block converts G" tail instead of this second launch was measured 4x slower than the whole chain it replaced:
// an agent-scope release fence per block means an L2 write-back on this 8-XCD part)
// res_abc / res_C (optional): the stride-1 shortcut branch's BatchNorm backward, whose x terms are linear in a0 too —
//   da0[m][c] += sum_{c' = c + j*C < res_C} (A2sc[c'] * a0[m][c] + A3sc[c'])  ->  G[c][c] += sum_j A2sc[c'], r3[c] += sum_j A3sc[c']
// (the res_abc[0] * dout term is the GEMM's residual epilogue: dwn.h dwn_gemm_nn_args.res)
template <typename T>
__global__ __launch_bounds__(256) void pw_bwd_gram_store_kernel(const float* gacc, int E, int C, T* bp, const float* res_abc, int res_C,
                                                                float* r3) {
    const i64 idx = (i64)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (i64)C * C) return;
    const int cp = (int)(idx / C), c = (int)(idx % C);
    float g = gacc[idx];
    if (res_abc && cp == c) {
        float d2 = 0.f, d3 = 0.f;
        for (int cc = c; cc < res_C; cc += C) { d2 += res_abc[res_C + cc]; d3 += res_abc[2 * res_C + cc]; }
        g += d2;
        r3[c] += d3;          // one thread per c; the prep kernel's atomics on r3 are complete (previous launch)
    }
    bp[(i64)c * ((i64)E + C) + E + cp] = from_f<T>(g);       // Bp[n = c][E + c'] = G[c'][c]
}
// gacc [C*C] fp32 and r3 [C] fp32 must be zero on entry (the block backward's prep launch clears them)
int k_pw_bwd_prep(const float* w1, const float* abc, int E, int C, void* bp, float* gacc, float* r3, int dtype,
                  const float* res_abc, int res_C, hipStream_t s) {
    const int nscale = ((E + 63) / 64) * ((C + 63) / 64);
    // rows of E per Gram workgroup: 32 with one 64 x 64 tile of G (C = 64: 14 -> 9 us, more workgroups on a latency chain), 64 with
    // more (at C = 256 halving it doubles the atomic adders per element of G: 39 -> 62 us; four chunks per workgroup: 54 us)
    const int gr = C <= 64 ? 32 : 64;
    const int gx = (C + 63) / 64, gy = (C + 63) / 64, gz = (E + gr - 1) / gr;
    const int ngram = gx * gy * gz;
    DISPATCH_T(dtype,
        hipLaunchKernelGGL((pw_bwd_prep_kernel<bf16_t>), dim3(nscale + ngram), dim3(256), 0, s, w1, abc, E, C, (bf16_t*)bp, gacc, r3, nscale, gx, gy, gr),
        hipLaunchKernelGGL((pw_bwd_prep_kernel<float>), dim3(nscale + ngram), dim3(256), 0, s, w1, abc, E, C, (float*)bp, gacc, r3, nscale, gx, gy, gr));
    DWN_CHECK_LAUNCH();
    dim3 g3((unsigned)(((i64)C * C + 255) / 256));
    DISPATCH_T(dtype,
        hipLaunchKernelGGL((pw_bwd_gram_store_kernel<bf16_t>), g3, dim3(256), 0, s, gacc, E, C, (bf16_t*)bp, res_abc, res_C, r3),
        hipLaunchKernelGGL((pw_bwd_gram_store_kernel<float>), g3, dim3(256), 0, s, gacc, E, C, (float*)bp, res_abc, res_C, r3));
    DWN_CHECK_LAUNCH();
    return 0;
}

// ------------------------------------------------------------------------------------------------
// conv_pw weight gradient without reading y1.  dW1 = dy1^T a0 with dy1 = A1*dh1 + A2*y1 + A3 and y1 = a0 . W1^T:
//   dW1 = diag(A1) (dh1^T a0) + diag(A2) W1 (a0^T a0) + A3 (1^T a0)
// The three raw products (T1, Ga, s: rows of tacc) come from one pass over dh1 and the narrow a0 — the fused kernel of the
// 64-channel blocks, or gemm_tn with the concatenating P loader [dh1 | a0 | 1] — and this kernel folds them: one thread per
// dW element, a C-long dot product against Ga's column (coalesced) and W1's row (one row per wave: broadcast loads).
// ------------------------------------------------------------------------------------------------
size_t pw_wgrad_tacc_floats(int E, int C) { return (size_t)(E + C + 8) * C; }
template <typename T>
__global__ __launch_bounds__(256) void pw_wgrad_fold_kernel(const float* __restrict__ tacc, const float* __restrict__ abc,
                                                            const float* __restrict__ w1, int E, int C, float* __restrict__ dw) {
    const i64 idx = (i64)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (i64)E * C) return;
    const int e = (int)(idx / C), c = (int)(idx % C);
    const float* ga = tacc + (i64)E * C + c;
    const float* wrow = w1 + (i64)e * C;
    float g0 = 0.f, g1 = 0.f, g2 = 0.f, g3 = 0.f;
    for (int cp = 0; cp < C; cp += 4) {          // C % 8 == 0
        g0 = fmaf(round_t<T>(wrow[cp]), ga[(i64)cp * C], g0);
        g1 = fmaf(round_t<T>(wrow[cp + 1]), ga[(i64)(cp + 1) * C], g1);
        g2 = fmaf(round_t<T>(wrow[cp + 2]), ga[(i64)(cp + 2) * C], g2);
        g3 = fmaf(round_t<T>(wrow[cp + 3]), ga[(i64)(cp + 3) * C], g3);
    }
    const float g = (g0 + g1) + (g2 + g3);
    dw[idx] = fmaf(abc[e], tacc[idx], fmaf(abc[E + e], g, abc[2 * E + e] * tacc[(i64)(E + C) * C + c]));
}
// C in {64, 128, 256}: a workgroup owns 4 * (256 / C) rows of dW; a thread one column c and 4 rows.  W1's rows are staged
// transposed in LDS ([c'][row]: one broadcast ds_read_b128 per c'), Ga's column element is one coalesced load per c', 16 of
// them in flight — the thread-per-element kernel above re-reads W1's row and Ga's column per element and ran at the L1 rate
// (30 us at E = 1792, C = 256); 16 rows per thread left 7 workgroups at C = 64 walking 64 dependent loads (20 us)
template <typename T, int CC>
__global__ __launch_bounds__(256) void pw_wgrad_fold_tile_kernel(const float* __restrict__ tacc, const float* __restrict__ abc,
                                                                 const float* __restrict__ w1, int E, float* __restrict__ dw) {
    constexpr int G = 256 / CC, ROWS = 4 * G;
    __shared__ __attribute__((aligned(16))) float sw[CC][ROWS];          // sw[c'][row] = round(W1[e0 + row][c'])
    const int tid = threadIdx.x, e0 = blockIdx.x * ROWS;
    for (int i = tid; i < ROWS * CC; i += 256) {
        const int row = i / CC, cp = i % CC;
        sw[cp][row] = e0 + row < E ? round_t<T>(w1[(i64)(e0 + row) * CC + cp]) : 0.f;
    }
    __syncthreads();
    const int c = tid % CC, gq = tid / CC;
    const float* ga = tacc + (i64)E * CC + c;
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 1
    for (int cp0 = 0; cp0 < CC; cp0 += 16) {          // not unrolled: all C loads hoisted at once cost 256 registers + scratch
        float gv[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) gv[u] = ga[(i64)(cp0 + u) * CC];
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            const float4 w4 = *reinterpret_cast<const float4*>(&sw[cp0 + u][gq * 4]);
            acc[0] = fmaf(w4.x, gv[u], acc[0]);
            acc[1] = fmaf(w4.y, gv[u], acc[1]);
            acc[2] = fmaf(w4.z, gv[u], acc[2]);
            acc[3] = fmaf(w4.w, gv[u], acc[3]);
        }
    }
    const float sc = tacc[(i64)(E + CC) * CC + c];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int e = e0 + gq * 4 + r;
        if (e < E) dw[(i64)e * CC + c] = fmaf(abc[e], tacc[(i64)e * CC + c], fmaf(abc[E + e], acc[r], abc[2 * E + e] * sc));
    }
}
int k_pw_wgrad_fold(const float* tacc, const float* abc, const float* w1, int E, int C, float* dw, int dtype, hipStream_t s) {
#define FOLD_TILE(CC_) do { \
        dim3 grid((unsigned)((E + 4 * (256 / CC_) - 1) / (4 * (256 / CC_)))); \
        DISPATCH_T(dtype, \
            hipLaunchKernelGGL((pw_wgrad_fold_tile_kernel<bf16_t, CC_>), grid, dim3(256), 0, s, tacc, abc, w1, E, dw), \
            hipLaunchKernelGGL((pw_wgrad_fold_tile_kernel<float, CC_>), grid, dim3(256), 0, s, tacc, abc, w1, E, dw)); \
        DWN_CHECK_LAUNCH(); \
        return 0; } while (0)
    if (C == 64) FOLD_TILE(64);
    if (C == 128) FOLD_TILE(128);
    if (C == 256) FOLD_TILE(256);
#undef FOLD_TILE
    dim3 grid((unsigned)(((i64)E * C + 255) / 256));
    DISPATCH_T(dtype,
        hipLaunchKernelGGL((pw_wgrad_fold_kernel<bf16_t>), grid, dim3(256), 0, s, tacc, abc, w1, E, C, dw),
        hipLaunchKernelGGL((pw_wgrad_fold_kernel<float>), grid, dim3(256), 0, s, tacc, abc, w1, E, C, dw));
    DWN_CHECK_LAUNCH();
    return 0;
}
