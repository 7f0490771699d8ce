// MFMA GEMM kernels for the point-wise (1x1x1) convolutions, cortex / readout grouped Conv1d and
// their weight gradients (reference call sites: src/models/dwiseneuro.py:91,118,207,276).
//
//  * gemm_nn : C[M][N] = load(A)[M][K] . B[N][K]^T.  M is the huge (b,t,h,w) dimension, K and N are
//    channel counts (64..4096).  The A operand goes global -> registers -> LDS so that the
//    producer's BN/SiLU/SE-gate/positional-encoding/BN-backward affine is applied in flight
//    (dwn_common.h loaders); the epilogue stages the tile through LDS so every global store is a full
//    contiguous row segment, and folds in the batch-norm Σ/Σ² of the *next* BN.
//  * gemm_tn : dW[R][Cc] += load(P)^T . load(Q), contraction over M, split over workgroups with fp32
//    atomics.  bf16 fragments come from row-major LDS tiles through ds_read_b64_tr_b16.
//
// MFMA shapes: v_mfma_f32_16x16x32_bf16 (bf16 storage) and v_mfma_f32_16x16x4_f32 (fp32 parity
// mode, bit-exact fp32 FMA chain).  Operand maps (cdna_hip_programming.md §3): lane l holds
// A[row l&15][k = 8*(l>>4)+j], B[k = 8*(l>>4)+j][col l&15]; C/D: col = l&15, row = 4*(l>>4)+reg.
#include "dwn_internal.h"

typedef __attribute__((ext_vector_type(8))) short bf16x8_t;
typedef __attribute__((ext_vector_type(4))) short s16x4_t;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;

template <typename T> struct Mma;
template <> struct Mma<bf16_t> {
    static __device__ __forceinline__ void run(const uint4& a, const uint4& b, f32x4_t& acc) {
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, a),
                                                      __builtin_bit_cast(bf16x8_t, b), acc, 0, 0, 0);
    }
};
template <> struct Mma<float> {
    // one 16-byte chunk = 4 k-values per lane group; MFMA e consumes element e of every lane group
    static __device__ __forceinline__ void run(const uint4& a, const uint4& b, f32x4_t& acc) {
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a.x), __uint_as_float(b.x), acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a.y), __uint_as_float(b.y), acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a.z), __uint_as_float(b.z), acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a.w), __uint_as_float(b.w), acc, 0, 0, 0);
    }
};

template <int KIND, typename T>
__device__ __forceinline__ uint4 load_op_packed(const LoadDesc& d, i64 row, int col) {
    if constexpr (KIND == LD_PLAIN) {
        return *reinterpret_cast<const uint4*>(reinterpret_cast<const T*>(d.p) + row * d.ld + col);
    } else {
        float v[TT<T>::KC];
        load_op<KIND, T>(d, row, col, v);
        return pack16<T>(v);
    }
}

// ------------------------------------------------------------------------------------------------
// NN
// ------------------------------------------------------------------------------------------------
template <typename T, int ALD, int EPI, int BN>
__global__ __launch_bounds__(256) void gemm_nn_kernel(const GemmNN g) {
    constexpr int KC = TT<T>::KC;
    constexpr int BM = 128;
    constexpr int ROWB = 128;                          // bytes per tile row per k-step
    constexpr int BK = ROWB / (int)sizeof(T);          // 64 bf16 / 32 f32
    constexpr int NJ = BN / 32;                        // 16-column sub-tiles per wave (2x2 waves)
    constexpr int A_CH = BM * 8 / 256;
    constexpr int B_CH = BN * 8 / 256;
    constexpr int CROW = BN * (int)sizeof(T) + 16;     // epilogue staging row stride (bytes)
    constexpr int SM_AB = (BM + BN) * ROWB;
    constexpr int SM_C = 64 * CROW;
    constexpr int SMEM = SM_AB > SM_C ? SM_AB : SM_C;
    __shared__ __attribute__((aligned(16))) unsigned char smem[SMEM];
    __shared__ float lstat[2 * BN];
    unsigned char* sA = smem;
    unsigned char* sB = smem + BM * ROWB;

    const int tid = threadIdx.x;
    const int ntn = (g.N + BN - 1) / BN;
    const int ntm = (g.M + BM - 1) / BM;
    // XCD-aware order: blocks b and b+8 share an XCD's L2, so the N-tiles of one M-tile (which
    // re-read the same A rows) are placed 8 apart.
    const int bid = blockIdx.x;
    const int xcd = bid & 7;
    const int jj = bid >> 3;
    const int nt = jj % ntn;
    const int mt = (jj / ntn) * 8 + xcd;
    if (mt >= ntm) return;
    const int grp = blockIdx.y;
    const int m0 = mt * BM, n0 = nt * BN;
    const int acol0 = grp * g.K;
    const T* Bp = reinterpret_cast<const T*>(g.b) + (i64)grp * g.N * g.ldb;
    const int ccol0 = grp * g.N;

    if (tid < 2 * BN) lstat[tid] = 0.f;

    uint4 ra[A_CH], rb[B_CH];
    auto load_tiles = [&](int k0) {
#pragma unroll
        for (int i = 0; i < A_CH; ++i) {
            int c = tid + 256 * i;
            int row = c >> 3, kc = c & 7;
            int m = m0 + row, k = k0 + kc * KC;
            if (m < g.M && k < g.K) ra[i] = load_op_packed<ALD, T>(g.a, (i64)m, acol0 + k);
            else ra[i] = make_uint4(0, 0, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < B_CH; ++i) {
            int c = tid + 256 * i;
            int row = c >> 3, kc = c & 7;
            int n = n0 + row, k = k0 + kc * KC;
            if (n < g.N && k < g.K) rb[i] = *reinterpret_cast<const uint4*>(Bp + (i64)n * g.ldb + k);
            else rb[i] = make_uint4(0, 0, 0, 0);
        }
    };
    auto store_tiles = [&]() {
#pragma unroll
        for (int i = 0; i < A_CH; ++i) {
            int c = tid + 256 * i;
            int row = c >> 3, kc = c & 7;
            *reinterpret_cast<uint4*>(sA + row * ROWB + ((kc ^ (row & 7)) << 4)) = ra[i];
        }
#pragma unroll
        for (int i = 0; i < B_CH; ++i) {
            int c = tid + 256 * i;
            int row = c >> 3, kc = c & 7;
            *reinterpret_cast<uint4*>(sB + row * ROWB + ((kc ^ (row & 7)) << 4)) = rb[i];
        }
    };

    f32x4_t acc[4][NJ];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};

    const int wave = tid >> 6, lane = tid & 63;
    const int wm = wave >> 1, wn = wave & 1;
    const int lr = lane & 15, lg = lane >> 4;

    load_tiles(0);
    store_tiles();
    __syncthreads();
    for (int k0 = 0; k0 < g.K; k0 += BK) {
        const bool has_next = (k0 + BK) < g.K;
        if (has_next) load_tiles(k0 + BK);
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
            uint4 af[4], bfr[NJ];
            const int chunk = kb * 4 + lg;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                int row = wm * 64 + i * 16 + lr;
                af[i] = *reinterpret_cast<const uint4*>(sA + row * ROWB + ((chunk ^ (row & 7)) << 4));
            }
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                int row = wn * (BN / 2) + j * 16 + lr;
                bfr[j] = *reinterpret_cast<const uint4*>(sB + row * ROWB + ((chunk ^ (row & 7)) << 4));
            }
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < NJ; ++j) Mma<T>::run(af[i], bfr[j], acc[i][j]);
        }
        __syncthreads();
        if (has_next) {
            store_tiles();
            __syncthreads();
        }
    }

    if constexpr (EPI == EPI_READOUT) {
        // out[b][n][t] = softplus_beta(acc + bias[n]); 4 accumulator regs = 4 consecutive rows m = b*Tn + t
        const float beta = g.sp_beta;
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            int nl = n0 + wn * (BN / 2) + j * 16 + lr;
            int n = ccol0 + nl;
            if (nl >= g.N || n >= g.n_valid) continue;
            float bias = g.bias ? g.bias[n] : 0.f;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                int mb = m0 + wm * 64 + i * 16 + lg * 4;
                float o[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    float z = acc[i][j][r] + bias;
                    float bz = z * beta;
                    o[r] = bz > 20.f ? z : log1pf(__expf(bz)) / beta;
                }
                if ((g.Tn & 3) == 0 && mb + 3 < g.M) {
                    int b = mb / g.Tn, t = mb % g.Tn;
                    *reinterpret_cast<float4*>(g.out_nct + ((i64)b * g.n_valid + n) * g.Tn + t) =
                        make_float4(o[0], o[1], o[2], o[3]);
                } else {
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        int m = mb + r;
                        if (m < g.M) {
                            int b = m / g.Tn, t = m % g.Tn;
                            g.out_nct[((i64)b * g.n_valid + n) * g.Tn + t] = o[r];
                        }
                    }
                }
            }
        }
        return;
    } else {
        // ---- batch-norm statistics of the stored (T-rounded) outputs, from registers
        if (g.stats) {
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                float s = 0.f, ss = 0.f;
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        float v = round_t<T>(acc[i][j][r]);
                        s += v;
                        ss += v * v;
                    }
                s += __shfl_xor(s, 16); ss += __shfl_xor(ss, 16);
                s += __shfl_xor(s, 32); ss += __shfl_xor(ss, 32);
                if (lg == 0) {
                    int col = wn * (BN / 2) + j * 16 + lr;
                    atomicAdd(&lstat[col], s);
                    atomicAdd(&lstat[BN + col], ss);
                }
            }
        }
        if (g.stats) {
            __syncthreads();
            if (tid < 2 * BN) {
                int col = tid % BN, which = tid / BN;
                if (n0 + col < g.N)
                    stat_add(g.stats, (int)(blockIdx.x % DWN_NREP), g.stat_nchan, which, ccol0 + n0 + col, lstat[tid]);
            }
        }
        // ---- stage the tile through LDS (two 64-row halves) so global stores are whole row segments
        constexpr int CPR = BN / KC;
        T* Cp = reinterpret_cast<T*>(g.c);
        [[maybe_unused]] __shared__ float ldg[BN];
        for (int half = 0; half < 2; ++half) {
            if (wm == half) {
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < NJ; ++j)
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            int row = i * 16 + lg * 4 + r;
                            int col = wn * (BN / 2) + j * 16 + lr;
                            *reinterpret_cast<T*>(smem + row * CROW + col * (int)sizeof(T)) = from_f<T>(acc[i][j][r]);
                        }
            }
            __syncthreads();
            const int mh = m0 + half * 64;
            if constexpr (EPI == EPI_STORE) {
                for (int c = tid; c < 64 * CPR; c += 256) {
                    int row = c / CPR, ch = c % CPR;
                    int m = mh + row, n = n0 + ch * KC;
                    if (m < g.M && n < g.N)
                        *reinterpret_cast<uint4*>(Cp + (i64)m * g.ldc + ccol0 + n) =
                            *reinterpret_cast<const uint4*>(smem + row * CROW + ch * 16);
                }
            } else {  // EPI_DG: store + dg[b][n] += sum_rows C[m][n] * silu(s3[n]*y3[m][n] + t3[n])
                const int ch = tid % CPR;              // fixed per thread (256 % CPR == 0)
                const int n = n0 + ch * KC;
                float s3[KC], t3[KC];
                if (n < g.N) { ld_coef<KC>(g.s3 + n, s3); ld_coef<KC>(g.t3 + n, t3); }
                const int mend = (mh + 64 < g.M ? mh + 64 : g.M);
                if (mh < mend) {
                    const int b_first = mh / g.rows_per_sample, b_last = (mend - 1) / g.rows_per_sample;
                    for (int b = b_first; b <= b_last; ++b) {
                        float part[KC];
#pragma unroll
                        for (int i = 0; i < KC; ++i) part[i] = 0.f;
                        if (n < g.N) {
                            for (int row = tid / CPR; row < 64; row += 256 / CPR) {
                                int m = mh + row;
                                if (m >= g.M || m / g.rows_per_sample != b) continue;
                                uint4 raw = *reinterpret_cast<const uint4*>(smem + row * CROW + ch * 16);
                                *reinterpret_cast<uint4*>(Cp + (i64)m * g.ldc + n) = raw;
                                float du[KC], y[KC];
                                unpack16<T>(raw, du);
                                ld_vec<T>(reinterpret_cast<const T*>(g.y3) + (i64)m * g.ldy3 + n, y);
#pragma unroll
                                for (int i = 0; i < KC; ++i) part[i] += du[i] * siluf_(fmaf(y[i], s3[i], t3[i]));
                            }
                        }
                        if (tid < BN) ldg[tid] = 0.f;
                        __syncthreads();
                        if (n < g.N) {
#pragma unroll
                            for (int i = 0; i < KC; ++i) atomicAdd(&ldg[ch * KC + i], part[i]);
                        }
                        __syncthreads();
                        if (tid < BN && n0 + tid < g.N) atomicAdd(g.dg + (i64)b * g.dg_ld + n0 + tid, ldg[tid]);
                        __syncthreads();
                    }
                }
            }
            __syncthreads();
        }
    }
}

template <typename T, int ALD, int EPI>
static int launch_nn_t(const GemmNN& g, hipStream_t s) {
    const int BM = 128;
    const int ntm = (g.M + BM - 1) / BM;
    const int ntm8 = (ntm + 7) / 8 * 8;
    // BN = 64 when N <= 64 (pw-linear into 64 channels); 128 otherwise
    if (g.N <= 64) {
        dim3 grid(ntm8 * ((g.N + 63) / 64), g.groups);
        hipLaunchKernelGGL((gemm_nn_kernel<T, ALD, EPI, 64>), grid, dim3(256), 0, s, g);
    } else {
        dim3 grid(ntm8 * ((g.N + 127) / 128), g.groups);
        hipLaunchKernelGGL((gemm_nn_kernel<T, ALD, EPI, 128>), grid, dim3(256), 0, s, g);
    }
    DWN_CHECK_LAUNCH();
    return 0;
}

template <typename T>
static int launch_nn_d(const GemmNN& g, hipStream_t s) {
    if (g.K % TT<T>::KC != 0) return dwn_set_error(-2, "gemm_nn: K must be a multiple of the 16-byte vector");
    if (g.epi == EPI_READOUT) {
        if (g.a_kind == LD_BNACT) return launch_nn_t<T, LD_BNACT, EPI_READOUT>(g, s);
        if (g.a_kind == LD_PLAIN) return launch_nn_t<T, LD_PLAIN, EPI_READOUT>(g, s);
        return dwn_set_error(-3, "gemm_nn: unsupported loader for readout epilogue");
    }
    if (g.N % TT<T>::KC != 0) return dwn_set_error(-2, "gemm_nn: N must be a multiple of the 16-byte vector");
    if (g.epi == EPI_DG) {
        if (g.a_kind == LD_PLAIN && g.groups == 1) return launch_nn_t<T, LD_PLAIN, EPI_DG>(g, s);
        return dwn_set_error(-3, "gemm_nn: unsupported loader for dg epilogue");
    }
    switch (g.a_kind) {
        case LD_PLAIN: return launch_nn_t<T, LD_PLAIN, EPI_STORE>(g, s);
        case LD_PE: return launch_nn_t<T, LD_PE, EPI_STORE>(g, s);
        case LD_BNACT: return launch_nn_t<T, LD_BNACT, EPI_STORE>(g, s);
        case LD_AFFINE2: return launch_nn_t<T, LD_AFFINE2, EPI_STORE>(g, s);
    }
    return dwn_set_error(-3, "gemm_nn: unsupported loader kind");
}

int launch_gemm_nn(const GemmNN& g, int dtype, hipStream_t s) {
    if (g.M <= 0 || g.N <= 0 || g.K <= 0) return 0;
    return dtype == DWN_BF16 ? launch_nn_d<bf16_t>(g, s) : launch_nn_d<float>(g, s);
}

// ------------------------------------------------------------------------------------------------
// TN (weight gradient): dW[R][Cc] += sum_m P[m][r] Q[m][c]
// ------------------------------------------------------------------------------------------------
template <typename T> struct TnCfg;
template <> struct TnCfg<bf16_t> { static constexpr int PAD = 32; };   // 8 rows x 32 B shift -> conflict-free tr reads
template <> struct TnCfg<float>  { static constexpr int PAD = 64; };   // 16-bank shift between the two rows of a half-wave

template <typename T, int PLD, int QLD>
__global__ __launch_bounds__(256) void gemm_tn_kernel(const GemmTN g) {
    constexpr int KC = TT<T>::KC;
    constexpr int BR = 128, BC = 128, BMK = 32;
    constexpr int RS = BR * (int)sizeof(T) + TnCfg<T>::PAD;      // LDS row stride (bytes), both tiles
    constexpr int CPR = BR / KC;                                  // 16-byte chunks per tile row
    constexpr int NCH = BMK * CPR / 256;                          // chunks per thread per tile (2 bf16 / 4 f32)
    __shared__ __attribute__((aligned(16))) unsigned char smem[2 * BMK * RS];
    unsigned char* sP = smem;
    unsigned char* sQ = smem + BMK * RS;

    const int tid = threadIdx.x;
    const int ntc = (g.Cc + BC - 1) / BC;
    const int rt = blockIdx.x / ntc, ct = blockIdx.x % ntc;
    const int r0 = rt * BR, c0 = ct * BC;
    const int grp = blockIdx.z;
    const int Rl = g.R_load > 0 ? g.R_load : g.R;
    const int pcol0 = grp * Rl, qcol0 = grp * g.Cc;
    const i64 mbeg = (i64)blockIdx.y * g.rows_per_split;
    i64 mend = mbeg + g.rows_per_split;
    if (mend > g.M) mend = g.M;

    const int wave = tid >> 6, lane = tid & 63;
    const int wm = wave >> 1, wn = wave & 1;
    const int lr = lane & 15, lg = lane >> 4;

    f32x4_t acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};

    uint4 rp[NCH], rq[NCH];
    auto load_tiles = [&](i64 mb) {
#pragma unroll
        for (int i = 0; i < NCH; ++i) {
            int c = tid + 256 * i;
            int mrow = c / CPR, ch = c % CPR;
            i64 m = mb + mrow;
            int r = r0 + ch * KC, cc = c0 + ch * KC;
            rp[i] = (m < mend && r < Rl) ? load_op_packed<PLD, T>(g.p, m, pcol0 + r) : make_uint4(0, 0, 0, 0);
            rq[i] = (m < mend && cc < g.Cc) ? load_op_packed<QLD, T>(g.q, m, qcol0 + cc) : make_uint4(0, 0, 0, 0);
        }
    };
    auto store_tiles = [&]() {
#pragma unroll
        for (int i = 0; i < NCH; ++i) {
            int c = tid + 256 * i;
            int mrow = c / CPR, ch = c % CPR;
            *reinterpret_cast<uint4*>(sP + mrow * RS + ch * 16) = rp[i];
            *reinterpret_cast<uint4*>(sQ + mrow * RS + ch * 16) = rq[i];
        }
    };

    if (mbeg < mend) {
        load_tiles(mbeg);
        store_tiles();
    }
    __syncthreads();
    for (i64 mb = mbeg; mb < mend; mb += BMK) {
        const bool has_next = (mb + BMK) < mend;
        if (has_next) load_tiles(mb + BMK);
        if constexpr (TT<T>::IS_BF16) {
            // transposed fragments: lane 4q+p of each 16-lane group addresses row (8*lg + 4h + q), cols 4p..4p+3
            const int q = lr >> 2, p = lr & 3;
            bf16x8_t af[4], bfr[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                int colb = (wm * 64 + i * 16 + 4 * p) * 2;
                auto lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                    (__attribute__((address_space(3))) s16x4_t*)(sP + (8 * lg + q) * RS + colb));
                auto hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                    (__attribute__((address_space(3))) s16x4_t*)(sP + (8 * lg + 4 + q) * RS + colb));
                af[i] = bf16x8_t{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                int colb = (wn * 64 + j * 16 + 4 * p) * 2;
                auto lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                    (__attribute__((address_space(3))) s16x4_t*)(sQ + (8 * lg + q) * RS + colb));
                auto hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                    (__attribute__((address_space(3))) s16x4_t*)(sQ + (8 * lg + 4 + q) * RS + colb));
                bfr[j] = bf16x8_t{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
            }
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i], bfr[j], acc[i][j], 0, 0, 0);
        } else {
#pragma unroll
            for (int ks = 0; ks < BMK / 4; ++ks) {
                float af[4], bfr[4];
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    af[i] = *reinterpret_cast<const float*>(sP + (ks * 4 + lg) * RS + (wm * 64 + i * 16 + lr) * 4);
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    bfr[j] = *reinterpret_cast<const float*>(sQ + (ks * 4 + lg) * RS + (wn * 64 + j * 16 + lr) * 4);
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[i], bfr[j], acc[i][j], 0, 0, 0);
            }
        }
        __syncthreads();
        if (has_next) {
            store_tiles();
            __syncthreads();
        }
    }
    if (mbeg >= mend) return;
    float* dw = g.dw + (i64)grp * g.R * g.lddw;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                int rr = r0 + wm * 64 + i * 16 + lg * 4 + r;
                int cc = c0 + wn * 64 + j * 16 + lr;
                if (rr < g.R && cc < g.Cc) atomicAdd(dw + (i64)rr * g.lddw + cc, acc[i][j][r]);
            }
}

template <typename T, int PLD, int QLD>
static int launch_tn_t(const GemmTN& g, hipStream_t s) {
    dim3 grid(((g.R + 127) / 128) * ((g.Cc + 127) / 128), g.nsplit, g.groups);
    hipLaunchKernelGGL((gemm_tn_kernel<T, PLD, QLD>), grid, dim3(256), 0, s, g);
    DWN_CHECK_LAUNCH();
    return 0;
}

template <typename T>
static int launch_tn_d(const GemmTN& g, hipStream_t s) {
    const int Rl = g.R_load > 0 ? g.R_load : g.R;
    if (Rl % TT<T>::KC != 0 || g.Cc % TT<T>::KC != 0)
        return dwn_set_error(-2, "gemm_tn: R and Cc must be multiples of the 16-byte vector");
    const int pk = g.p_kind, qk = g.q_kind;
    if (pk == LD_AFFINE2 && qk == LD_PE) return launch_tn_t<T, LD_AFFINE2, LD_PE>(g, s);
    if (pk == LD_PLAIN && qk == LD_BNACT) return launch_tn_t<T, LD_PLAIN, LD_BNACT>(g, s);
    if (pk == LD_PLAIN && qk == LD_PLAIN) return launch_tn_t<T, LD_PLAIN, LD_PLAIN>(g, s);
    if (pk == LD_AFFINE2 && qk == LD_PLAIN) return launch_tn_t<T, LD_AFFINE2, LD_PLAIN>(g, s);
    return dwn_set_error(-3, "gemm_tn: unsupported loader combination");
}

int launch_gemm_tn(const GemmTN& g_in, int dtype, hipStream_t s) {
    if (g_in.M <= 0 || g_in.R <= 0 || g_in.Cc <= 0) return 0;
    GemmTN g = g_in;
    if (g.nsplit <= 0) {
        int tiles = ((g.R + 127) / 128) * ((g.Cc + 127) / 128) * g.groups;
        int want = (1024 + tiles - 1) / tiles;
        int maxsplit = (g.M + 255) / 256;
        if (want > maxsplit) want = maxsplit;
        if (want < 1) want = 1;
        int rows = (g.M + want - 1) / want;
        rows = (rows + 31) / 32 * 32;
        g.rows_per_split = rows;
        g.nsplit = (g.M + rows - 1) / rows;
    }
    return dtype == DWN_BF16 ? launch_tn_d<bf16_t>(g, s) : launch_tn_d<float>(g, s);
}
