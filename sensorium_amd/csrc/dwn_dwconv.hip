// Depth-wise convolution kernels of InvertedResidual3d (reference: src/models/dwiseneuro.py:96-111):
//   spatial  (1,k,k) stride (1,s,s) pad k/2   and   temporal (k,1,1) pad k/2, groups = channels.
//
// HBM-bound: every kernel reads its input tensor once and writes its output once, in
// 128-byte channel segments of channels-last rows.  The producer's BatchNorm+SiLU is applied while
// loading (LD_BNACT), the BN Σ/Σ² of the output are accumulated in the same pass, and the backward
// kernels fuse data-gradient, weight-gradient, SiLU' and the BN-backward sums.
//
// Work decomposition: a workgroup owns one channel slice of 8 vectors (8*KC channels: 64 bf16 / 32
// fp32 = 128 B per pixel) so a thread's channels never change -> per-channel weights, weight-grad and
// statistics accumulators live in registers for the whole (persistent, grid-stride) kernel.
#include "dwn_internal.h"

#define NCV 8   // 16-byte vectors per pixel per channel slice

extern __shared__ __attribute__((aligned(16))) unsigned char dyn_smem[];

template <int KC>
__device__ __forceinline__ void block_stats_flush(float* lstat, const float* s0, const float* s1, int cv,
                                                 int c0, int C, double* stats, int rep) {
    // lstat: [2][NCV*KC] zeroed by caller + barrier
#pragma unroll
    for (int i = 0; i < KC; ++i) {
        atomicAdd(&lstat[cv * KC + i], s0[i]);
        atomicAdd(&lstat[NCV * KC + cv * KC + i], s1[i]);
    }
    __syncthreads();
    const int tid = threadIdx.x;
    if (tid < 2 * NCV * KC) {
        int which = tid / (NCV * KC), c = c0 + tid % (NCV * KC);
        if (c < C) stat_add(stats, rep, C, which, c, lstat[tid]);
    }
}

// ------------------------------------------------------------------------------------------------
// spatial forward
// ------------------------------------------------------------------------------------------------
template <typename T, int KS>
__global__ __launch_bounds__(256) void dw_spatial_fwd_kernel(const DwSpatialFwd a) {
    constexpr int KC = TT<T>::KC;
    constexpr int P = KS / 2;
    __shared__ float lstat[2 * NCV * KC];
    const int tid = threadIdx.x;
    const int cv = tid % NCV, pl = tid / NCV;
    const int c0 = blockIdx.y * NCV * KC;
    const int chan = c0 + cv * KC;
    const bool chan_ok = chan < a.C;
    if (tid < 2 * NCV * KC) lstat[tid] = 0.f;

    float w[KS * KS][KC];
#pragma unroll
    for (int k = 0; k < KS * KS; ++k) {
        if (chan_ok) ld_coef<KC>(a.w + (i64)k * a.C + chan, w[k]);
        else {
#pragma unroll
            for (int i = 0; i < KC; ++i) w[k][i] = 0.f;
        }
    }
    float st0[KC], st1[KC];
#pragma unroll
    for (int i = 0; i < KC; ++i) { st0[i] = 0.f; st1[i] = 0.f; }

    const int Wp = a.Win + 2 * P;
    const int nbands = (a.Hout + a.rows_band - 1) / a.rows_band;
    const int ntiles = a.planes * nbands;
    T* outp = reinterpret_cast<T*>(a.out);
    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const int plane = tile / nbands, band = tile % nbands;
        const int ho0 = band * a.rows_band;
        const int nro = (a.Hout - ho0 < a.rows_band) ? a.Hout - ho0 : a.rows_band;
        const int hi0 = ho0 * a.stride - P;
        const int rows_in = (nro - 1) * a.stride + KS;
        // stage the activated input rows (zero padded) once
        for (int idx = tid; idx < rows_in * Wp * NCV; idx += 256) {
            int cvv = idx % NCV, pix = idx / NCV;
            int wi = pix % Wp - P, hi = hi0 + pix / Wp;
            uint4 val = make_uint4(0, 0, 0, 0);
            int ch = c0 + cvv * KC;
            if (hi >= 0 && hi < a.Hin && wi >= 0 && wi < a.Win && ch < a.C) {
                float v[KC];
                load_op<LD_BNACT, T>(a.in, ((i64)plane * a.Hin + hi) * a.Win + wi, ch, v);
                val = pack16<T>(v);
            }
            *reinterpret_cast<uint4*>(dyn_smem + (i64)idx * 16) = val;
        }
        __syncthreads();
        for (int item = pl; item < nro * a.Wout; item += 256 / NCV) {
            int oy = item / a.Wout, ox = item % a.Wout;
            float acc[KC];
#pragma unroll
            for (int i = 0; i < KC; ++i) acc[i] = 0.f;
#pragma unroll
            for (int dy = 0; dy < KS; ++dy)
#pragma unroll
                for (int dx = 0; dx < KS; ++dx) {
                    int pix = (oy * a.stride + dy) * Wp + ox * a.stride + dx;
                    float v[KC];
                    unpack16<T>(*reinterpret_cast<const uint4*>(dyn_smem + ((i64)pix * NCV + cv) * 16), v);
#pragma unroll
                    for (int i = 0; i < KC; ++i) acc[i] = fmaf(w[dy * KS + dx][i], v[i], acc[i]);
                }
            if (chan_ok) {
                st_vec<T>(outp + (((i64)plane * a.Hout + ho0 + oy) * a.Wout + ox) * a.C + chan, acc);
#pragma unroll
                for (int i = 0; i < KC; ++i) {
                    float r = round_t<T>(acc[i]);
                    st0[i] += r;
                    st1[i] += r * r;
                }
            }
        }
        __syncthreads();
    }
    if (a.stats) block_stats_flush<KC>(lstat, st0, st1, cv, c0, a.C, a.stats, blockIdx.x % DWN_NREP);
}

// ------------------------------------------------------------------------------------------------
// spatial backward: dh1 = (dwS^T dy2) * silu'(h1), dW, Σdh1, Σdh1·ŷ1
// ------------------------------------------------------------------------------------------------
template <typename T, int KS>
__global__ __launch_bounds__(256) void dw_spatial_bwd_kernel(const DwSpatialBwd a) {
    constexpr int KC = TT<T>::KC;
    constexpr int P = KS / 2;
    constexpr int CS = NCV * KC;
    __shared__ float lstat[2 * CS];
    __shared__ float lw[KS * KS * CS];
    const int tid = threadIdx.x;
    const int cv = tid % NCV, pl = tid / NCV;
    const int c0 = blockIdx.y * CS;
    const int chan = c0 + cv * KC;
    const bool chan_ok = chan < a.C;
    if (tid < 2 * CS) lstat[tid] = 0.f;
    for (int i = tid; i < KS * KS * CS; i += 256) {
        int k = i / CS, c = c0 + i % CS;
        lw[i] = c < a.C ? a.w[(i64)k * a.C + c] : 0.f;
    }
    float bs[KC], bt[KC], bm[KC], bi[KC];
#pragma unroll
    for (int i = 0; i < KC; ++i) { bs[i] = 0.f; bt[i] = 0.f; bm[i] = 0.f; bi[i] = 0.f; }
    if (chan_ok) {
        ld_coef<KC>(a.y1.v1 + chan, bs); ld_coef<KC>(a.y1.v2 + chan, bt);
        ld_coef<KC>(a.y1.v3 + chan, bm); ld_coef<KC>(a.y1.v4 + chan, bi);
    }
    float dwacc[KS * KS][KC];
#pragma unroll
    for (int k = 0; k < KS * KS; ++k)
#pragma unroll
        for (int i = 0; i < KC; ++i) dwacc[k][i] = 0.f;
    float st0[KC], st1[KC];
#pragma unroll
    for (int i = 0; i < KC; ++i) { st0[i] = 0.f; st1[i] = 0.f; }
    __syncthreads();

    const int Wq = a.Wout + 2;                      // staged columns wo = -1 .. Wout
    const int nbands = (a.Hin + a.rows_band - 1) / a.rows_band;
    const int ntiles = a.planes * nbands;
    const int s = a.stride;
    T* dhp = reinterpret_cast<T*>(a.dh1);
    const T* y1p = reinterpret_cast<const T*>(a.y1.p);
    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const int plane = tile / nbands, band = tile % nbands;
        const int hi0 = band * a.rows_band;
        const int nri = (a.Hin - hi0 < a.rows_band) ? a.Hin - hi0 : a.rows_band;
        // output rows that touch input rows [hi0, hi0+nri): ho = (hi + P - dy) / s
        int lo_num = hi0 + P - (KS - 1);
        const int ho_lo = lo_num >= 0 ? lo_num / s : -((-lo_num + s - 1) / s);
        const int ho_hi = (hi0 + nri - 1 + P) / s;
        const int rows_q = ho_hi - ho_lo + 1;
        for (int idx = tid; idx < rows_q * Wq * NCV; idx += 256) {
            int cvv = idx % NCV, pix = idx / NCV;
            int wo = pix % Wq - 1, ho = ho_lo + pix / Wq;
            uint4 val = make_uint4(0, 0, 0, 0);
            int ch = c0 + cvv * KC;
            if (ho >= 0 && ho < a.Hout && wo >= 0 && wo < a.Wout && ch < a.C) {
                float v[KC];
                load_op<LD_AFFINE2, T>(a.dy, ((i64)plane * a.Hout + ho) * a.Wout + wo, ch, v);
                val = pack16<T>(v);
            }
            *reinterpret_cast<uint4*>(dyn_smem + (i64)idx * 16) = val;
        }
        __syncthreads();
        for (int item = pl; item < nri * a.Win; item += 256 / NCV) {
            int iy = item / a.Win, wi = item % a.Win;
            int hi = hi0 + iy;
            if (!chan_ok) continue;
            const i64 row = ((i64)plane * a.Hin + hi) * a.Win + wi;
            float y[KC];
            ld_vec<T>(y1p + row * a.y1.ld + chan, y);
            float z1[KC], dsl[KC];
#pragma unroll
            for (int i = 0; i < KC; ++i) {
                float h = fmaf(y[i], bs[i], bt[i]);
                float sg = sigmoidf_(h);
                z1[i] = h * sg;
                dsl[i] = sg * (1.0f + h * (1.0f - sg));
            }
            float dz[KC];
#pragma unroll
            for (int i = 0; i < KC; ++i) dz[i] = 0.f;
#pragma unroll
            for (int dy = 0; dy < KS; ++dy) {
                int nh = hi + P - dy;
                if (nh < 0 || nh % s != 0) continue;
                int ho = nh / s;
                if (ho > ho_hi) continue;
#pragma unroll
                for (int dx = 0; dx < KS; ++dx) {
                    int nw = wi + P - dx;
                    if (nw < 0 || nw % s != 0) continue;
                    int wo = nw / s;
                    if (wo > a.Wout) continue;
                    int pix = (ho - ho_lo) * Wq + wo + 1;
                    float g[KC], wv[KC];
                    unpack16<T>(*reinterpret_cast<const uint4*>(dyn_smem + ((i64)pix * NCV + cv) * 16), g);
                    ld_coef<KC>(&lw[(dy * KS + dx) * CS + cv * KC], wv);
#pragma unroll
                    for (int i = 0; i < KC; ++i) {
                        dz[i] = fmaf(wv[i], g[i], dz[i]);
                        dwacc[dy * KS + dx][i] = fmaf(z1[i], g[i], dwacc[dy * KS + dx][i]);
                    }
                }
            }
            float dh[KC];
#pragma unroll
            for (int i = 0; i < KC; ++i) dh[i] = dz[i] * dsl[i];
            st_vec<T>(dhp + row * a.C + chan, dh);
#pragma unroll
            for (int i = 0; i < KC; ++i) {
                float r = round_t<T>(dh[i]);
                st0[i] += r;
                st1[i] += r * (y[i] - bm[i]) * bi[i];
            }
        }
        __syncthreads();
    }
    // weight gradient: reduce over the threads sharing a channel vector through LDS, then global fp32 atomics
    __syncthreads();
    for (int i = tid; i < KS * KS * CS; i += 256) lw[i] = 0.f;
    __syncthreads();
#pragma unroll
    for (int k = 0; k < KS * KS; ++k)
#pragma unroll
        for (int i = 0; i < KC; ++i) atomicAdd(&lw[k * CS + cv * KC + i], dwacc[k][i]);
    __syncthreads();
    for (int i = tid; i < KS * KS * CS; i += 256) {
        int k = i / CS, c = c0 + i % CS;
        if (c < a.C) atomicAdd(a.dw + (i64)c * (KS * KS) + k, lw[i]);
    }
    if (a.stats) block_stats_flush<KC>(lstat, st0, st1, cv, c0, a.C, a.stats, blockIdx.x % DWN_NREP);
}

// ------------------------------------------------------------------------------------------------
// temporal forward: y3[t] = sum_k w[k] z2[t + k - P]
// ------------------------------------------------------------------------------------------------
template <typename T, int KT>
__global__ __launch_bounds__(256) void dw_temporal_fwd_kernel(const DwTemporalFwd a) {
    constexpr int KC = TT<T>::KC;
    constexpr int P = KT / 2;
    __shared__ float lstat[2 * NCV * KC];
    const int tid = threadIdx.x;
    const int cv = tid % NCV, pl = tid / NCV;
    const int c0 = blockIdx.y * NCV * KC;
    const int chan = c0 + cv * KC;
    const bool chan_ok = chan < a.C;
    if (tid < 2 * NCV * KC) lstat[tid] = 0.f;
    __syncthreads();
    float w[KT][KC];
#pragma unroll
    for (int k = 0; k < KT; ++k) {
#pragma unroll
        for (int i = 0; i < KC; ++i) w[k][i] = 0.f;
        if (chan_ok) ld_coef<KC>(a.w + (i64)k * a.C + chan, w[k]);
    }
    float st0[KC], st1[KC];
#pragma unroll
    for (int i = 0; i < KC; ++i) { st0[i] = 0.f; st1[i] = 0.f; }

    const i64 npos = (i64)a.B * a.HW;
    T* outp = reinterpret_cast<T*>(a.out);
    if (chan_ok) {
        for (i64 pos = (i64)blockIdx.x * (256 / NCV) + pl; pos < npos; pos += (i64)gridDim.x * (256 / NCV)) {
            const i64 b = pos / a.HW, hw = pos % a.HW;
            const i64 row0 = b * a.T * a.HW + hw;           // row(t) = row0 + t*HW
            float win[KT][KC];
#pragma unroll
            for (int k = 0; k < KT - 1; ++k) {
                int t = k - P;
                if (t >= 0 && t < a.T) load_op<LD_BNACT, T>(a.in, row0 + (i64)t * a.HW, chan, win[k]);
                else {
#pragma unroll
                    for (int i = 0; i < KC; ++i) win[k][i] = 0.f;
                }
            }
#pragma unroll 4
            for (int t = 0; t < a.T; ++t) {
                if (t + P < a.T) load_op<LD_BNACT, T>(a.in, row0 + (i64)(t + P) * a.HW, chan, win[KT - 1]);
                else {
#pragma unroll
                    for (int i = 0; i < KC; ++i) win[KT - 1][i] = 0.f;
                }
                float acc[KC];
#pragma unroll
                for (int i = 0; i < KC; ++i) acc[i] = 0.f;
#pragma unroll
                for (int k = 0; k < KT; ++k)
#pragma unroll
                    for (int i = 0; i < KC; ++i) acc[i] = fmaf(w[k][i], win[k][i], acc[i]);
                st_vec<T>(outp + (row0 + (i64)t * a.HW) * a.C + chan, acc);
#pragma unroll
                for (int i = 0; i < KC; ++i) {
                    float r = round_t<T>(acc[i]);
                    st0[i] += r;
                    st1[i] += r * r;
                }
#pragma unroll
                for (int k = 0; k < KT - 1; ++k)
#pragma unroll
                    for (int i = 0; i < KC; ++i) win[k][i] = win[k + 1][i];
            }
        }
    }
    if (a.stats) block_stats_flush<KC>(lstat, st0, st1, cv, c0, a.C, a.stats, blockIdx.x % DWN_NREP);
}

// ------------------------------------------------------------------------------------------------
// temporal backward: dz2[t] = sum_k w[k] dy3[t - k + P];  dW[k] = sum z2[t] dy3[t - k + P];
//                    dh2 = dz2 * silu'(h2);  Σdh2, Σdh2·ŷ2
// ------------------------------------------------------------------------------------------------
template <typename T, int KT, int DYK>
__global__ __launch_bounds__(256) void dw_temporal_bwd_kernel(const DwTemporalBwd a) {
    constexpr int KC = TT<T>::KC;
    constexpr int P = KT / 2;
    constexpr int CS = NCV * KC;
    __shared__ float lstat[2 * CS];
    __shared__ float lw[KT * CS];
    const int tid = threadIdx.x;
    const int cv = tid % NCV, pl = tid / NCV;
    const int c0 = blockIdx.y * CS;
    const int chan = c0 + cv * KC;
    const bool chan_ok = chan < a.C;
    if (tid < 2 * CS) lstat[tid] = 0.f;
    for (int i = tid; i < KT * CS; i += 256) lw[i] = 0.f;
    __syncthreads();
    float w[KT][KC], dwacc[KT][KC];
#pragma unroll
    for (int k = 0; k < KT; ++k) {
#pragma unroll
        for (int i = 0; i < KC; ++i) { w[k][i] = 0.f; dwacc[k][i] = 0.f; }
        if (chan_ok) ld_coef<KC>(a.w + (i64)k * a.C + chan, w[k]);
    }
    float bs[KC], bt[KC], bm[KC], bi[KC];
#pragma unroll
    for (int i = 0; i < KC; ++i) { bs[i] = 0.f; bt[i] = 0.f; bm[i] = 0.f; bi[i] = 0.f; }
    if (chan_ok) {
        ld_coef<KC>(a.y2.v1 + chan, bs); ld_coef<KC>(a.y2.v2 + chan, bt);
        ld_coef<KC>(a.y2.v3 + chan, bm); ld_coef<KC>(a.y2.v4 + chan, bi);
    }
    float st0[KC], st1[KC];
#pragma unroll
    for (int i = 0; i < KC; ++i) { st0[i] = 0.f; st1[i] = 0.f; }

    const i64 npos = (i64)a.B * a.HW;
    T* dhp = reinterpret_cast<T*>(a.dh2);
    const T* y2p = reinterpret_cast<const T*>(a.y2.p);
    if (chan_ok) {
        for (i64 pos = (i64)blockIdx.x * (256 / NCV) + pl; pos < npos; pos += (i64)gridDim.x * (256 / NCV)) {
            const i64 b = pos / a.HW, hw = pos % a.HW;
            const i64 row0 = b * a.T * a.HW + hw;
            float win[KT][KC];                           // win[j] = dy3(t + j - P)
#pragma unroll
            for (int k = 0; k < KT - 1; ++k) {
                int t = k - P;
                if (t >= 0 && t < a.T) load_op<DYK, T>(a.dy, row0 + (i64)t * a.HW, chan, win[k]);
                else {
#pragma unroll
                    for (int i = 0; i < KC; ++i) win[k][i] = 0.f;
                }
            }
#pragma unroll 2
            for (int t = 0; t < a.T; ++t) {
                if (t + P < a.T) load_op<DYK, T>(a.dy, row0 + (i64)(t + P) * a.HW, chan, win[KT - 1]);
                else {
#pragma unroll
                    for (int i = 0; i < KC; ++i) win[KT - 1][i] = 0.f;
                }
                const i64 row = row0 + (i64)t * a.HW;
                float y[KC];
                ld_vec<T>(y2p + row * a.y2.ld + chan, y);
                float dh[KC];
#pragma unroll
                for (int i = 0; i < KC; ++i) {
                    float h = fmaf(y[i], bs[i], bt[i]);
                    float sg = sigmoidf_(h);
                    float z2 = h * sg;
                    float dz = 0.f;
#pragma unroll
                    for (int k = 0; k < KT; ++k) {
                        float g = win[KT - 1 - k][i];
                        dz = fmaf(w[k][i], g, dz);
                        dwacc[k][i] = fmaf(z2, g, dwacc[k][i]);
                    }
                    dh[i] = dz * (sg * (1.0f + h * (1.0f - sg)));
                }
                st_vec<T>(dhp + row * a.C + chan, dh);
#pragma unroll
                for (int i = 0; i < KC; ++i) {
                    float r = round_t<T>(dh[i]);
                    st0[i] += r;
                    st1[i] += r * (y[i] - bm[i]) * bi[i];
                }
#pragma unroll
                for (int k = 0; k < KT - 1; ++k)
#pragma unroll
                    for (int i = 0; i < KC; ++i) win[k][i] = win[k + 1][i];
            }
        }
    }
#pragma unroll
    for (int k = 0; k < KT; ++k)
#pragma unroll
        for (int i = 0; i < KC; ++i) atomicAdd(&lw[k * CS + cv * KC + i], dwacc[k][i]);
    __syncthreads();
    for (int i = tid; i < KT * CS; i += 256) {
        int k = i / CS, c = c0 + i % CS;
        if (c < a.C) atomicAdd(a.dw + (i64)c * KT + k, lw[i]);
    }
    if (a.stats) block_stats_flush<KC>(lstat, st0, st1, cv, c0, a.C, a.stats, blockIdx.x % DWN_NREP);
}

// ------------------------------------------------------------------------------------------------
// launchers
// ------------------------------------------------------------------------------------------------
static inline int grid_cap(i64 work, int cap) { return (int)(work < cap ? (work > 0 ? work : 1) : cap); }

template <typename T>
static int spatial_fwd_t(DwSpatialFwd a, hipStream_t s) {
    constexpr int KC = TT<T>::KC;
    if (a.ks != 3) return dwn_set_error(-4, "dw_spatial: only spatial_kernel=3 is built");
    if (a.C % KC) return dwn_set_error(-2, "dw_spatial: C must be a multiple of 8");
    const int Wp = a.Win + 2;
    if (a.rows_band <= 0) {
        int rb = 1;
        while (rb < a.Hout && ((rb) * a.stride + 3) * Wp * NCV * 16 <= 40 * 1024) ++rb;
        a.rows_band = rb;
    }
    const int rows_in = (a.rows_band - 1) * a.stride + 3;
    const size_t lds = (size_t)rows_in * Wp * NCV * 16;
    if (lds > 64 * 1024) return dwn_set_error(-5, "dw_spatial: plane too wide for the LDS tile");
    const int nbands = (a.Hout + a.rows_band - 1) / a.rows_band;
    const int slices = (a.C + NCV * KC - 1) / (NCV * KC);
    dim3 grid(grid_cap((i64)a.planes * nbands, (2048 + slices - 1) / slices), slices);
    hipLaunchKernelGGL((dw_spatial_fwd_kernel<T, 3>), grid, dim3(256), lds, s, a);
    DWN_CHECK_LAUNCH();
    return 0;
}
int launch_dw_spatial_fwd(const DwSpatialFwd& a, int dtype, hipStream_t s) {
    return dtype == DWN_BF16 ? spatial_fwd_t<bf16_t>(a, s) : spatial_fwd_t<float>(a, s);
}

template <typename T>
static int spatial_bwd_t(DwSpatialBwd a, hipStream_t s) {
    constexpr int KC = TT<T>::KC;
    if (a.ks != 3) return dwn_set_error(-4, "dw_spatial: only spatial_kernel=3 is built");
    if (a.C % KC) return dwn_set_error(-2, "dw_spatial: C must be a multiple of 8");
    const int Wq = a.Wout + 2;
    auto rows_q = [&](int rb) { return (rb - 1 + 2) / a.stride + 2; };   // upper bound on staged output rows
    if (a.rows_band <= 0) {
        int rb = 1;
        while (rb < a.Hin && rows_q(rb + 1) * Wq * NCV * 16 <= 40 * 1024) ++rb;
        a.rows_band = rb;
    }
    const size_t lds = (size_t)rows_q(a.rows_band) * Wq * NCV * 16;
    if (lds > 60 * 1024) return dwn_set_error(-5, "dw_spatial_bwd: plane too wide for the LDS tile");
    const int nbands = (a.Hin + a.rows_band - 1) / a.rows_band;
    const int slices = (a.C + NCV * KC - 1) / (NCV * KC);
    dim3 grid(grid_cap((i64)a.planes * nbands, (1024 + slices - 1) / slices), slices);
    hipLaunchKernelGGL((dw_spatial_bwd_kernel<T, 3>), grid, dim3(256), lds, s, a);
    DWN_CHECK_LAUNCH();
    return 0;
}
int launch_dw_spatial_bwd(const DwSpatialBwd& a, int dtype, hipStream_t s) {
    return dtype == DWN_BF16 ? spatial_bwd_t<bf16_t>(a, s) : spatial_bwd_t<float>(a, s);
}

template <typename T>
static int temporal_fwd_t(const DwTemporalFwd& a, hipStream_t s) {
    constexpr int KC = TT<T>::KC;
    if (a.C % KC) return dwn_set_error(-2, "dw_temporal: C must be a multiple of 8");
    const int slices = (a.C + NCV * KC - 1) / (NCV * KC);
    const i64 npos = (i64)a.B * a.HW;
    dim3 grid(grid_cap((npos + 31) / 32, (4096 + slices - 1) / slices), slices);
    if (a.kt == 5) hipLaunchKernelGGL((dw_temporal_fwd_kernel<T, 5>), grid, dim3(256), 0, s, a);
    else if (a.kt == 3) hipLaunchKernelGGL((dw_temporal_fwd_kernel<T, 3>), grid, dim3(256), 0, s, a);
    else return dwn_set_error(-4, "dw_temporal: only temporal_kernel 3 or 5 is built");
    DWN_CHECK_LAUNCH();
    return 0;
}
int launch_dw_temporal_fwd(const DwTemporalFwd& a, int dtype, hipStream_t s) {
    return dtype == DWN_BF16 ? temporal_fwd_t<bf16_t>(a, s) : temporal_fwd_t<float>(a, s);
}

template <typename T>
static int temporal_bwd_t(const DwTemporalBwd& a, hipStream_t s) {
    constexpr int KC = TT<T>::KC;
    if (a.C % KC) return dwn_set_error(-2, "dw_temporal: C must be a multiple of 8");
    const int slices = (a.C + NCV * KC - 1) / (NCV * KC);
    const i64 npos = (i64)a.B * a.HW;
    dim3 grid(grid_cap((npos + 31) / 32, (2048 + slices - 1) / slices), slices);
    const bool dy3 = a.dy_kind == LD_DY3;
    if (!dy3 && a.dy_kind != LD_AFFINE2) return dwn_set_error(-3, "dw_temporal_bwd: unsupported dy loader");
    if (a.kt == 5) {
        if (dy3) hipLaunchKernelGGL((dw_temporal_bwd_kernel<T, 5, LD_DY3>), grid, dim3(256), 0, s, a);
        else hipLaunchKernelGGL((dw_temporal_bwd_kernel<T, 5, LD_AFFINE2>), grid, dim3(256), 0, s, a);
    } else if (a.kt == 3) {
        if (dy3) hipLaunchKernelGGL((dw_temporal_bwd_kernel<T, 3, LD_DY3>), grid, dim3(256), 0, s, a);
        else hipLaunchKernelGGL((dw_temporal_bwd_kernel<T, 3, LD_AFFINE2>), grid, dim3(256), 0, s, a);
    } else return dwn_set_error(-4, "dw_temporal: only temporal_kernel 3 or 5 is built");
    DWN_CHECK_LAUNCH();
    return 0;
}
int launch_dw_temporal_bwd(const DwTemporalBwd& a, int dtype, hipStream_t s) {
    return dtype == DWN_BF16 ? temporal_bwd_t<bf16_t>(a, s) : temporal_bwd_t<float>(a, s);
}
