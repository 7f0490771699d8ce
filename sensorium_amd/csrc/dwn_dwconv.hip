// Depth-wise convolution kernels of InvertedResidual3d (reference: src/models/dwiseneuro.py:96-111):
//   spatial  (1,k,k) stride (1,s,s) pad k/2   and   temporal (k,1,1) pad k/2, groups = channels.
//
// HBM-bound: every kernel reads its input tensor(s) once and writes its output once, in
// 128-byte channel segments of channels-last rows.  The producer's BatchNorm+SiLU is applied while loading,
// the BN Σ/Σ² of the output are accumulated in the same pass, and the backward kernels fuse data-gradient,
// weight-gradient, SiLU' and the BN-backward sums.
//
// Work decomposition: a workgroup owns one 128-byte channel slice (V4<T>::NCV vectors of 4 channels: 64 bf16 /
// 32 fp32 channels); a thread's 4 channels never change, so its BN coefficients, stencil weights,
// weight-gradient and statistics accumulators live in registers for the whole persistent (grid-stride) kernel.
// Loads are issued in batches of 4 independent 8/16-byte vectors per tensor so that, at 3-5 waves per SIMD,
// each CU keeps tens of KB in flight (the HBM latency-bandwidth product).
//
// Kernel families in this file:
//   dw_spatial_fwd_kernel / dw_spatial_bwd_kernel      generic (fp32 parity mode, any stride): packed-fp32 taps
//   dw_spatial_fwd_pair_kernel (stride 1, 2)            bf16: x-pair-packed LDS tile + v_dot2c_f32_bf16 taps
//   dw_spatial_bwd_pair_kernel (stride 1)               bf16: same idea for the data and weight gradients
//   dw_temporal_fwd_kernel / dw_temporal_bwd_kernel     register sliding window along T
// Integer index math is kept off the per-tap path (FastDiv / 24-bit multiplies / per-pixel tile index): on gfx950 an
// integer VALU op costs ~1.5 FMAs and a bf16->fp32 unpack is such an op (profiles/r1c_valu_rates.txt).
#include "dwn_internal.h"
#include <stdlib.h>
#include <type_traits>

extern __shared__ __attribute__((aligned(16))) unsigned char dyn_smem[];

#ifndef DWS_THREADS
#define DWS_THREADS 512      // threads per workgroup of the spatial forward kernel (more waves per LDS tile)
#endif
#define DWS_BWD_THREADS 256
#ifndef DWS_BWD_PAIR_MINW
#define DWS_BWD_PAIR_MINW 3       // waves per SIMD the pair-packed stride-1 backward is compiled for
#endif
#ifndef DWS_BWD_PAIR_XB
#define DWS_BWD_PAIR_XB 2
#endif
#ifndef DWS_BWD_MINB1
#define DWS_BWD_MINB1 3          // resident workgroups per CU the stride-1 spatial backward is compiled for
#endif
#ifndef DWT_RC_MINW
#define DWT_RC_MINW 2        // waves per SIMD the y3-recomputing temporal backward is compiled for
#endif
#ifndef DWT_BWD_TB
#define DWT_BWD_TB 4          // timesteps of loads in flight per thread in the temporal backward (4 or 8)
#endif
// LDS tile budgets of the spatial kernels (bytes); tuned on MI355X with tools/microbench.py
#ifndef DWS_FWD_LDS_BUDGET
#define DWS_FWD_LDS_BUDGET (a.stride >= 2 ? 80 * 1024 : 48 * 1024)
#endif
#ifndef DWS_BWD_LDS_BUDGET
#define DWS_BWD_LDS_BUDGET (44 * 1024)
#endif

template <typename T> struct SL {
    static constexpr int NCV = V4<T>::NCV;       // vectors per slice
    static constexpr int CS = NCV * 4;           // channels per slice
    static constexpr int LP = 256 / NCV;         // pixel lanes per workgroup
    typedef typename V4<T>::raw_t raw_t;
};

template <typename T>
__device__ __forceinline__ void block_stats_flush(float* lstat, const float* s0, const float* s1, int cv, int c0, int C,
                                                 double* stats, int rep) {
    constexpr int CS = SL<T>::CS;
    DET_WAVES_BEGIN
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        atomicAdd(&lstat[cv * 4 + i], s0[i]);
        atomicAdd(&lstat[CS + cv * 4 + i], s1[i]);
    }
    DET_WAVES_END
    __syncthreads();
    DET_ENTER();
    const int tid = threadIdx.x;
    if (tid < 2 * CS) {
        int which = tid / CS, c = c0 + tid % CS;
        if (c < C) stat_add(stats, rep, C, which, c, lstat[tid]);
    }
}

#ifndef DWT_PRIO
#define DWT_PRIO 1         // issue priority while a wave issues its next global loads (0 = off)
#endif
typedef float f2_t __attribute__((ext_vector_type(2)));
// two adjacent channels as one register pair: the compiler emits v_pk_fma_f32 / v_pk_mul_f32 without shuffles
template <typename T> __device__ __forceinline__ void unpack_pairs(const typename V4<T>::raw_t& r, f2_t& lo, f2_t& hi);
template <> __device__ __forceinline__ void unpack_pairs<bf16_t>(const uint2& r, f2_t& lo, f2_t& hi) {
    lo = f2_t{__uint_as_float(r.x << 16), __uint_as_float(r.x & 0xffff0000u)};
    hi = f2_t{__uint_as_float(r.y << 16), __uint_as_float(r.y & 0xffff0000u)};
}
template <> __device__ __forceinline__ void unpack_pairs<float>(const uint4& r, f2_t& lo, f2_t& hi) {
    lo = f2_t{__uint_as_float(r.x), __uint_as_float(r.y)};
    hi = f2_t{__uint_as_float(r.z), __uint_as_float(r.w)};
}

__device__ __forceinline__ void bn_silu4(float* v, const float* s, const float* t) {
#pragma unroll
    for (int i = 0; i < 4; ++i) v[i] = fmaf(v[i], s[i], t[i]);
    silu_n<4>(v);
}

// ------------------------------------------------------------------------------------------------
// spatial forward
// ------------------------------------------------------------------------------------------------
template <typename T, int KS, int ST, int NT>
__global__ __launch_bounds__(NT, 4) void dw_spatial_fwd_kernel(const DwSpatialFwd a) {
    constexpr int NCV = SL<T>::NCV, CS = SL<T>::CS, LP = NT / NCV, P = KS / 2;
    constexpr int XB = 4;                              // x-iterations batched per row (loads in flight per thread)
    typedef typename SL<T>::raw_t raw_t;
    const int stride = ST > 0 ? ST : a.stride;
    __shared__ float lstat[2 * CS];
    const int tid = threadIdx.x;
    const int cv = tid % NCV, pl = tid / NCV;
    const int c0 = blockIdx.y * CS;
    const int chan = c0 + cv * 4;
    const bool chan_ok = chan < a.C;
    const int chs = chan_ok ? chan : 0;               // safe channel for predicated loads
    if (tid < 2 * CS) lstat[tid] = 0.f;
    __syncthreads();

    float w[KS * KS][4], bs[4], bt[4];
#pragma unroll
    for (int k = 0; k < KS * KS; ++k) {
        ldc4(a.w + (i64)k * a.C + chs, w[k]);
        if (!chan_ok) { w[k][0] = w[k][1] = w[k][2] = w[k][3] = 0.f; }
    }
    ldc4(a.in.v1 + chs, bs);
    ldc4(a.in.v2 + chs, bt);
    float st0[4] = {0.f, 0.f, 0.f, 0.f}, st1[4] = {0.f, 0.f, 0.f, 0.f};

    const int Wp = a.Win + 2 * P;
    const FastDiv dvp(Wp), dvo(a.Wout);
    const int nbands = (a.Hout + a.rows_band - 1) / a.rows_band;
    const int ntiles = a.planes * nbands;
    const T* inp = reinterpret_cast<const T*>(a.in.p);
    T* outp = reinterpret_cast<T*>(a.out);
    raw_t* tile = reinterpret_cast<raw_t*>(dyn_smem);
    for (int tile_id = blockIdx.x; tile_id < ntiles; tile_id += gridDim.x) {
        const int plane = tile_id / nbands, band = tile_id % nbands;
        const int ho0 = band * a.rows_band;
        const int nro = (a.Hout - ho0 < a.rows_band) ? a.Hout - ho0 : a.rows_band;
        const int hi0 = ho0 * stride - P;
        const int rows_in = (nro - 1) * stride + KS;
        const i64 plane_row0 = (i64)plane * a.Hin * a.Win;
        // stage the activated input rows (zero padded).  The (row, x) walk is flat and NB loads are issued before
        // the first one is consumed: a row-by-row loop would serialise one HBM round trip per input row.
        const int total_st = rows_in * Wp;                  // flat (row, column) walk: flat index == tile index
        const T* in0 = inp + plane_row0 * a.in.ld;
        if constexpr (TT<T>::IS_BF16) {
            // bf16: stage 8 channels (16 B) per thread — halves the per-vector address/predicate/LDS-store overhead of
            // this VALU-bound pass; the tile keeps the [pixel][64 ch] layout the 4-channel compute lanes read
            constexpr int NB = 8, SV = 8, LPS = NT / SV;          // 8 x 16-B vectors per pixel slice
            const int scv = tid % SV, spl = tid / SV;
            const int sch = c0 + scv * 8;
            const bool sch_ok = sch < a.C;                        // C % 8 == 0: a vector is all-valid or all-invalid
            const int schs = sch_ok ? sch : 0;
            f2_t s8[4], t8[4];
            {
                float sf[8], tf[8];
                ldc4(a.in.v1 + schs, sf); ldc4(a.in.v1 + schs + 4, sf + 4);
                ldc4(a.in.v2 + schs, tf); ldc4(a.in.v2 + schs + 4, tf + 4);
#pragma unroll
                for (int i = 0; i < 4; ++i) { s8[i] = f2_t{sf[2 * i], sf[2 * i + 1]}; t8[i] = f2_t{tf[2 * i], tf[2 * i + 1]}; }
            }
            uint4* tile16 = reinterpret_cast<uint4*>(dyn_smem);
            for (int f0 = spl; f0 < total_st; f0 += NB * LPS) {
                uint4 raw[NB];
                bool okv[NB];
#pragma unroll
                for (int u = 0; u < NB; ++u) {
                    const int f = f0 + u * LPS;
                    const int r = dvp.div(f);
                    const int hi = hi0 + r, wi = dvp.rem(f, r) - P;
                    okv[u] = sch_ok && f < total_st && (unsigned)hi < (unsigned)a.Hin && (unsigned)wi < (unsigned)a.Win;
                    const unsigned off = okv[u] ? __umul24(__mul24(hi, a.Win) + wi, (unsigned)a.in.ld) : 0u;
                    raw[u] = *reinterpret_cast<const uint4*>(in0 + off + schs);
                }
#pragma unroll
                for (int u = 0; u < NB; ++u) {
                    const int f = f0 + u * LPS;
                    if (f < total_st) {
                        const unsigned rw[4] = {raw[u].x, raw[u].y, raw[u].z, raw[u].w};
                        unsigned o[4];
#pragma unroll
                        for (int i = 0; i < 4; ++i) {
                            const f2_t y = f2_t{__uint_as_float(rw[i] << 16), __uint_as_float(rw[i] & 0xffff0000u)};
                            const f2_t h = y * s8[i] + t8[i];
                            const f2_t z = h * sigmoid2f_(h);
                            o[i] = pk_bf16(z.x, z.y);
                        }
                        tile16[f * SV + scv] = okv[u] ? make_uint4(o[0], o[1], o[2], o[3]) : make_uint4(0, 0, 0, 0);
                    }
                }
            }
        } else {
            constexpr int NB = 12;
            for (int f0 = pl; f0 < total_st; f0 += NB * LP) {
                raw_t raw[NB];
                bool okv[NB];
#pragma unroll
                for (int u = 0; u < NB; ++u) {
                    const int f = f0 + u * LP;
                    const int r = dvp.div(f);
                    const int hi = hi0 + r, wi = dvp.rem(f, r) - P;
                    okv[u] = chan_ok && f < total_st && (unsigned)hi < (unsigned)a.Hin && (unsigned)wi < (unsigned)a.Win;
                    const unsigned off = okv[u] ? __umul24(__mul24(hi, a.Win) + wi, (unsigned)a.in.ld) : 0u;
                    raw[u] = ld4_raw<T>(in0 + off + chs);
                }
#pragma unroll
                for (int u = 0; u < NB; ++u) {
                    const int f = f0 + u * LP;
                    if (f < total_st) {
                        float v[4];
                        V4<T>::unpack(raw[u], v);
                        bn_silu4(v, bs, bt);
                        tile[f * NCV + cv] = okv[u] ? V4<T>::pack(v) : V4<T>::zero();
                    }
                }
            }
        }
        __syncthreads();
        {
            const int total = nro * a.Wout;                 // flat walk: narrow planes still fill every pixel lane
            const i64 orow0 = ((i64)plane * a.Hout + ho0) * a.Wout;
            T* out0 = outp + orow0 * a.C + chan;
            const int wdiff = stride * Wp - stride * a.Wout;   // tile index of output i = (i*stride + oy*wdiff) (+ taps)
            for (int i = pl; i < total; i += LP) {
                const int oy = dvo.div(i);
                f2_t acc0 = f2_t{0.f, 0.f}, acc1 = f2_t{0.f, 0.f};
                const raw_t* tp = tile + (i * stride + __mul24(oy, wdiff)) * NCV + cv;
#pragma unroll
                for (int dy = 0; dy < KS; ++dy)
#pragma unroll
                    for (int dx = 0; dx < KS; ++dx) {
                        f2_t v0, v1;
                        unpack_pairs<T>(tp[(dy * Wp + dx) * NCV], v0, v1);
                        acc0 += f2_t{w[dy * KS + dx][0], w[dy * KS + dx][1]} * v0;
                        acc1 += f2_t{w[dy * KS + dx][2], w[dy * KS + dx][3]} * v1;
                    }
                if (chan_ok) {
                    const float acc[4] = {acc0.x, acc0.y, acc1.x, acc1.y};
                    const raw_t packed = V4<T>::pack(acc);
                    *reinterpret_cast<raw_t*>(out0 + __umul24((unsigned)i, (unsigned)a.C)) = packed;
                    float r[4];
                    V4<T>::unpack(packed, r);
#pragma unroll
                    for (int q = 0; q < 4; ++q) { st0[q] += r[q]; st1[q] += r[q] * r[q]; }
                }
            }
        }
        __syncthreads();
    }
    if (a.stats) block_stats_flush<T>(lstat, st0, st1, cv, c0, a.C, a.stats, blockIdx.x % DWN_NREP);
    DET_EXIT();
}

// ------------------------------------------------------------------------------------------------
// spatial forward, bf16 storage, 3x3, stride 1 or 2: x-pair-packed LDS tile + v_dot2c_f32_bf16.
//
// The activated input is staged as dwords holding two horizontally adjacent pixels of one channel,
// tile[row][xp][c] = (z[row][2xp][c], z[row][2xp+1][c])  (xs = wi + 1 is the zero-padded column index).
// A 3-tap row of the stencil is then 2 dot2 instructions on packed operands — dot2(P[j], (w0,w1)) +
// dot2(P[j+1], (w2,0)) for an even output column, dot2(P[j], (0,w0)) + dot2(P[j+1], (w1,w2)) for an odd one —
// instead of 3 bf16->fp32 unpacks + 1.5 packed FMAs per channel, and a thread's 4 channels of a pixel pair are
// one ds_read_b128.  Weights are rounded to bf16 (as the reference's autocast conv does); accumulation is fp32.
// ------------------------------------------------------------------------------------------------
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;
__device__ __forceinline__ float dot2_bf16(unsigned a, unsigned b, float c) {
    return __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2_t, a), __builtin_bit_cast(bf16x2_t, b), c, false);
}
__device__ __forceinline__ unsigned pack_bf16x2(float lo, float hi) { return pk_bf16(lo, hi); }

template <int ST, int NT>
__global__ __launch_bounds__(NT, 4) void dw_spatial_fwd_pair_kernel(const DwSpatialFwd a) {
    typedef bf16_t T;
    constexpr int NCV = 16, CS = 64, LP = NT / NCV, P = 1;
    constexpr int NWC = ST == 1 ? 4 : 2;                 // packed weight combinations per stencil row
    __shared__ float lstat[2 * CS];
    const int tid = threadIdx.x;
    const int cv = tid % NCV, pl = tid / NCV;
    const int c0 = blockIdx.y * CS;
    const int chan = c0 + cv * 4;
    const bool chan_ok = chan < a.C;
    const int chs = chan_ok ? chan : 0;
    if (tid < 2 * CS) lstat[tid] = 0.f;
    __syncthreads();

    // packed weights per stencil row dy and channel: A = (w0,w1), B = (w2,0) [, C = (0,w0), D = (w1,w2)]
    unsigned wp[3][NWC][4];
    {
        float w[9][4];
#pragma unroll
        for (int k = 0; k < 9; ++k) {
            ldc4(a.w + (i64)k * a.C + chs, w[k]);
            if (!chan_ok) { w[k][0] = w[k][1] = w[k][2] = w[k][3] = 0.f; }
        }
#pragma unroll
        for (int dy = 0; dy < 3; ++dy)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                wp[dy][0][q] = pack_bf16x2(w[dy * 3 + 0][q], w[dy * 3 + 1][q]);
                wp[dy][1][q] = pack_bf16x2(w[dy * 3 + 2][q], 0.f);
                if constexpr (ST == 1) {
                    wp[dy][2][q] = pack_bf16x2(0.f, w[dy * 3 + 0][q]);
                    wp[dy][3][q] = pack_bf16x2(w[dy * 3 + 1][q], w[dy * 3 + 2][q]);
                }
            }
    }
    f2_t st0[2] = {f2_t{0.f, 0.f}, f2_t{0.f, 0.f}}, st1[2] = {f2_t{0.f, 0.f}, f2_t{0.f, 0.f}};

    const int Wp = a.Win + 2 * P;
    const int Wpp = (Wp + 1) >> 1;                       // staged pixel pairs per row
    const int Wop = (a.Wout + 1) >> 1;                   // output pixel pairs per row
    const FastDiv dvpp(Wpp), dvop(Wop);
    const int nbands = (a.Hout + a.rows_band - 1) / a.rows_band;
    const int ntiles = a.planes * nbands;
    const T* inp = reinterpret_cast<const T*>(a.in.p);
    T* outp = reinterpret_cast<T*>(a.out);
    unsigned* tile = reinterpret_cast<unsigned*>(dyn_smem);          // [rows_in][Wpp][64] dwords
    // staging role: 8 channels x one pixel pair
    constexpr int SV = 8, LPS = NT / SV, NB = 4;
    const int scv = tid % SV, spl = tid / SV;
    const int sch = c0 + scv * 8;
    const bool sch_ok = sch < a.C;
    const int schs = sch_ok ? sch : 0;
    f2_t s8[4], t8[4];
    {
        float sf[8], tf[8];
        ldc4(a.in.v1 + schs, sf); ldc4(a.in.v1 + schs + 4, sf + 4);
        ldc4(a.in.v2 + schs, tf); ldc4(a.in.v2 + schs + 4, tf + 4);
#pragma unroll
        for (int i = 0; i < 4; ++i) { s8[i] = f2_t{sf[2 * i], sf[2 * i + 1]}; t8[i] = f2_t{tf[2 * i], tf[2 * i + 1]}; }
    }
    for (int tile_id = blockIdx.x; tile_id < ntiles; tile_id += gridDim.x) {
        const int plane = tile_id / nbands, band = tile_id % nbands;
        const int ho0 = band * a.rows_band;
        const int nro = (a.Hout - ho0 < a.rows_band) ? a.Hout - ho0 : a.rows_band;
        const int hi0 = ho0 * ST - P;
        const int rows_in = (nro - 1) * ST + 3;
        const i64 plane_row0 = (i64)plane * a.Hin * a.Win;
        const int total_st = rows_in * Wpp;
        const T* in0 = inp + plane_row0 * a.in.ld + schs;
        for (int f0 = spl; f0 < total_st; f0 += NB * LPS) {
            uint4 ra[NB], rb[NB];
            unsigned msk[NB];
#pragma unroll
            for (int u = 0; u < NB; ++u) {
                const int f = f0 + u * LPS;
                const int r = dvpp.div(f);
                const int xp = dvpp.rem(f, r);
                const int hi = hi0 + r, wi0 = 2 * xp - P;
                const bool rok = sch_ok && f < total_st && (unsigned)hi < (unsigned)a.Hin;
                const bool ok0 = rok && (unsigned)wi0 < (unsigned)a.Win, ok1 = rok && (unsigned)(wi0 + 1) < (unsigned)a.Win;
                const int rowpix = __mul24(hi, a.Win);
                ra[u] = *reinterpret_cast<const uint4*>(in0 + (ok0 ? __umul24((unsigned)(rowpix + wi0), (unsigned)a.in.ld) : 0u));
                rb[u] = *reinterpret_cast<const uint4*>(in0 + (ok1 ? __umul24((unsigned)(rowpix + wi0 + 1), (unsigned)a.in.ld) : 0u));
                msk[u] = (ok0 ? 0x0000ffffu : 0u) | (ok1 ? 0xffff0000u : 0u);
            }
#pragma unroll
            for (int u = 0; u < NB; ++u) {
                const int f = f0 + u * LPS;
                if (f < total_st) {
                    const unsigned wa[4] = {ra[u].x, ra[u].y, ra[u].z, ra[u].w};
                    const unsigned wb[4] = {rb[u].x, rb[u].y, rb[u].z, rb[u].w};
                    unsigned o[8];
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const f2_t ya = f2_t{__uint_as_float(wa[i] << 16), __uint_as_float(wa[i] & 0xffff0000u)};
                        const f2_t yb = f2_t{__uint_as_float(wb[i] << 16), __uint_as_float(wb[i] & 0xffff0000u)};
                        const f2_t ha = ya * s8[i] + t8[i], hb = yb * s8[i] + t8[i];
                        const f2_t za = ha * sigmoid2f_(ha);
                        const f2_t zb = hb * sigmoid2f_(hb);
                        o[2 * i] = pack_bf16x2(za.x, zb.x) & msk[u];
                        o[2 * i + 1] = pack_bf16x2(za.y, zb.y) & msk[u];
                    }
                    uint4* dst = reinterpret_cast<uint4*>(tile + f * CS + scv * 8);
                    dst[0] = make_uint4(o[0], o[1], o[2], o[3]);
                    dst[1] = make_uint4(o[4], o[5], o[6], o[7]);
                }
            }
        }
        __syncthreads();
        {
            const int total = nro * Wop;
            const i64 orow0 = ((i64)plane * a.Hout + ho0) * a.Wout;
            T* out0 = outp + orow0 * a.C + chan;
            const int rowdw = Wpp * CS;                  // dwords per staged row
            for (int i = pl; i < total; i += LP) {
                const int oy = dvop.div(i);
                const int j = dvop.rem(i, oy);
                const int ox0 = 2 * j;
                float acc0[4] = {0.f, 0.f, 0.f, 0.f}, acc1[4] = {0.f, 0.f, 0.f, 0.f};
                if constexpr (ST == 1) {
                    const unsigned* tp = tile + (__mul24(oy, Wpp) + j) * CS + cv * 4;
#pragma unroll
                    for (int dy = 0; dy < 3; ++dy) {
                        const uint4 p0 = *reinterpret_cast<const uint4*>(tp + dy * rowdw);
                        const uint4 p1 = *reinterpret_cast<const uint4*>(tp + dy * rowdw + CS);
                        const unsigned a0[4] = {p0.x, p0.y, p0.z, p0.w}, a1[4] = {p1.x, p1.y, p1.z, p1.w};
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            acc0[q] = dot2_bf16(a0[q], wp[dy][0][q], acc0[q]);
                            acc0[q] = dot2_bf16(a1[q], wp[dy][1][q], acc0[q]);
                            acc1[q] = dot2_bf16(a0[q], wp[dy][2][q], acc1[q]);
                            acc1[q] = dot2_bf16(a1[q], wp[dy][3][q], acc1[q]);
                        }
                    }
                } else {
                    // stride 2: output ox reads staged columns 2ox .. 2ox+2 = P[ox] and the low half of P[ox+1]
                    const int jp2 = (ox0 + 2 < Wpp) ? 2 : 1;      // P[ox0+2] only feeds the (possibly invalid) odd output
                    const unsigned* tp = tile + (__mul24(oy * 2, Wpp) + ox0) * CS + cv * 4;
#pragma unroll
                    for (int dy = 0; dy < 3; ++dy) {
                        const uint4 p0 = *reinterpret_cast<const uint4*>(tp + dy * rowdw);
                        const uint4 p1 = *reinterpret_cast<const uint4*>(tp + dy * rowdw + CS);
                        const uint4 p2 = *reinterpret_cast<const uint4*>(tp + dy * rowdw + jp2 * CS);
                        const unsigned a0[4] = {p0.x, p0.y, p0.z, p0.w}, a1[4] = {p1.x, p1.y, p1.z, p1.w};
                        const unsigned a2[4] = {p2.x, p2.y, p2.z, p2.w};
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            acc0[q] = dot2_bf16(a0[q], wp[dy][0][q], acc0[q]);
                            acc0[q] = dot2_bf16(a1[q], wp[dy][1][q], acc0[q]);
                            acc1[q] = dot2_bf16(a1[q], wp[dy][0][q], acc1[q]);
                            acc1[q] = dot2_bf16(a2[q], wp[dy][1][q], acc1[q]);
                        }
                    }
                }
                if (chan_ok) {
                    const unsigned pix = (unsigned)(__mul24(oy, a.Wout) + ox0);
                    const uint2 pk0 = make_uint2(pack_bf16x2(acc0[0], acc0[1]), pack_bf16x2(acc0[2], acc0[3]));
                    T* dst = out0 + __umul24(pix, (unsigned)a.C);
                    *reinterpret_cast<uint2*>(dst) = pk0;
                    f2_t r0, r1;
                    unpack_pairs<T>(pk0, r0, r1);
                    st0[0] += r0; st0[1] += r1; st1[0] += r0 * r0; st1[1] += r1 * r1;
                    if (ox0 + 1 < a.Wout) {
                        const uint2 pk1 = make_uint2(pack_bf16x2(acc1[0], acc1[1]), pack_bf16x2(acc1[2], acc1[3]));
                        *reinterpret_cast<uint2*>(dst + a.C) = pk1;
                        unpack_pairs<T>(pk1, r0, r1);
                        st0[0] += r0; st0[1] += r1; st1[0] += r0 * r0; st1[1] += r1 * r1;
                    }
                }
            }
        }
        __syncthreads();
    }
    if (a.stats) {
        const float s0[4] = {st0[0].x, st0[0].y, st0[1].x, st0[1].y}, s1[4] = {st1[0].x, st1[0].y, st1[1].x, st1[1].y};
        block_stats_flush<T>(lstat, s0, s1, cv, c0, a.C, a.stats, blockIdx.x % DWN_NREP);
    }
    DET_EXIT();
}

// ------------------------------------------------------------------------------------------------
// spatial backward: dh1 = (dwS^T dy2) * silu'(h1), dW, Σdh1, Σdh1·ŷ1
// ------------------------------------------------------------------------------------------------
template <typename T, int KS, int ST>
__global__ __launch_bounds__(DWS_BWD_THREADS, (ST == 1 ? DWS_BWD_MINB1 : 3)) void dw_spatial_bwd_kernel(const DwSpatialBwd a) {
    constexpr int NT = DWS_BWD_THREADS;
    constexpr int NCV = SL<T>::NCV, CS = SL<T>::CS, LP = NT / NCV, P = KS / 2;
    constexpr int XB = 4;
    typedef typename SL<T>::raw_t raw_t;
    __shared__ float lstat[2 * CS];
    __shared__ __attribute__((aligned(16))) float lw[KS * KS * CS];      // stencil weights (broadcast reads), later dW
    const int tid = threadIdx.x;
    const int cv = tid % NCV, pl = tid / NCV;
    const int c0 = blockIdx.y * CS;
    const int chan = c0 + cv * 4;
    const bool chan_ok = chan < a.C;
    const int chs = chan_ok ? chan : 0;
    if (tid < 2 * CS) lstat[tid] = 0.f;
    for (int i = tid; i < KS * KS * CS; i += NT) {
        int k = i / CS, c = c0 + i % CS;
        lw[i] = c < a.C ? a.w[(i64)k * a.C + c] : 0.f;
    }
    __syncthreads();

    f2_t dwp[KS * KS][2];
#pragma unroll
    for (int k = 0; k < KS * KS; ++k) { dwp[k][0] = f2_t{0.f, 0.f}; dwp[k][1] = f2_t{0.f, 0.f}; }
    float bs[4], bt[4], bm[4], bi[4];
    ldc4(a.y1.v1 + chs, bs); ldc4(a.y1.v2 + chs, bt); ldc4(a.y1.v3 + chs, bm); ldc4(a.y1.v4 + chs, bi);
    const f2_t bs2[2] = {f2_t{bs[0], bs[1]}, f2_t{bs[2], bs[3]}}, bt2[2] = {f2_t{bt[0], bt[1]}, f2_t{bt[2], bt[3]}};
    const f2_t bi2[2] = {f2_t{bi[0], bi[1]}, f2_t{bi[2], bi[3]}};
    const f2_t nbm2[2] = {f2_t{-bm[0] * bi[0], -bm[1] * bi[1]}, f2_t{-bm[2] * bi[2], -bm[3] * bi[3]}};   // yhat = y*bi + nbm
    f2_t sp0[2] = {f2_t{0.f, 0.f}, f2_t{0.f, 0.f}}, sp1[2] = {f2_t{0.f, 0.f}, f2_t{0.f, 0.f}};

    const int Wq = a.Wout + 2;                      // staged columns wo = -1 .. Wout
    const FastDiv dvq(Wq), dvw(a.Win);
    const int nbands = (a.Hin + a.rows_band - 1) / a.rows_band;
    const int ntiles = a.planes * nbands;
    const int s = ST > 0 ? ST : a.stride;
    T* dhp = reinterpret_cast<T*>(a.dh1);
    const T* y1p = reinterpret_cast<const T*>(a.y1.p);
    const T* dpp = reinterpret_cast<const T*>(a.dy.p);
    const T* dqp = reinterpret_cast<const T*>(a.dy.q);
    raw_t* tile = reinterpret_cast<raw_t*>(dyn_smem);
    for (int tile_id = blockIdx.x; tile_id < ntiles; tile_id += gridDim.x) {
        const int plane = tile_id / nbands, band = tile_id % nbands;
        const int hi0 = band * a.rows_band;
        const int nri = (a.Hin - hi0 < a.rows_band) ? a.Hin - hi0 : a.rows_band;
        // output rows that touch input rows [hi0, hi0+nri): ho = (hi + P - dy) / s
        const int lo_num = hi0 + P - (KS - 1);
        const int ho_lo = lo_num >= 0 ? lo_num / s : -((-lo_num + s - 1) / s);
        const int ho_hi = (hi0 + nri - 1 + P) / s;
        const int rows_q = ho_hi - ho_lo + 1;
        const i64 orow0 = (i64)plane * a.Hout * a.Wout;
        // stage dL/dy2 = A1*dh2 + A2*y2 + A3 (BatchNorm backward) with zero padding; flat walk over the (row, column)
        // entries of the tile (flat index == tile index), 2*NB loads in flight per thread
        {
            constexpr int NB = 8;
            const int total_st = rows_q * Wq;
            // BN-backward coefficients: (re)loaded per tile so they do not occupy registers during the tap phase
            float a1[4], a2[4], a3[4];
            ldc4(a.dy.v1 + chs, a1); ldc4(a.dy.v2 + chs, a2); ldc4(a.dy.v3 + chs, a3);
            const f2_t a1v[2] = {f2_t{a1[0], a1[1]}, f2_t{a1[2], a1[3]}}, a2v[2] = {f2_t{a2[0], a2[1]}, f2_t{a2[2], a2[3]}};
            const f2_t a3v[2] = {f2_t{a3[0], a3[1]}, f2_t{a3[2], a3[3]}};
            const T* dp0 = dpp + orow0 * a.dy.ld + chs;
            const T* dq0 = dqp + orow0 * a.dy.ld + chs;
            for (int f0 = pl; f0 < total_st; f0 += NB * LP) {
                raw_t rp[NB], rq[NB];
                bool okv[NB];
#pragma unroll
                for (int u = 0; u < NB; ++u) {
                    const int f = f0 + u * LP;
                    const int r = dvq.div(f);
                    const int ho = ho_lo + r, wo = dvq.rem(f, r) - 1;
                    okv[u] = chan_ok && f < total_st && (unsigned)ho < (unsigned)a.Hout && (unsigned)wo < (unsigned)a.Wout;
                    const unsigned off = okv[u] ? __umul24(__mul24(ho, a.Wout) + wo, (unsigned)a.dy.ld) : 0u;
                    rp[u] = ld4_raw<T>(dp0 + off);
                    rq[u] = ld4_raw<T>(dq0 + off);
                }
#pragma unroll
                for (int u = 0; u < NB; ++u) {
                    const int f = f0 + u * LP;
                    if (f < total_st) {
                        f2_t p0, p1, q0, q1;
                        unpack_pairs<T>(rp[u], p0, p1);
                        unpack_pairs<T>(rq[u], q0, q1);
                        p0 = a1v[0] * p0 + (a2v[0] * q0 + a3v[0]);
                        p1 = a1v[1] * p1 + (a2v[1] * q1 + a3v[1]);
                        const float p[4] = {p0.x, p0.y, p1.x, p1.y};
                        tile[f * NCV + cv] = okv[u] ? V4<T>::pack(p) : V4<T>::zero();
                    }
                }
            }
        }
        __syncthreads();
        {
            const i64 prow0 = (i64)plane * a.Hin * a.Win;
            // one tap: g = dL/dy2 at (ho, wo) from the LDS tile; dz += w*g; dW[tap] += z1*g   (channel pairs)
            auto tap_at = [&](const int k, const raw_t* tp, const f2_t* z1, f2_t* dz) {
                f2_t g0, g1;
                unpack_pairs<T>(*tp, g0, g1);
                const float4 wv = *reinterpret_cast<const float4*>(&lw[k * CS + cv * 4]);
                dz[0] += f2_t{wv.x, wv.y} * g0;
                dz[1] += f2_t{wv.z, wv.w} * g1;
                dwp[k][0] += z1[0] * g0;
                dwp[k][1] += z1[1] * g1;
            };
            auto tap = [&](const int k, const int ho, const int wo, const f2_t* z1, f2_t* dz) {
                tap_at(k, tile + ((ho - ho_lo) * Wq + wo + 1) * NCV + cv, z1, dz);
            };
            // everything after the taps: dh1 = dz * silu'(h1), store, BN-backward sums
            // per-plane base pointers + 32-bit in-plane offsets (24-bit multiplies): no 64-bit VALU address math
            T* dh0 = dhp + prow0 * a.C + chan;
            const T* y10 = y1p + prow0 * a.y1.ld + chs;
            auto finish_at = [&](const int pix, const f2_t* y, const f2_t* dsl, const f2_t* dz) {
                const f2_t d0 = dz[0] * dsl[0], d1 = dz[1] * dsl[1];
                float dh[4] = {d0.x, d0.y, d1.x, d1.y};
                const typename V4<T>::raw_t packed = V4<T>::pack(dh);
                *reinterpret_cast<typename V4<T>::raw_t*>(dh0 + __umul24((unsigned)pix, (unsigned)a.C)) = packed;
                f2_t r0, r1;
                unpack_pairs<T>(packed, r0, r1);              // statistics of the values as stored
                sp0[0] += r0; sp0[1] += r1;
                sp1[0] += r0 * (y[0] * bi2[0] + nbm2[0]);
                sp1[1] += r1 * (y[1] * bi2[1] + nbm2[1]);
            };
            auto finish = [&](const int hi, const int wi, const f2_t* y, const f2_t* dsl, const f2_t* dz) {
                finish_at(hi * a.Win + wi, y, dsl, dz);
            };
            auto activate = [&](const raw_t& raw, f2_t* y, f2_t* z1, f2_t* dsl) {
                unpack_pairs<T>(raw, y[0], y[1]);
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    const f2_t h = y[i] * bs2[i] + bt2[i];
                    const f2_t sg = sigmoid2f_(h);
                    z1[i] = h * sg;
                    dsl[i] = sg * (1.0f + h * (1.0f - sg));
                }
            };
            // Pixels are walked as ONE flat index space (rows x columns of the band / of a parity class), XB loads in
            // flight per thread, so narrow planes (W = 8..16) still fill all pixel lanes and every load batch.
            if constexpr (ST == 2 && KS == 3) {
                // stride 2: an input pixel receives 1, 2, 2 or 4 taps depending on the parity of (hi, wi): one pass per
                // parity class, taps known at compile time.
                auto run_class = [&](auto hp_c, auto px_c) {
                    constexpr int HP = decltype(hp_c)::value, PX = decltype(px_c)::value;
                    const int hfirst = hi0 + (((hi0 & 1) == HP) ? 0 : 1);
                    const int nr = (hi0 + nri - hfirst + 1) >> 1;
                    const int nc = (a.Win - PX + 1) >> 1;
                    const int total = nr > 0 ? nr * nc : 0;
                    // pixel (k, j) of the class: hi = hfirst + 2k, wi = PX + 2j.  Its taps read tile entries at
                    // workgroup-uniform offsets from entry (k, j): one index per pixel, uniform offsets per tap.
                    const int wdiff = Wq - nc;
                    const FastDiv dvc(nc > 0 ? nc : 1);
                    const int r_dn = ((hfirst + HP) >> 1) - ho_lo;          // row of (hi + HP) >> 1 at k = 0
                    const int o00 = (r_dn * Wq + 1) * NCV;                  // (hi+HP)>>1, (wi-PX)>>1 ... column j
                    const int o01 = o00 + NCV;                              // column j + 1
                    const int o10 = o00 - Wq * NCV, o11 = o01 - Wq * NCV;   // row (hi-1)>>1 (odd rows only)
                    for (int i0 = pl; i0 < total; i0 += XB * LP) {
                        raw_t ry[XB];
                        int ti[XB];
                        int gg[XB];
#pragma unroll
                        for (int u = 0; u < XB; ++u) {
                            const int i = i0 + u * LP;
                            const bool ok = chan_ok && i < total;
                            const int iv = ok ? i : 0;
                            const int k = dvc.div(iv);
                            const int j = dvc.rem(iv, k);
                            ti[u] = (iv + __mul24(k, wdiff)) * NCV + cv;
                            gg[u] = __mul24(hfirst + 2 * k, a.Win) + (PX + 2 * j);
                            ry[u] = ld4_raw<T>(y10 + __umul24((unsigned)gg[u], (unsigned)a.y1.ld));
                        }
#pragma unroll
                        for (int u = 0; u < XB; ++u) {
                            if (!chan_ok || i0 + u * LP >= total) continue;
                            f2_t y[2], z1[2], dsl[2], dz[2] = {f2_t{0.f, 0.f}, f2_t{0.f, 0.f}};
                            activate(ry[u], y, z1, dsl);
                            const raw_t* tp = tile + ti[u];
                            if constexpr (HP == 0 && PX == 0) {
                                tap_at(4, tp + o00, z1, dz);
                            } else if constexpr (HP == 0 && PX == 1) {
                                tap_at(3, tp + o01, z1, dz); tap_at(5, tp + o00, z1, dz);
                            } else if constexpr (HP == 1 && PX == 0) {
                                tap_at(1, tp + o00, z1, dz); tap_at(7, tp + o10, z1, dz);
                            } else {
                                tap_at(0, tp + o01, z1, dz); tap_at(2, tp + o00, z1, dz);
                                tap_at(6, tp + o11, z1, dz); tap_at(8, tp + o10, z1, dz);
                            }
                            finish_at(gg[u], y, dsl, dz);
                        }
                    }
                };
                run_class(std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{});
                run_class(std::integral_constant<int, 0>{}, std::integral_constant<int, 1>{});
                run_class(std::integral_constant<int, 1>{}, std::integral_constant<int, 0>{});
                run_class(std::integral_constant<int, 1>{}, std::integral_constant<int, 1>{});
            } else {
                const int total = nri * a.Win;
                if constexpr (ST == 1) {
                    // stride 1: every tap is valid (the zero halo supplies the borders): branch-free.  The tile entry of
                    // tap (dy, dx) sits at a fixed offset from the pixel's own entry: one index per pixel, not per tap.
                    const int wdiff = Wq - a.Win;                // staged row is wider than the input row
                    const int rowstep = Wq * NCV;
                    const int grow0 = hi0 * a.Win;
                    for (int i0 = pl; i0 < total; i0 += XB * LP) {
                        raw_t ry[XB];
                        int ii[XB], ti[XB];
#pragma unroll
                        for (int u = 0; u < XB; ++u) {
                            const int i = i0 + u * LP;
                            const bool ok = chan_ok && i < total;
                            ii[u] = ok ? i : 0;
                            const int iy = dvw.div(ii[u]);
                            ti[u] = (ii[u] + __mul24(iy, wdiff) + (P + 2 - KS)) * NCV + cv;
                            ry[u] = ld4_raw<T>(y10 + __umul24((unsigned)(grow0 + ii[u]), (unsigned)a.y1.ld));
                        }
#pragma unroll
                        for (int u = 0; u < XB; ++u) {
                            if (!chan_ok || i0 + u * LP >= total) continue;
                            f2_t y[2], z1[2], dsl[2], dz[2] = {f2_t{0.f, 0.f}, f2_t{0.f, 0.f}};
                            activate(ry[u], y, z1, dsl);
                            const raw_t* tp = tile + ti[u];
#pragma unroll
                            for (int dy = 0; dy < KS; ++dy) {
                                const raw_t* tr = tp + (KS - 1 - dy) * rowstep;
#pragma unroll
                                for (int dx = 0; dx < KS; ++dx) tap_at(dy * KS + dx, tr + (KS - 1 - dx) * NCV, z1, dz);
                            }
                            finish_at(grow0 + ii[u], y, dsl, dz);
                        }
                    }
                } else {
                for (int i0 = pl; i0 < total; i0 += XB * LP) {
                    raw_t ry[XB];
                    int hh[XB], ww[XB];
#pragma unroll
                    for (int u = 0; u < XB; ++u) {
                        const int i = i0 + u * LP;
                        const bool ok = chan_ok && i < total;
                        const int iy = ok ? i / a.Win : 0;
                        hh[u] = hi0 + iy;
                        ww[u] = (ok ? i : 0) - iy * a.Win;
                        ry[u] = ld4_raw<T>(y10 + (ok ? (i64)hh[u] * a.Win + ww[u] : 0) * a.y1.ld);
                    }
#pragma unroll
                    for (int u = 0; u < XB; ++u) {
                        if (!chan_ok || i0 + u * LP >= total) continue;
                        const int hi = hh[u], wi = ww[u];
                        f2_t y[2], z1[2], dsl[2], dz[2] = {f2_t{0.f, 0.f}, f2_t{0.f, 0.f}};
                        activate(ry[u], y, z1, dsl);
#pragma unroll
                        for (int dy = 0; dy < KS; ++dy) {
                            const int nh = hi + P - dy;
                            if (nh < 0 || nh % s != 0) continue;
                            const int ho = nh / s;
                            if (ho > ho_hi) continue;
#pragma unroll
                            for (int dx = 0; dx < KS; ++dx) {
                                const int nw = wi + P - dx;
                                if (nw < 0 || nw % s != 0) continue;
                                const int wo = nw / s;
                                if (wo > a.Wout) continue;
                                tap(dy * KS + dx, ho, wo, z1, dz);
                            }
                        }
                        finish(hi, wi, y, dsl, dz);
                    }
                }
                }
            }
        }
        __syncthreads();
    }
    // weight gradient: reduce over the threads sharing a channel vector through LDS, then global fp32 atomics
    __syncthreads();
    for (int i = tid; i < KS * KS * CS; i += NT) lw[i] = 0.f;
    __syncthreads();
    DET_WAVES_BEGIN
    if (chan_ok) {
#pragma unroll
        for (int k = 0; k < KS * KS; ++k) {
            atomicAdd(&lw[k * CS + cv * 4 + 0], dwp[k][0].x);
            atomicAdd(&lw[k * CS + cv * 4 + 1], dwp[k][0].y);
            atomicAdd(&lw[k * CS + cv * 4 + 2], dwp[k][1].x);
            atomicAdd(&lw[k * CS + cv * 4 + 3], dwp[k][1].y);
        }
    }
    DET_WAVES_END
    __syncthreads();
    DET_ENTER();
    for (int i = tid; i < KS * KS * CS; i += NT) {
        int k = i / CS, c = c0 + i % CS;
        if (c < a.C) atomicAdd(a.dw + (i64)c * (KS * KS) + k, lw[i]);
    }
    const float st0[4] = {sp0[0].x, sp0[0].y, sp0[1].x, sp0[1].y}, st1[4] = {sp1[0].x, sp1[0].y, sp1[1].x, sp1[1].y};
    if (a.stats) block_stats_flush<T>(lstat, st0, st1, cv, c0, a.C, a.stats, blockIdx.x % DWN_NREP);
    DET_EXIT();
}

// ------------------------------------------------------------------------------------------------
// temporal forward: y3[t] = sum_k w[k] z2[t + k - P]
// ------------------------------------------------------------------------------------------------
// ZOUT (eval mode): the output is z3 = SiLU(BN3(y3)) and the SqueezeExcite pooling sums ride along (see dwn.h)
template <typename T, int KT, bool ZOUT = false>
__global__ __launch_bounds__(256) void dw_temporal_fwd_kernel(const DwTemporalFwd a) {
    constexpr int NCV = SL<T>::NCV, CS = SL<T>::CS, LP = SL<T>::LP, P = KT / 2;
    typedef typename SL<T>::raw_t raw_t;
    __shared__ float lstat[2 * CS];
    const int tid = threadIdx.x;
    const int cv = tid % NCV, pl = tid / NCV;
    const int c0 = blockIdx.y * CS;
    const int chan = c0 + cv * 4;
    const bool chan_ok = chan < a.C;
    const int chs = chan_ok ? chan : 0;
    if (tid < 2 * CS) lstat[tid] = 0.f;
    __syncthreads();
    float w[KT][4], bs[4], bt[4];
#pragma unroll
    for (int k = 0; k < KT; ++k) ldc4(a.w + (i64)k * a.C + chs, w[k]);
    ldc4(a.in.v1 + chs, bs);
    ldc4(a.in.v2 + chs, bt);
    float st0[4] = {0.f, 0.f, 0.f, 0.f}, st1[4] = {0.f, 0.f, 0.f, 0.f};
    [[maybe_unused]] float zs[4], zt[4];
    if constexpr (ZOUT) { ldc4(a.z_scale + chs, zs); ldc4(a.z_shift + chs, zt); }

    const i64 npos = (i64)a.B * a.HW;
    const T* inp = reinterpret_cast<const T*>(a.in.p);
    T* outp = reinterpret_cast<T*>(a.out);
    const i64 tstride = (i64)a.HW * a.C;
    if (chan_ok) {
        // ZOUT: a workgroup takes a CONTIGUOUS run of positions, so its lanes stay inside one sample for many iterations and the
        // SqueezeExcite pooling sums wait in registers (pacc) until the sample changes: ~10x fewer atomics than a flush per
        // position (5 M eight-byte atomics on 40 K addresses per launch at the inference batch)
        i64 p_beg = (i64)blockIdx.x * LP + pl, p_end = npos, p_step = (i64)gridDim.x * LP;
        if constexpr (ZOUT) {
            const i64 per = ((npos + (i64)gridDim.x * LP - 1) / ((i64)gridDim.x * LP)) * LP;
            p_beg = (i64)blockIdx.x * per + pl;
            p_end = p_beg - pl + per < npos ? p_beg - pl + per : npos;
            p_step = LP;
        }
        [[maybe_unused]] float pacc[4] = {0.f, 0.f, 0.f, 0.f};
        [[maybe_unused]] i64 pacc_b = -1;
        // adds pacc into pooled[pacc_b] (64-bit fixed point, pool_fix: whatever the arrival order, the same total).  The wave's
        // pixel lanes that share a channel vector hold one sample almost always: then they are folded with xor-shuffles and
        // the first NCV lanes add 4 channels each.  Called by all lanes that are still in the loop (or by all, after it).
        [[maybe_unused]] auto pool_flush = [&]() {
            const int b0 = __builtin_amdgcn_readfirstlane((int)pacc_b);
            const bool uniform = __popcll(__ballot(1)) == 64 && __all((int)pacc_b == b0) && b0 >= 0;
            if (uniform) {
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int o = NCV; o < 64; o <<= 1) pacc[i] += __shfl_xor(pacc[i], o);
                if ((tid & 63) < NCV) {
                    unsigned long long* dst = reinterpret_cast<unsigned long long*>(a.pooled) + (i64)b0 * a.C + chan;
#pragma unroll
                    for (int i = 0; i < 4; ++i) atomicAdd(dst + i, (unsigned long long)pool_fix(pacc[i]));
                }
            } else if (pacc_b >= 0) {
                unsigned long long* dst = reinterpret_cast<unsigned long long*>(a.pooled) + pacc_b * a.C + chan;
#pragma unroll
                for (int i = 0; i < 4; ++i) atomicAdd(dst + i, (unsigned long long)pool_fix(pacc[i]));
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) pacc[i] = 0.f;
        };
        for (i64 pos = p_beg; pos < p_end; pos += p_step) {
            const i64 b = pos / a.HW, hw = pos % a.HW;
            const T* ip = inp + (b * a.T * a.HW + hw) * a.C + chan;      // element (t = 0)
            T* op = outp + (b * a.T * a.HW + hw) * a.C + chan;
            // ring of KT slots, KT frames per unrolled batch: at step u of a batch z2(t + k - P) lives in slot (u + k) % KT — compile-time
            // indices, no register shifting (the shifting version spent 52 moves per 16 outputs)
            float win[KT][4];
            [[maybe_unused]] float ps[4] = {0.f, 0.f, 0.f, 0.f};         // ZOUT: this position's sums over t of the stored z3
#pragma unroll
            for (int k = 0; k < KT - 1; ++k) {
                int t = k - P;
                if (t >= 0 && t < a.T) { ld4<T>(ip + t * tstride, win[k]); bn_silu4(win[k], bs, bt); }
                else { win[k][0] = win[k][1] = win[k][2] = win[k][3] = 0.f; }
            }
            for (int t0 = 0; t0 < a.T; t0 += KT) {
                raw_t raw[KT];
#pragma unroll
                for (int u = 0; u < KT; ++u) {
                    int tl = t0 + u + P;
                    raw[u] = ld4_raw<T>(ip + (tl < a.T ? tl : 0) * tstride);
                }
#pragma unroll
                for (int u = 0; u < KT; ++u) {
                    int t = t0 + u;
                    if (t >= a.T) break;
                    const int sn = (u + KT - 1) % KT;                   // newest slot: frame t + P
                    if (t + P < a.T) { V4<T>::unpack(raw[u], win[sn]); bn_silu4(win[sn], bs, bt); }
                    else { win[sn][0] = win[sn][1] = win[sn][2] = win[sn][3] = 0.f; }
                    float acc[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int k = 0; k < KT; ++k)
#pragma unroll
                        for (int i = 0; i < 4; ++i) acc[i] = fmaf(w[k][i], win[(u + k) % KT][i], acc[i]);
                    if constexpr (ZOUT) {
                        float z[4];
#pragma unroll
                        for (int i = 0; i < 4; ++i) z[i] = fmaf(round_t<T>(acc[i]), zs[i], zt[i]);     // BN3 + SiLU of the value a stored y3 holds
                        silu_n<4>(z);
#pragma unroll
                        for (int i = 0; i < 4; ++i) ps[i] += round_t<T>(z[i]);
                        st4<T>(op + t * tstride, z);
                    } else {
                        st4<T>(op + t * tstride, acc);
#pragma unroll
                        for (int i = 0; i < 4; ++i) {
                            float r = round_t<T>(acc[i]);
                            st0[i] += r;
                            st1[i] += r * r;
                        }
                    }
                }
            }
            if constexpr (ZOUT) {
                if (a.pooled) {
                    if (__any(b != pacc_b)) { pool_flush(); pacc_b = b; }      // a lane entered another sample: everybody flushes
#pragma unroll
                    for (int i = 0; i < 4; ++i) pacc[i] += ps[i];
                }
            }
        }
        if constexpr (ZOUT) {
            if (a.pooled) pool_flush();
        }
    }
    if (a.stats) block_stats_flush<T>(lstat, st0, st1, cv, c0, a.C, a.stats, blockIdx.x % DWN_NREP);
    DET_EXIT();
}

// ------------------------------------------------------------------------------------------------
// temporal backward: dz2[t] = sum_k w[k] dy3[t - k + P];  dW[k] = sum z2[t] dy3[t - k + P];
//                    dh2 = dz2 * silu'(h2);  Σdh2, Σdh2·ŷ2
// dy3 is never materialised: DYK = LD_DY3 rebuilds it from (du, y3) — SE gate/gradient, SiLU' and the bn3
// backward affine — DYK = LD_AFFINE2 from (dh3, y3).
// ------------------------------------------------------------------------------------------------
template <typename T, int KT, int DYK, int TB>
__global__ __launch_bounds__(256, TB >= 8 ? 2 : 3) void dw_temporal_bwd_kernel(const DwTemporalBwd a) {
    constexpr int NCV = SL<T>::NCV, CS = SL<T>::CS, LP = SL<T>::LP, P = KT / 2;
    typedef typename SL<T>::raw_t raw_t;
    __shared__ float lstat[2 * CS];
    __shared__ float lw[KT * CS];
    const int tid = threadIdx.x;
    const int cv = tid % NCV, pl = tid / NCV;
    const int c0 = blockIdx.y * CS;
    const int chan = c0 + cv * 4;
    const bool chan_ok = chan < a.C;
    const int chs = chan_ok ? chan : 0;
    if (tid < 2 * CS) lstat[tid] = 0.f;
    for (int i = tid; i < KT * CS; i += 256) lw[i] = 0.f;
    __syncthreads();
    float w[KT][4], dwacc[KT][4];
#pragma unroll
    for (int k = 0; k < KT; ++k) {
        ldc4(a.w + (i64)k * a.C + chs, w[k]);
        dwacc[k][0] = dwacc[k][1] = dwacc[k][2] = dwacc[k][3] = 0.f;
    }
    float bs[4], bt[4], bm[4], bi[4], a1[4], a2[4], a3[4], s3[4] = {0, 0, 0, 0}, t3[4] = {0, 0, 0, 0};
    ldc4(a.y2.v1 + chs, bs); ldc4(a.y2.v2 + chs, bt); ldc4(a.y2.v3 + chs, bm); ldc4(a.y2.v4 + chs, bi);
    ldc4(a.dy.v1 + chs, a1); ldc4(a.dy.v2 + chs, a2); ldc4(a.dy.v3 + chs, a3);
    if constexpr (DYK == LD_DY3) { ldc4(a.dy.v4 + chs, s3); ldc4(a.dy.v5 + chs, t3); }
    float st0[4] = {0.f, 0.f, 0.f, 0.f}, st1[4] = {0.f, 0.f, 0.f, 0.f};

    const i64 npos = (i64)a.B * a.HW;
    T* dhp = reinterpret_cast<T*>(a.dh2);
    const T* y2p = reinterpret_cast<const T*>(a.y2.p);
    const T* dpp = reinterpret_cast<const T*>(a.dy.p);
    const T* dqp = reinterpret_cast<const T*>(a.dy.q);
    const i64 tstride = (i64)a.HW * a.C;

    auto make_dy = [&](const raw_t& rp, const raw_t& rq, const float* g, const float* g2, float* o) {
        float p[4], q[4];
        V4<T>::unpack(rp, p);
        V4<T>::unpack(rq, q);
        [[maybe_unused]] float sp4[4];
        if constexpr (DYK == LD_DY3) {
            float h4[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) h4[i] = fmaf(q[i], s3[i], t3[i]);
            silu_grad_n<4>(h4, sp4);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            if constexpr (DYK == LD_DY3) {
                float dh = fmaf(p[i], g[i], g2[i]) * sp4[i];
                o[i] = fmaf(a1[i], dh, fmaf(a2[i], q[i], a3[i]));
            } else {
                o[i] = fmaf(a1[i], p[i], fmaf(a2[i], q[i], a3[i]));
            }
        }
    };

    if (chan_ok) {
        for (i64 pos = (i64)blockIdx.x * LP + pl; pos < npos; pos += (i64)gridDim.x * LP) {
            const i64 b = pos / a.HW, hw = pos % a.HW;
            const i64 e0 = (b * a.T * a.HW + hw) * a.C + chan;
            float g[4] = {1.f, 1.f, 1.f, 1.f}, g2[4] = {0.f, 0.f, 0.f, 0.f};
            if constexpr (DYK == LD_DY3) {
                ldc4(a.dy.gate + b * a.dy.gate_ld + chan, g);
                ldc4(a.dy.gate2 + b * a.dy.gate_ld + chan, g2);
            }
            float win[KT][4];                           // win[j] = dy3(t + j - P)
#pragma unroll
            for (int k = 0; k < KT - 1; ++k) {
                int t = k - P;
                if (t >= 0 && t < a.T) make_dy(ld4_raw<T>(dpp + e0 + t * tstride), ld4_raw<T>(dqp + e0 + t * tstride), g, g2, win[k]);
                else { win[k][0] = win[k][1] = win[k][2] = win[k][3] = 0.f; }
            }
            for (int t0 = 0; t0 < a.T; t0 += TB) {
                raw_t rp[TB], rq[TB], ry[TB];
#pragma unroll
                for (int u = 0; u < TB; ++u) {
                    int tl = t0 + u + P;
                    i64 off = e0 + (tl < a.T ? tl : 0) * tstride;
                    rp[u] = ld4_raw<T>(dpp + off);
                    rq[u] = ld4_raw<T>(dqp + off);
                    int ty = t0 + u;
                    ry[u] = ld4_raw<T>(y2p + e0 + (ty < a.T ? ty : 0) * tstride);
                }
#pragma unroll
                for (int u = 0; u < TB; ++u) {
                    int t = t0 + u;
                    if (t >= a.T) break;
                    if (t + P < a.T) make_dy(rp[u], rq[u], g, g2, win[KT - 1]);
                    else { win[KT - 1][0] = win[KT - 1][1] = win[KT - 1][2] = win[KT - 1][3] = 0.f; }
                    float y[4], dh[4];
                    V4<T>::unpack(ry[u], y);
                    float h4[4], sg4[4];
#pragma unroll
                    for (int i = 0; i < 4; ++i) h4[i] = fmaf(y[i], bs[i], bt[i]);
                    sigmoid_n<4>(h4, sg4);
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const float h = h4[i], sg = sg4[i];
                        float z2 = h * sg;
                        float dz = 0.f;
#pragma unroll
                        for (int k = 0; k < KT; ++k) {
                            float gk = win[KT - 1 - k][i];
                            dz = fmaf(w[k][i], gk, dz);
                            dwacc[k][i] = fmaf(z2, gk, dwacc[k][i]);
                        }
                        dh[i] = dz * (sg * (1.0f + h * (1.0f - sg)));
                    }
                    st4<T>(dhp + e0 + t * tstride, dh);
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        float r = round_t<T>(dh[i]);
                        st0[i] += r;
                        st1[i] += r * (y[i] - bm[i]) * bi[i];
                    }
#pragma unroll
                    for (int k = 0; k < KT - 1; ++k)
#pragma unroll
                        for (int i = 0; i < 4; ++i) win[k][i] = win[k + 1][i];
                }
            }
        }
    }
    DET_WAVES_BEGIN
    if (chan_ok) {
#pragma unroll
        for (int k = 0; k < KT; ++k)
#pragma unroll
            for (int i = 0; i < 4; ++i) atomicAdd(&lw[k * CS + cv * 4 + i], dwacc[k][i]);
    }
    DET_WAVES_END
    __syncthreads();
    DET_ENTER();
    for (int i = tid; i < KT * CS; i += 256) {
        int k = i / CS, c = c0 + i % CS;
        if (c < a.C) atomicAdd(a.dw + (i64)c * KT + k, lw[i]);
    }
    if (a.stats) block_stats_flush<T>(lstat, st0, st1, cv, c0, a.C, a.stats, blockIdx.x % DWN_NREP);
    DET_EXIT();
}

// ------------------------------------------------------------------------------------------------
// launchers
// ------------------------------------------------------------------------------------------------
static inline int grid_cap(i64 work, int cap) { return (int)(work < cap ? (work > 0 ? work : 1) : cap); }
// persistent grids: one full resident wave of workgroups (256 CUs x blocks/CU from the occupancy query), so no
// ragged second wave; `slices` workgroups share each x index
template <typename K>
static int resident_grid_x(K kernel, size_t dyn_lds, int slices, i64 work, int threads = 256) {
    if (dyn_lds > 48 * 1024) {     // opt in to > default dynamic LDS (160 KiB per CU on gfx950, minus the static part)
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)dyn_lds) != hipSuccess)
            (void)hipGetLastError();   // clear: the launch itself reports a too-large tile
    }
    int bpc = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&bpc, kernel, threads, dyn_lds) != hipSuccess || bpc < 1) bpc = 2;
    int gx = (256 * bpc) / slices;     // 256 CUs
    if (gx < 1) gx = 1;
    return grid_cap(work, gx);
}

template <typename T>
static int spatial_fwd_t(DwSpatialFwd a, hipStream_t s) {
    constexpr int CS = SL<T>::CS;
    if (a.ks != 3) return dwn_set_error(-4, "dw_spatial: only spatial_kernel=3 is built");
    if (a.C % 8) return dwn_set_error(-2, "dw_spatial: C must be a multiple of 8");
    const int Wp = a.Win + 2;
    // bf16, stride 1/2: x-pair-packed tile (dot2 kernel), rows of ceil(Wp/2) pairs x 256 bytes
    const bool pair = TT<T>::IS_BF16 && (a.stride == 1 || a.stride == 2);
    auto tile_bytes = [&](int rb) {
        return pair ? (size_t)((rb - 1) * a.stride + 3) * ((Wp + 1) / 2) * 256 : (size_t)((rb - 1) * a.stride + 3) * Wp * 128;
    };
    if (a.rows_band <= 0) {
        // largest band whose tile fits the LDS budget: fewer halo rows re-read, fewer barriers per byte
        int rb = 1;
        while (rb < a.Hout && tile_bytes(rb + 1) <= (size_t)DWS_FWD_LDS_BUDGET) ++rb;
        const int nb = (a.Hout + rb - 1) / rb;          // even split: no ragged last band
        a.rows_band = (a.Hout + nb - 1) / nb;
    }
    if (a.rows_band > a.Hout) a.rows_band = a.Hout;
    const size_t lds = tile_bytes(a.rows_band);
    if (lds > 156 * 1024) return dwn_set_error(-5, "dw_spatial: plane too wide for the LDS tile");
    const int nbands = (a.Hout + a.rows_band - 1) / a.rows_band;
    const int slices = (a.C + CS - 1) / CS;
    const i64 work = (i64)a.planes * nbands;
    // 512-thread workgroups (twice the waves per LDS tile) pay off on large planes; tiny planes prefer 256
    const bool big = a.Hin * a.Win >= 512;
#define DWS_FWD_LAUNCH(ST_, NT_) do { \
        dim3 grid(resident_grid_x(dw_spatial_fwd_kernel<T, 3, ST_, NT_>, lds, slices, work, NT_), slices); \
        hipLaunchKernelGGL((dw_spatial_fwd_kernel<T, 3, ST_, NT_>), grid, dim3(NT_), lds, s, a); } while (0)
    if constexpr (TT<T>::IS_BF16) {
        if (pair) {
#define DWS_PAIR_LAUNCH(ST_, NT_) do { \
        dim3 grid(resident_grid_x(dw_spatial_fwd_pair_kernel<ST_, NT_>, lds, slices, work, NT_), slices); \
        hipLaunchKernelGGL((dw_spatial_fwd_pair_kernel<ST_, NT_>), grid, dim3(NT_), lds, s, a); } while (0)
            if (a.stride == 1) { if (big) DWS_PAIR_LAUNCH(1, 512); else DWS_PAIR_LAUNCH(1, 256); }
            else { if (big) DWS_PAIR_LAUNCH(2, 512); else DWS_PAIR_LAUNCH(2, 256); }
#undef DWS_PAIR_LAUNCH
            DWN_CHECK_LAUNCH();
            return 0;
        }
    }
    if (a.stride == 1) { if (big) DWS_FWD_LAUNCH(1, 512); else DWS_FWD_LAUNCH(1, 256); }
    else if (a.stride == 2) { if (big) DWS_FWD_LAUNCH(2, 512); else DWS_FWD_LAUNCH(2, 256); }
    else DWS_FWD_LAUNCH(0, 256);
#undef DWS_FWD_LAUNCH
    DWN_CHECK_LAUNCH();
    return 0;
}
bool dw_spatial_fwd_walk_supported(const DwSpatialFwd& a, int dtype);
int launch_dw_spatial_fwd_walk(const DwSpatialFwd& a, hipStream_t s);
int launch_dw_spatial_fwd(const DwSpatialFwd& a, int dtype, hipStream_t s) {
    if (a.a0 && !dw_spatial_fwd_walk_supported(a, dtype))
        return dwn_set_error(-3, "dw_spatial_fwd: rebuilt-input mode (a0 != NULL) is built into the chained row-walk kernels only (dwn_dw_spatial_fwd_rc_supported)");
    if (dw_spatial_fwd_walk_supported(a, dtype)) return launch_dw_spatial_fwd_walk(a, s);     // row-walk kernels (dwn_dwfwd.hip)
    return dtype == DWN_BF16 ? spatial_fwd_t<bf16_t>(a, s) : spatial_fwd_t<float>(a, s);
}

// ------------------------------------------------------------------------------------------------
// spatial backward, bf16 storage, 3x3, stride 1: x-pair-packed gradient tile + v_dot2c_f32_bf16.
//
// The staged gradient g = dL/dy2 (BatchNorm-backward affine applied, zero halo; column index xq = wo + 1) is
// packed as G[k] = (g[2k], g[2k+1]) per channel.  A thread owns the input pixel pair (2j, 2j+1) of a row:
//   dz[2j]   = dot2(G[j], (w2,w1)) + dot2(G[j+1], (w0,0))         dz[2j+1] = dot2(G[j], (0,w2)) + dot2(G[j+1], (w1,w0))
//   dW[.,2] += dot2(Z, G[j])      dW[.,0] += dot2(Z, G[j+1])      dW[.,1] += dot2(Z, (G[j].hi, G[j+1].lo))
// with Z = (z1[2j], z1[2j+1]) rounded to bf16 (the reference's autocast stores z1 in bf16): 8 VALU ops per stencil row,
// channel and pixel pair instead of 6 unpacks + 6 packed FMAs.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(DWS_BWD_THREADS, DWS_BWD_PAIR_MINW) void dw_spatial_bwd_pair_kernel(const DwSpatialBwd a) {
    typedef bf16_t T;
    constexpr int NT = DWS_BWD_THREADS, NCV = 16, CS = 64, LP = NT / NCV, KS = 3, P = 1;
    __shared__ float lstat[2 * CS];
    __shared__ __attribute__((aligned(16))) float lw[KS * KS * CS];          // dW reduction at the end
    __shared__ __attribute__((aligned(16))) unsigned lwp[3 * 4 * CS];        // packed weights [dy][combo][channel]
    const int tid = threadIdx.x;
    const int cv = tid % NCV, pl = tid / NCV;
    const int c0 = blockIdx.y * CS;
    const int chan = c0 + cv * 4;
    const bool chan_ok = chan < a.C;
    const int chs = chan_ok ? chan : 0;
    if (tid < 2 * CS) lstat[tid] = 0.f;
    for (int i = tid; i < 3 * CS; i += NT) {
        const int dy = i / CS, cc = i % CS, c = c0 + cc;
        float w0 = 0.f, w1 = 0.f, w2 = 0.f;
        if (c < a.C) { w0 = a.w[(i64)(dy * 3 + 0) * a.C + c]; w1 = a.w[(i64)(dy * 3 + 1) * a.C + c]; w2 = a.w[(i64)(dy * 3 + 2) * a.C + c]; }
        lwp[(dy * 4 + 0) * CS + cc] = pack_bf16x2(w2, w1);
        lwp[(dy * 4 + 1) * CS + cc] = pack_bf16x2(w0, 0.f);
        lwp[(dy * 4 + 2) * CS + cc] = pack_bf16x2(0.f, w2);
        lwp[(dy * 4 + 3) * CS + cc] = pack_bf16x2(w1, w0);
    }
    __syncthreads();

    float dwp[KS * KS][4];
#pragma unroll
    for (int k = 0; k < KS * KS; ++k) { dwp[k][0] = dwp[k][1] = dwp[k][2] = dwp[k][3] = 0.f; }
    float bs[4], bt[4], bm[4], bi[4];
    ldc4(a.y1.v1 + chs, bs); ldc4(a.y1.v2 + chs, bt); ldc4(a.y1.v3 + chs, bm); ldc4(a.y1.v4 + chs, bi);
    const f2_t bs2[2] = {f2_t{bs[0], bs[1]}, f2_t{bs[2], bs[3]}}, bt2[2] = {f2_t{bt[0], bt[1]}, f2_t{bt[2], bt[3]}};
    const f2_t bi2[2] = {f2_t{bi[0], bi[1]}, f2_t{bi[2], bi[3]}};
    const f2_t nbm2[2] = {f2_t{-bm[0] * bi[0], -bm[1] * bi[1]}, f2_t{-bm[2] * bi[2], -bm[3] * bi[3]}};   // yhat = y*bi + nbm
    f2_t sp0[2] = {f2_t{0.f, 0.f}, f2_t{0.f, 0.f}}, sp1[2] = {f2_t{0.f, 0.f}, f2_t{0.f, 0.f}};

    const int Wq = a.Wout + 2;                       // staged columns wo = -1 .. Wout
    const int Wqp = (Wq + 1) >> 1;                   // ... as pairs
    const int Wip = (a.Win + 1) >> 1;                // input pixel pairs per row
    const FastDiv dvq(Wqp), dvw(Wip);
    const int nbands = (a.Hin + a.rows_band - 1) / a.rows_band;
    const int ntiles = a.planes * nbands;
    T* dhp = reinterpret_cast<T*>(a.dh1);
    const T* y1p = reinterpret_cast<const T*>(a.y1.p);
    const T* dpp = reinterpret_cast<const T*>(a.dy.p);
    const T* dqp = reinterpret_cast<const T*>(a.dy.q);
    unsigned* tile = reinterpret_cast<unsigned*>(dyn_smem);      // [rows_q][Wqp][64] dwords
    const int rowdw = Wqp * CS;
    for (int tile_id = blockIdx.x; tile_id < ntiles; tile_id += gridDim.x) {
        const int plane = tile_id / nbands, band = tile_id % nbands;
        const int hi0 = band * a.rows_band;
        const int nri = (a.Hin - hi0 < a.rows_band) ? a.Hin - hi0 : a.rows_band;
        const int ho_lo = hi0 - 1;                   // stride 1: rows ho = hi - 1 .. hi + 1
        const int rows_q = nri + 2;
        const i64 orow0 = (i64)plane * a.Hout * a.Wout;
        {
            constexpr int NB = 4;
            const int total_st = rows_q * Wqp;
            float a1[4], a2[4], a3[4];
            ldc4(a.dy.v1 + chs, a1); ldc4(a.dy.v2 + chs, a2); ldc4(a.dy.v3 + chs, a3);
            const f2_t a1v[2] = {f2_t{a1[0], a1[1]}, f2_t{a1[2], a1[3]}}, a2v[2] = {f2_t{a2[0], a2[1]}, f2_t{a2[2], a2[3]}};
            const f2_t a3v[2] = {f2_t{a3[0], a3[1]}, f2_t{a3[2], a3[3]}};
            const T* dp0 = dpp + orow0 * a.dy.ld + chs;
            const T* dq0 = dqp + orow0 * a.dy.ld + chs;
            for (int f0 = pl; f0 < total_st; f0 += NB * LP) {
                uint2 rp[NB][2], rq[NB][2];
                unsigned msk[NB];
#pragma unroll
                for (int u = 0; u < NB; ++u) {
                    const int f = f0 + u * LP;
                    const int r = dvq.div(f);
                    const int xp = dvq.rem(f, r);
                    const int ho = ho_lo + r, wo0 = 2 * xp - 1;
                    const bool rok = chan_ok && f < total_st && (unsigned)ho < (unsigned)a.Hout;
                    const bool ok0 = rok && (unsigned)wo0 < (unsigned)a.Wout, ok1 = rok && (unsigned)(wo0 + 1) < (unsigned)a.Wout;
                    const int rowpix = __mul24(ho, a.Wout);
                    const unsigned off0 = ok0 ? __umul24((unsigned)(rowpix + wo0), (unsigned)a.dy.ld) : 0u;
                    const unsigned off1 = ok1 ? __umul24((unsigned)(rowpix + wo0 + 1), (unsigned)a.dy.ld) : 0u;
                    rp[u][0] = ld4_raw<T>(dp0 + off0); rq[u][0] = ld4_raw<T>(dq0 + off0);
                    rp[u][1] = ld4_raw<T>(dp0 + off1); rq[u][1] = ld4_raw<T>(dq0 + off1);
                    msk[u] = (ok0 ? 0x0000ffffu : 0u) | (ok1 ? 0xffff0000u : 0u);
                }
#pragma unroll
                for (int u = 0; u < NB; ++u) {
                    const int f = f0 + u * LP;
                    if (f < total_st) {
                        f2_t g[2][2];
#pragma unroll
                        for (int h = 0; h < 2; ++h) {
                            f2_t p0, p1, q0, q1;
                            unpack_pairs<T>(rp[u][h], p0, p1);
                            unpack_pairs<T>(rq[u][h], q0, q1);
                            g[h][0] = a1v[0] * p0 + (a2v[0] * q0 + a3v[0]);
                            g[h][1] = a1v[1] * p1 + (a2v[1] * q1 + a3v[1]);
                        }
                        const uint4 o = make_uint4(pack_bf16x2(g[0][0].x, g[1][0].x) & msk[u], pack_bf16x2(g[0][0].y, g[1][0].y) & msk[u],
                                                   pack_bf16x2(g[0][1].x, g[1][1].x) & msk[u], pack_bf16x2(g[0][1].y, g[1][1].y) & msk[u]);
                        *reinterpret_cast<uint4*>(tile + f * CS + cv * 4) = o;
                    }
                }
            }
        }
        __syncthreads();
        {
            const i64 prow0 = (i64)plane * a.Hin * a.Win;
            T* dh0 = dhp + prow0 * a.C + chan;
            const T* y10 = y1p + prow0 * a.y1.ld + chs;
            const int total = nri * Wip;
            const int grow0 = hi0 * a.Win;
            constexpr int XB = DWS_BWD_PAIR_XB;      // pixel pairs in flight per thread
            for (int i0 = pl; i0 < total; i0 += XB * LP) {
                uint2 ry[XB][2];
                int jj[XB], iyv[XB];
                bool v1[XB];
#pragma unroll
                for (int u = 0; u < XB; ++u) {
                    const int i = i0 + u * LP;
                    const bool ok = chan_ok && i < total;
                    const int iv = ok ? i : 0;
                    iyv[u] = dvw.div(iv);
                    jj[u] = dvw.rem(iv, iyv[u]);
                    const int pix = grow0 + __mul24(iyv[u], a.Win) + 2 * jj[u];
                    v1[u] = 2 * jj[u] + 1 < a.Win;
                    ry[u][0] = ld4_raw<T>(y10 + __umul24((unsigned)pix, (unsigned)a.y1.ld));
                    ry[u][1] = ld4_raw<T>(y10 + __umul24((unsigned)(pix + (v1[u] ? 1 : 0)), (unsigned)a.y1.ld));
                }
#pragma unroll
                for (int u = 0; u < XB; ++u) {
                    if (!chan_ok || i0 + u * LP >= total) continue;
                    f2_t y[2][2], z1[2][2], dsl[2][2];
#pragma unroll
                    for (int h = 0; h < 2; ++h) {
                        unpack_pairs<T>(ry[u][h], y[h][0], y[h][1]);
#pragma unroll
                        for (int q = 0; q < 2; ++q) {
                            const f2_t hh = y[h][q] * bs2[q] + bt2[q];
                            const f2_t sg = sigmoid2f_(hh);
                            z1[h][q] = hh * sg;
                            dsl[h][q] = sg * (1.0f + hh * (1.0f - sg));
                        }
                    }
                    if (!v1[u]) { z1[1][0] = f2_t{0.f, 0.f}; z1[1][1] = f2_t{0.f, 0.f}; }
                    const unsigned Z[4] = {pack_bf16x2(z1[0][0].x, z1[1][0].x), pack_bf16x2(z1[0][0].y, z1[1][0].y),
                                           pack_bf16x2(z1[0][1].x, z1[1][1].x), pack_bf16x2(z1[0][1].y, z1[1][1].y)};
                    float dz0[4] = {0.f, 0.f, 0.f, 0.f}, dz1[4] = {0.f, 0.f, 0.f, 0.f};
                    const unsigned* tp = tile + (__mul24(iyv[u], Wqp) + jj[u]) * CS + cv * 4;
#pragma unroll
                    for (int dy = 0; dy < 3; ++dy) {
                        // tile row iy + (2 - dy): output row ho = hi + 1 - dy
                        const uint4 G0 = *reinterpret_cast<const uint4*>(tp + (2 - dy) * rowdw);
                        const uint4 G1 = *reinterpret_cast<const uint4*>(tp + (2 - dy) * rowdw + CS);
                        const uint4 Wa = *reinterpret_cast<const uint4*>(&lwp[(dy * 4 + 0) * CS + cv * 4]);
                        const uint4 Wb = *reinterpret_cast<const uint4*>(&lwp[(dy * 4 + 1) * CS + cv * 4]);
                        const uint4 Wc = *reinterpret_cast<const uint4*>(&lwp[(dy * 4 + 2) * CS + cv * 4]);
                        const uint4 Wd = *reinterpret_cast<const uint4*>(&lwp[(dy * 4 + 3) * CS + cv * 4]);
                        const unsigned g0[4] = {G0.x, G0.y, G0.z, G0.w}, g1[4] = {G1.x, G1.y, G1.z, G1.w};
                        const unsigned wa[4] = {Wa.x, Wa.y, Wa.z, Wa.w}, wb[4] = {Wb.x, Wb.y, Wb.z, Wb.w};
                        const unsigned wc[4] = {Wc.x, Wc.y, Wc.z, Wc.w}, wd[4] = {Wd.x, Wd.y, Wd.z, Wd.w};
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            dz0[q] = dot2_bf16(g0[q], wa[q], dz0[q]);
                            dz0[q] = dot2_bf16(g1[q], wb[q], dz0[q]);
                            dz1[q] = dot2_bf16(g0[q], wc[q], dz1[q]);
                            dz1[q] = dot2_bf16(g1[q], wd[q], dz1[q]);
                            const unsigned gm = __builtin_amdgcn_alignbit(g1[q], g0[q], 16);      // (G0.hi, G1.lo)
                            dwp[dy * 3 + 2][q] = dot2_bf16(Z[q], g0[q], dwp[dy * 3 + 2][q]);
                            dwp[dy * 3 + 1][q] = dot2_bf16(Z[q], gm, dwp[dy * 3 + 1][q]);
                            dwp[dy * 3 + 0][q] = dot2_bf16(Z[q], g1[q], dwp[dy * 3 + 0][q]);
                        }
                    }
                    const int pix = grow0 + __mul24(iyv[u], a.Win) + 2 * jj[u];
#pragma unroll
                    for (int h = 0; h < 2; ++h) {
                        if (h == 1 && !v1[u]) break;
                        const float* dz = h == 0 ? dz0 : dz1;
                        const f2_t d0 = f2_t{dz[0], dz[1]} * dsl[h][0], d1 = f2_t{dz[2], dz[3]} * dsl[h][1];
                        const uint2 packed = make_uint2(pack_bf16x2(d0.x, d0.y), pack_bf16x2(d1.x, d1.y));
                        *reinterpret_cast<uint2*>(dh0 + __umul24((unsigned)(pix + h), (unsigned)a.C)) = packed;
                        f2_t r0, r1;
                        unpack_pairs<T>(packed, r0, r1);
                        sp0[0] += r0; sp0[1] += r1;
                        sp1[0] += r0 * (y[h][0] * bi2[0] + nbm2[0]);
                        sp1[1] += r1 * (y[h][1] * bi2[1] + nbm2[1]);
                    }
                }
            }
        }
        __syncthreads();
    }
    // weight gradient: reduce over the threads sharing a channel vector through LDS, then global fp32 atomics
    for (int i = tid; i < KS * KS * CS; i += NT) lw[i] = 0.f;
    __syncthreads();
    DET_WAVES_BEGIN
    if (chan_ok) {
#pragma unroll
        for (int k = 0; k < KS * KS; ++k)
#pragma unroll
            for (int q = 0; q < 4; ++q) atomicAdd(&lw[k * CS + cv * 4 + q], dwp[k][q]);
    }
    DET_WAVES_END
    __syncthreads();
    DET_ENTER();
    for (int i = tid; i < KS * KS * CS; i += NT) {
        const int k = i / CS, c = c0 + i % CS;
        if (c < a.C) atomicAdd(&a.dw[(i64)c * (KS * KS) + k], lw[i]);
    }
    if (a.stats) {
        const float s0[4] = {sp0[0].x, sp0[0].y, sp0[1].x, sp0[1].y}, s1[4] = {sp1[0].x, sp1[0].y, sp1[1].x, sp1[1].y};
        block_stats_flush<T>(lstat, s0, s1, cv, c0, a.C, a.stats, blockIdx.x % DWN_NREP);
    }
    DET_EXIT();
}

template <typename T>
static int spatial_bwd_t(DwSpatialBwd a, hipStream_t s) {
    constexpr int CS = SL<T>::CS;
    if (a.ks != 3) return dwn_set_error(-4, "dw_spatial: only spatial_kernel=3 is built");
    if (a.C % 8) return dwn_set_error(-2, "dw_spatial: C must be a multiple of 8");
    const int Wq = a.Wout + 2;
    auto rows_q = [&](int rb) { return (rb - 1 + 2) / a.stride + 2; };   // upper bound on staged output rows
    // bf16, stride 1: x-pair-packed gradient tile (dot2 kernel), rows of ceil(Wq/2) pairs x 256 bytes
    const bool pair = TT<T>::IS_BF16 && a.stride == 1;
    const size_t row_bytes = pair ? (size_t)((Wq + 1) / 2) * 256 : (size_t)Wq * 128;
    if (a.rows_band <= 0) {
        int rb = 1;
        while (rb < a.Hin && (size_t)rows_q(rb + 1) * row_bytes <= (size_t)DWS_BWD_LDS_BUDGET) ++rb;
        const int nb = (a.Hin + rb - 1) / rb;
        a.rows_band = (a.Hin + nb - 1) / nb;
    }
    if (a.rows_band > a.Hin) a.rows_band = a.Hin;
    const size_t lds = (size_t)rows_q(a.rows_band) * row_bytes;
    if (lds > 150 * 1024) return dwn_set_error(-5, "dw_spatial_bwd: plane too wide for the LDS tile");
    const int nbands = (a.Hin + a.rows_band - 1) / a.rows_band;
    const int slices = (a.C + CS - 1) / CS;
    const i64 work = (i64)a.planes * nbands;
    if (pair) { dim3 grid(resident_grid_x(dw_spatial_bwd_pair_kernel, lds, slices, work, DWS_BWD_THREADS), slices); hipLaunchKernelGGL(dw_spatial_bwd_pair_kernel, grid, dim3(DWS_BWD_THREADS), lds, s, a); }
    else if (a.stride == 1) { dim3 grid(resident_grid_x(dw_spatial_bwd_kernel<T, 3, 1>, lds, slices, work, DWS_BWD_THREADS), slices); hipLaunchKernelGGL((dw_spatial_bwd_kernel<T, 3, 1>), grid, dim3(DWS_BWD_THREADS), lds, s, a); }
    else if (a.stride == 2) { dim3 grid(resident_grid_x(dw_spatial_bwd_kernel<T, 3, 2>, lds, slices, work, DWS_BWD_THREADS), slices); hipLaunchKernelGGL((dw_spatial_bwd_kernel<T, 3, 2>), grid, dim3(DWS_BWD_THREADS), lds, s, a); }
    else { dim3 grid(resident_grid_x(dw_spatial_bwd_kernel<T, 3, 0>, lds, slices, work, DWS_BWD_THREADS), slices); hipLaunchKernelGGL((dw_spatial_bwd_kernel<T, 3, 0>), grid, dim3(DWS_BWD_THREADS), lds, s, a); }
    DWN_CHECK_LAUNCH();
    return 0;
}
bool dw_spatial_bwd_walk_supported(const DwSpatialBwd& a, int dtype);
int launch_dw_spatial_bwd_walk(const DwSpatialBwd& a, hipStream_t s);
int launch_dw_spatial_bwd(const DwSpatialBwd& a, int dtype, hipStream_t s) {
    if (a.a0 && !dw_spatial_bwd_walk_supported(a, dtype))
        return dwn_set_error(-3, "dw_spatial_bwd: rebuilt-y1 mode (a0 != NULL) is built into the row-walk kernels only (dwn_dw_spatial_bwd_rc_supported)");
    if (dw_spatial_bwd_walk_supported(a, dtype)) return launch_dw_spatial_bwd_walk(a, s);     // row-walk kernels (dwn_dwbwd.hip)
    return dtype == DWN_BF16 ? spatial_bwd_t<bf16_t>(a, s) : spatial_bwd_t<float>(a, s);
}

template <typename T>
static int temporal_fwd_t(const DwTemporalFwd& a, hipStream_t s) {
    constexpr int CS = SL<T>::CS, LP = SL<T>::LP;
    if (a.C % 8) return dwn_set_error(-2, "dw_temporal: C must be a multiple of 8");
    const int slices = (a.C + CS - 1) / CS;
    const i64 npos = (i64)a.B * a.HW;
    const i64 work = (npos + LP - 1) / LP;
    if (a.z_scale) {
        if (!a.z_shift || a.stats) return dwn_set_error(-2, "dw_temporal: the z3 epilogue needs z_shift and no statistics (eval mode)");
        if (a.kt == 5) { dim3 grid(resident_grid_x(dw_temporal_fwd_kernel<T, 5, true>, 0, slices, work), slices); hipLaunchKernelGGL((dw_temporal_fwd_kernel<T, 5, true>), grid, dim3(256), 0, s, a); }
        else if (a.kt == 3) { dim3 grid(resident_grid_x(dw_temporal_fwd_kernel<T, 3, true>, 0, slices, work), slices); hipLaunchKernelGGL((dw_temporal_fwd_kernel<T, 3, true>), grid, dim3(256), 0, s, a); }
        else return dwn_set_error(-4, "dw_temporal: only temporal_kernel 3 or 5 is built");
        DWN_CHECK_LAUNCH();
        return 0;
    }
    if (a.kt == 5) { dim3 grid(resident_grid_x(dw_temporal_fwd_kernel<T, 5>, 0, slices, work), slices); hipLaunchKernelGGL((dw_temporal_fwd_kernel<T, 5>), grid, dim3(256), 0, s, a); }
    else if (a.kt == 3) { dim3 grid(resident_grid_x(dw_temporal_fwd_kernel<T, 3>, 0, slices, work), slices); hipLaunchKernelGGL((dw_temporal_fwd_kernel<T, 3>), grid, dim3(256), 0, s, a); }
    else return dwn_set_error(-4, "dw_temporal: only temporal_kernel 3 or 5 is built");
    DWN_CHECK_LAUNCH();
    return 0;
}
int launch_dw_temporal_fwd(const DwTemporalFwd& a, int dtype, hipStream_t s) {
    return dtype == DWN_BF16 ? temporal_fwd_t<bf16_t>(a, s) : temporal_fwd_t<float>(a, s);
}

// ------------------------------------------------------------------------------------------------
// temporal backward without reading y3 (dy_kind == LD_PLAIN: dy.p = dh3, dy.v1..v3 = bn3-backward A1, A2, A3).
// dy3 = A1*dh3 + A2*y3 + A3 needs y3 = sum_j w[j] z2[t + j - P], and z2 = SiLU(BN2(y2)) is computed here anyway (for dW
// and SiLU'): y3 is recomputed from a (2P+1)-deep window of z2 that runs 2P frames ahead of the output frame.
// Three passes over the E-wide tensors (dh3, y2 in; dh2 out) instead of four; 5 extra FMAs per element.
// ------------------------------------------------------------------------------------------------
template <typename T, int KT>
__global__ __launch_bounds__(256, DWT_RC_MINW) void dw_temporal_bwd_rc_kernel(const DwTemporalBwd a) {
    constexpr int NCV = SL<T>::NCV, CS = SL<T>::CS, LP = SL<T>::LP, P = KT / 2, NW = 2 * P + 1;
    constexpr int TB = NW;            // frames per unrolled batch == window length: ring indices are compile-time, no shifting
    static_assert(NW == KT, "ring rotation below assumes an odd kernel: window length == kernel length");
    typedef typename SL<T>::raw_t raw_t;
    __shared__ float lstat[2 * CS];
    __shared__ float lw[KT * CS];
    const int tid = threadIdx.x;
    const int cv = tid % NCV, pl = tid / NCV;
    const int c0 = blockIdx.y * CS;
    const int chan = c0 + cv * 4;
    const bool chan_ok = chan < a.C;
    const int chs = chan_ok ? chan : 0;
    if (tid < 2 * CS) lstat[tid] = 0.f;
    for (int i = tid; i < KT * CS; i += 256) lw[i] = 0.f;
    __syncthreads();
    float w[KT][4], dwacc[KT][4];
#pragma unroll
    for (int k = 0; k < KT; ++k) {
        ldc4(a.w + (i64)k * a.C + chs, w[k]);
        dwacc[k][0] = dwacc[k][1] = dwacc[k][2] = dwacc[k][3] = 0.f;
    }
    float bs[4], bt[4], bm[4], bi[4], a1[4], a2[4], a3[4];
    ldc4(a.y2.v1 + chs, bs); ldc4(a.y2.v2 + chs, bt); ldc4(a.y2.v3 + chs, bm); ldc4(a.y2.v4 + chs, bi);
    ldc4(a.dy.v1 + chs, a1); ldc4(a.dy.v2 + chs, a2); ldc4(a.dy.v3 + chs, a3);
    float st0[4] = {0.f, 0.f, 0.f, 0.f}, st1[4] = {0.f, 0.f, 0.f, 0.f};

    const i64 npos = (i64)a.B * a.HW;
    T* dhp = reinterpret_cast<T*>(a.dh2);
    const T* y2p = reinterpret_cast<const T*>(a.y2.p);
    const T* dpp = reinterpret_cast<const T*>(a.dy.p);
    const i64 tstride = (i64)a.HW * a.C;

    if (chan_ok) {
        for (i64 pos = (i64)blockIdx.x * LP + pl; pos < npos; pos += (i64)gridDim.x * LP) {
            const i64 b = pos / a.HW, hw = pos % a.HW;
            const i64 e0 = (b * a.T * a.HW + hw) * a.C + chan;
            // rings of NW = KT slots: at unrolled step u of a batch, frame t + j of (y2, z2, SiLU') lives in slot
            // (u + j) % NW and dy3(t + k - P) in win[(u + k) % KT].  Frames outside [0, T) are zero slots.
            float yw[NW][4], zw[NW][4], dsw[NW][4], win[KT][4];
#pragma unroll
            for (int j = 0; j < NW; ++j)
#pragma unroll
                for (int i = 0; i < 4; ++i) { yw[j][i] = 0.f; zw[j][i] = 0.f; dsw[j][i] = 0.f; }
#pragma unroll
            for (int k = 0; k < KT; ++k) win[k][0] = win[k][1] = win[k][2] = win[k][3] = 0.f;
            auto activate = [&](const raw_t& raw, float* y, float* z, float* ds) {
                V4<T>::unpack(raw, y);
                float h4[4], sg4[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) h4[i] = fmaf(y[i], bs[i], bt[i]);
                sigmoid_n<4>(h4, sg4);
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const float h = h4[i], sg = sg4[i];
                    z[i] = h * sg;
                    ds[i] = sg * (1.0f + h * (1.0f - sg));
                }
            };
            // at virtual frame t = -P the window covers frames -P .. P-1: load frames 0 .. P-1 into slots P .. 2P-1
#pragma unroll
            for (int j = 0; j < P; ++j)
                if (j < a.T) activate(ld4_raw<T>(y2p + e0 + j * tstride), yw[P + j], zw[P + j], dsw[P + j]);
            for (int t0 = -P; t0 < a.T; t0 += TB) {
                raw_t rp[TB], ry[TB];
                // load issue at raised priority: a streaming kernel's next loads should leave before the other waves' arithmetic gets
                // the issue slot (measured: 240 -> 227 us per launch in the step; the forward kernel and se_pool, whose vector ALUs are half as busy, gain nothing)
                __builtin_amdgcn_s_setprio(DWT_PRIO);
#pragma unroll
                for (int u = 0; u < TB; ++u) {
                    const int td = t0 + u + P, ty = t0 + u + 2 * P;          // frames of dh3 / y2 fetched for output frame t0+u
                    rp[u] = ld4_raw<T>(dpp + e0 + (td < a.T ? td : 0) * tstride);
                    ry[u] = ld4_raw<T>(y2p + e0 + (ty < a.T ? ty : 0) * tstride);
                }
                __builtin_amdgcn_s_setprio(0);
#pragma unroll
                for (int u = 0; u < TB; ++u) {
                    const int t = t0 + u;
                    if (t >= a.T) break;
                    // newest window slot: frame t + 2P
                    const int sn = (u + NW - 1) % NW;                   // newest slot (frame t + 2P)
                    if (t + 2 * P < a.T) activate(ry[u], yw[sn], zw[sn], dsw[sn]);
                    else { for (int i = 0; i < 4; ++i) { yw[sn][i] = 0.f; zw[sn][i] = 0.f; dsw[sn][i] = 0.f; } }
                    const int s0 = u % NW;                              // slot of the output frame t
                    // dy3(t + P) from dh3 and the recomputed y3(t + P) = sum_j w[j] z2(t + j)
                    if (t + P < a.T) {
                        float p[4];
                        V4<T>::unpack(rp[u], p);
#pragma unroll
                        for (int i = 0; i < 4; ++i) {
                            float y3 = 0.f;
#pragma unroll
                            for (int j = 0; j < KT; ++j) y3 = fmaf(w[j][i], zw[(u + j) % NW][i], y3);
                            win[sn][i] = fmaf(a1[i], p[i], fmaf(a2[i], round_t<T>(y3), a3[i]));
                        }
                    } else { win[sn][0] = win[sn][1] = win[sn][2] = win[sn][3] = 0.f; }
                    if (t >= 0) {
                        float dh[4];
#pragma unroll
                        for (int i = 0; i < 4; ++i) {
                            float dz = 0.f;
#pragma unroll
                            for (int k = 0; k < KT; ++k) {
                                const float gk = win[(u + KT - 1 - k) % KT][i];
                                dz = fmaf(w[k][i], gk, dz);
                                dwacc[k][i] = fmaf(zw[s0][i], gk, dwacc[k][i]);
                            }
                            dh[i] = dz * dsw[s0][i];
                        }
                        st4<T>(dhp + e0 + t * tstride, dh);
#pragma unroll
                        for (int i = 0; i < 4; ++i) {
                            const float r = round_t<T>(dh[i]);
                            st0[i] += r;
                            st1[i] = fmaf(r, yw[s0][i], st1[i]);          // raw sum(dh2 * y2): normalised once, below
                        }
                    }
                }
            }
        }
    }
    // sum(dh2 * (y2 - mean) * invstd) = invstd * (sum(dh2 * y2) - mean * sum(dh2)): two VALU operations per element less in
    // a kernel that is VALU-bound (as in the spatial backward kernels and the dh3 GEMM epilogue)
#pragma unroll
    for (int i = 0; i < 4; ++i) st1[i] = bi[i] * fmaf(-bm[i], st0[i], st1[i]);
    DET_WAVES_BEGIN
    if (chan_ok) {
#pragma unroll
        for (int k = 0; k < KT; ++k)
#pragma unroll
            for (int i = 0; i < 4; ++i) atomicAdd(&lw[k * CS + cv * 4 + i], dwacc[k][i]);
    }
    DET_WAVES_END
    __syncthreads();
    DET_ENTER();
    for (int i = tid; i < KT * CS; i += 256) {
        int k = i / CS, c = c0 + i % CS;
        if (c < a.C) atomicAdd(a.dw + (i64)c * KT + k, lw[i]);
    }
    if (a.stats) block_stats_flush<T>(lstat, st0, st1, cv, c0, a.C, a.stats, blockIdx.x % DWN_NREP);
    DET_EXIT();
}

template <typename T>
static int temporal_bwd_t(const DwTemporalBwd& a, hipStream_t s) {
    constexpr int CS = SL<T>::CS, LP = SL<T>::LP;
    if (a.C % 8) return dwn_set_error(-2, "dw_temporal: C must be a multiple of 8");
    const int slices = (a.C + CS - 1) / CS;
    const i64 npos = (i64)a.B * a.HW;
    const i64 work = (npos + LP - 1) / LP;
    const bool dy3 = a.dy_kind == LD_DY3;
    if (a.dy_kind == LD_PLAIN) {       // dy.p = dh3 only: y3 is recomputed from y2 (three passes instead of four)
        if (a.kt == 5) {
            dim3 grid(resident_grid_x(dw_temporal_bwd_rc_kernel<T, 5>, 0, slices, work), slices);
            hipLaunchKernelGGL((dw_temporal_bwd_rc_kernel<T, 5>), grid, dim3(256), 0, s, a);
        } else if (a.kt == 3) {
            dim3 grid(resident_grid_x(dw_temporal_bwd_rc_kernel<T, 3>, 0, slices, work), slices);
            hipLaunchKernelGGL((dw_temporal_bwd_rc_kernel<T, 3>), grid, dim3(256), 0, s, a);
        } else return dwn_set_error(-4, "dw_temporal: only temporal_kernel 3 or 5 is built");
        DWN_CHECK_LAUNCH();
        return 0;
    }
    dim3 grid(resident_grid_x(dw_temporal_bwd_kernel<T, 5, LD_AFFINE2, 4>, 0, slices, work), slices);
    if (!dy3 && a.dy_kind != LD_AFFINE2) return dwn_set_error(-3, "dw_temporal_bwd: unsupported dy loader");
    constexpr bool tb8 = DWT_BWD_TB == 8;
    if (a.kt == 5) {
        if (dy3) hipLaunchKernelGGL((dw_temporal_bwd_kernel<T, 5, LD_DY3, 4>), grid, dim3(256), 0, s, a);
        else if (tb8) hipLaunchKernelGGL((dw_temporal_bwd_kernel<T, 5, LD_AFFINE2, 8>), grid, dim3(256), 0, s, a);
        else hipLaunchKernelGGL((dw_temporal_bwd_kernel<T, 5, LD_AFFINE2, 4>), grid, dim3(256), 0, s, a);
    } else if (a.kt == 3) {
        if (dy3) hipLaunchKernelGGL((dw_temporal_bwd_kernel<T, 3, LD_DY3, 4>), grid, dim3(256), 0, s, a);
        else hipLaunchKernelGGL((dw_temporal_bwd_kernel<T, 3, LD_AFFINE2, 4>), grid, dim3(256), 0, s, a);
    } else return dwn_set_error(-4, "dw_temporal: only temporal_kernel 3 or 5 is built");
    DWN_CHECK_LAUNCH();
    return 0;
}
int launch_dw_temporal_bwd(const DwTemporalBwd& a, int dtype, hipStream_t s) {
    return dtype == DWN_BF16 ? temporal_bwd_t<bf16_t>(a, s) : temporal_bwd_t<float>(a, s);
}
