// spat_covn_dw without a materialised conv_pw output (reference: src/models/dwiseneuro.py:90-102).
//
// y1 = a0 . W1^T (the expand conv's output, E = 7 Cin channels at input resolution) is the widest tensor of a block,
// yet it is a Cin-deep product of a tensor seven times narrower.  These kernels never read it from HBM: a workgroup
// owns a (plane, row band) tile, keeps the a0 tile in LDS and, for one 64-channel slice of E after another,
//   phase A  rebuilds the slice of y1 with v_mfma_f32_16x16x32_bf16 (weights = A operand, pixels = B operand, so a
//            lane's 4 accumulator registers are 4 consecutive channels of one pixel), applies BatchNorm-1 + SiLU on the
//            accumulators and writes the x-pair-packed stencil tile (tile[row][xp][c] = (z[2xp-1], z[2xp]) as one dword);
//   phase B  runs the 3x3 stencil on that tile with v_dot2c_f32_bf16 (as dw_spatial_fwd_pair_kernel does), stores y2
//            and accumulates the BatchNorm-2 sums.
// The two MFMA pixel tiles of a pair tile are the odd-x and the even-x pixels ("E" / "O" sub-tiles of the a0 tile),
// so a lane holds both halves of a pixel pair and packs them with one v_cvt_pk_bf16_f32 — no cross-lane traffic.
// Everything a slice needs besides a0 (its 64 W1 rows, the packed stencil weights, BN-1 scale / shift) is one
// contiguous "blob" per slice, laid out as the LDS image and copied by LDS-DMA (global_load_lds_dwordx4) one slice
// ahead; the a0 tile of the next tile is DMA'd while the last slice's stencil runs.  LDS images are bank-swizzled on
// the DMA *source* side (the DMA destination is lane-linear).
//
// HBM traffic: a0 once per tile (+ band halo rows) and y2 once — instead of (M_in + M_out) E-wide rows.
#include "dwn_internal.h"
#include <stdlib.h>

extern __shared__ __attribute__((aligned(16))) unsigned char rc_smem[];

typedef __attribute__((ext_vector_type(8))) short rc_bf16x8_t;
typedef __attribute__((ext_vector_type(4))) float rc_f32x4_t;
typedef __attribute__((ext_vector_type(2))) __bf16 rc_bf16x2_t;
typedef float rc_f2_t __attribute__((ext_vector_type(2)));
typedef unsigned rc_u32x2_t __attribute__((ext_vector_type(2)));

__device__ __forceinline__ float rc_dot2(unsigned a, unsigned b, float c) {
    return __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(rc_bf16x2_t, a), __builtin_bit_cast(rc_bf16x2_t, b), c, false);
}
__device__ __forceinline__ unsigned rc_pack2(float lo, float hi) { return pk_bf16(lo, hi); }
__device__ __forceinline__ rc_f32x4_t rc_mfma(const uint4& a, const uint4& b, const rc_f32x4_t& c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(rc_bf16x8_t, a), __builtin_bit_cast(rc_bf16x8_t, b), c, 0, 0, 0);
}
// LDS-DMA: 64 lanes x 16 bytes land at lds_dst + 16*lane (lds_dst wave-uniform); the source address is per lane.
// Invisible to the compiler's s_waitcnt bookkeeping: completion is waited for with rc_wait_vm0().
__device__ __forceinline__ void rc_glds16(const void* gsrc, unsigned lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}
__device__ __forceinline__ void rc_wait_vm0() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
// workgroup barrier for LDS hand-offs only (no vector-memory drain: stores stay in flight across it)
__device__ __forceinline__ void rc_lds_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}

constexpr int rc_blob_bytes_c(int cin) { return ((64 * cin * 2 + 3072 + 512 + 1023) / 1024) * 1024; }
// 16-byte chunk c of row (pixel slot / weight row) f is stored at chunk c ^ key(f): conflict-free ds_read_b128 of an MFMA
// fragment (16 consecutive rows, chunks c and c+1 over the four 16-lane groups)
template <int CIN> __device__ __forceinline__ int rc_key(int f) { return CIN == 64 ? ((f >> 1) & 7) : (f & 15); }

#ifdef RC_PROFILE
__device__ unsigned long long rc_prof[256 * 8 * 8];
#define RC_STAMP(i) do { const unsigned long long now__ = __builtin_amdgcn_s_memtime(); pacc[i] += now__ - tprev; tprev = now__; } while (0)
#else
#define RC_STAMP(i) do { } while (0)
#endif

struct RcFwd {
    const bf16_t* a0; i64 a0_ld; const unsigned char* blob; bf16_t* out; double* stats;
    int planes, Hin, Win, Hout, Wout, E, R, FP, round_y1;
};

// MU: pair tiles per wave and slice (phase A), MI: pixel pairs per thread and slice (phase B) — compile-time bounds of the
// per-tile geometry kept in registers; the launcher picks the band height so that a tile fits an instantiation
template <int CIN, int ST, int NT, bool ROUND, int MU, int MI>
__global__ __launch_bounds__(NT, NT / 256) void dw_spatial_fwd_rc_kernel(const RcFwd a) {
    constexpr int CH = CIN / 8, KB = CIN / 32, PIXB = CIN * 2, W1B = 64 * CIN * 2;
    constexpr int CHSH = CIN == 64 ? 3 : 4;
    constexpr int BLOB = rc_blob_bytes_c(CIN), WP_OFF = W1B, CF_OFF = W1B + 3072;
    constexpr int NW = NT / 64, LP = NT / 16, NQ = NW / 4;
    constexpr int NPB = ST == 1 ? 2 : 3;            // tile pairs read per stencil row
    static_assert(CIN == 64 || CIN == 128, "rc kernels are built for Cin 64 / 128");
    static_assert((NW & 3) == 0, "a wave owns one of the four 16-channel MFMA row tiles of a slice");
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lr = lane & 15, lg = lane >> 4;
    const int Wpp = (a.Win + 3) >> 1, Wop = (a.Wout + 1) >> 1;
    const int FP = a.FP, NPT = FP >> 4, NS = a.E >> 6;
    const unsigned lds0 = (unsigned)(size_t)rc_smem;
    const unsigned offA0 = 0u, offA1 = (unsigned)FP * PIXB, offZ = 2u * FP * PIXB;
    const unsigned offB0 = offZ + (unsigned)FP * 256u, offS = offB0 + 2u * BLOB, offW = offS + 8u * (unsigned)a.E;
    float* lstat = reinterpret_cast<float*>(rc_smem + offS);
    for (int i = tid; i < 2 * a.E; i += NT) lstat[i] = 0.f;

    const FastDiv dvpp(Wpp), dvop(Wop);
    const int nbands = (a.Hout + a.R - 1) / a.R;
    const int ntiles = a.planes * nbands;
    const int NIA = (FP << CHSH) >> 6;               // DMA instructions per a0 sub-tile
    const unsigned rowb = (unsigned)Wpp * 256u;      // bytes per stencil-tile row

    auto issue_a0 = [&](const int tile) {
        const int plane = tile / nbands, band = tile - plane * nbands;
        const int ho0 = band * a.R;
        const int nro = (a.Hout - ho0 < a.R) ? a.Hout - ho0 : a.R;
        const int hi_first = ho0 * ST - 1;
        const int F = ((nro - 1) * ST + 3) * Wpp;
        const bf16_t* pl0 = a.a0 + (i64)plane * a.Hin * a.Win * a.a0_ld;
        for (int i = wave; i < 2 * NIA; i += NW) {
            const int sub = i >= NIA ? 1 : 0;
            const int j = i - sub * NIA;
            const int u = (j << 6) + lane;
            const int f = u >> CHSH, cphys = u & (CH - 1);
            const int c = cphys ^ rc_key<CIN>(f);
            const int r = dvpp.div(f);
            const int p = dvpp.rem(f, r);
            const int hi = hi_first + r, wi = 2 * p - 1 + sub;
            const bool ok = f < F && (unsigned)hi < (unsigned)a.Hin && (unsigned)wi < (unsigned)a.Win;
            const i64 off = ok ? (i64)(hi * a.Win + wi) * a.a0_ld + c * 8 : 0;
            rc_glds16(pl0 + off, (unsigned)__builtin_amdgcn_readfirstlane((int)(lds0 + (sub ? offA1 : offA0) + ((unsigned)j << 10))));
        }
    };
    auto issue_blob = [&](const int slice, const int buf) {
        const unsigned char* src = a.blob + (i64)slice * BLOB + lane * 16;
        for (int i = wave; i < BLOB / 1024; i += NW)
            rc_glds16(src + i * 1024, (unsigned)__builtin_amdgcn_readfirstlane((int)(lds0 + offB0 + (unsigned)buf * BLOB + ((unsigned)i << 10))));
    };

    int tile = blockIdx.x;
    if (tile < ntiles) { issue_a0(tile); issue_blob(0, 0); }
    rc_wait_vm0();
    rc_lds_barrier();
    int it = 0;
    const int cv = tid & 15, pl = tid >> 4;
    const int n = wave & 3;                         // this wave's MFMA row tile: channels 16n .. 16n+15 of the slice
    const int q = 4 * n + lg;                       // its 16-byte chunk of a stencil-tile pair (4 channels x (E,O))
#ifdef RC_PROFILE
    unsigned long long pacc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    unsigned long long tprev = __builtin_amdgcn_s_memtime();
#endif
    for (; tile < ntiles; tile += gridDim.x) {
        const int plane = tile / nbands, band = tile - plane * nbands;
        const int ho0 = band * a.R;
        const int nro = (a.Hout - ho0 < a.R) ? a.Hout - ho0 : a.R;
        const int hi_first = ho0 * ST - 1;
        const int F = ((nro - 1) * ST + 3) * Wpp;
        const int ntile = tile + (int)gridDim.x;
        // ---- per-tile geometry, kept in registers for all E/64 slices
        // phase A unit k of this wave: pair tile t = wave/4 + k*NQ, this lane's pair slot f = 16 t + lr
        int ub[MU], uz[MU];
        unsigned um[MU];
#pragma unroll
        for (int k = 0; k < MU; ++k) {
            const int f = ((wave >> 2) + k * NQ) * 16 + lr;
            const int r = dvpp.div(f);
            const int p = dvpp.rem(f, r);
            const int hi = hi_first + r;
            const bool okrow = f < F && (unsigned)hi < (unsigned)a.Hin;
            const bool okE = okrow && (unsigned)(2 * p - 1) < (unsigned)a.Win, okO = okrow && 2 * p < a.Win;
            um[k] = (okE ? 0x0000ffffu : 0u) | (okO ? 0xffff0000u : 0u);
            ub[k] = f * PIXB + ((lg ^ rc_key<CIN>(f)) << 4);              // k-step kb reads at ub ^ (kb << 6)
            uz[k] = (int)offZ + f * 256 + ((q ^ (p & 7)) << 4);
        }
        const int nu = (NPT - (wave >> 2) + NQ - 1) / NQ;                  // units of this wave (wave-uniform)
        // phase B item k of this thread: output pixel pair i = pl + k*LP of the band
        const int total = nro * Wop;
        int zb[MI][NPB];
        unsigned opix[MI];
        bool iv[MI], iodd[MI];
#pragma unroll
        for (int k = 0; k < MI; ++k) {
            const int i = pl + k * LP;
            iv[k] = i < total;
            const int ii = iv[k] ? i : 0;
            const int oy = dvop.div(ii);
            const int j = dvop.rem(ii, oy);
            const int p0 = ST == 1 ? j : 2 * j;
            const int fb = __mul24(oy * ST, Wpp);
#pragma unroll
            for (int m = 0; m < NPB; ++m) {
                int pm = p0 + m;
                if (ST == 2 && m == 2 && pm >= Wpp) pm = p0 + 1;          // only feeds the (invalid) odd output
                zb[k][m] = (int)offZ + (fb + pm) * 256 + ((cv ^ (pm & 7)) << 4);
            }
            opix[k] = (unsigned)(__mul24(oy, a.Wout) + 2 * j) * (unsigned)a.E * 2u + (unsigned)cv * 8u;     // bytes
            iodd[k] = 2 * j + 1 < a.Wout;
        }
        unsigned char* outt = reinterpret_cast<unsigned char*>(a.out + (((i64)plane * a.Hout + ho0) * a.Wout) * a.E);
        for (int s = 0; s < NS; ++s, ++it) {
            const int buf = it & 1;
            const bool last_slice = s == NS - 1;
            if (!last_slice || ntile < ntiles) issue_blob(last_slice ? 0 : s + 1, buf ^ 1);
            const unsigned char* blobp = rc_smem + offB0 + buf * BLOB;
            // ---------------- phase A: y1 slice by MFMA -> BN1 + SiLU -> pair-packed stencil tile.  The MFMAs of unit
            // k+1 are issued before the VALU work of unit k.
            {
                uint4 afr[KB];
                float sc[4], sh[4];
                {
                    const int row = 16 * n + lr;
#pragma unroll
                    for (int kb = 0; kb < KB; ++kb)
                        afr[kb] = *reinterpret_cast<const uint4*>(blobp + row * PIXB + (((kb * 4 + lg) ^ rc_key<CIN>(row)) << 4));
                    const int ch0 = 16 * n + 4 * lg;
                    const float4 s4 = *reinterpret_cast<const float4*>(blobp + CF_OFF + ch0 * 4);
                    const float4 t4 = *reinterpret_cast<const float4*>(blobp + CF_OFF + 256 + ch0 * 4);
                    sc[0] = s4.x; sc[1] = s4.y; sc[2] = s4.z; sc[3] = s4.w;
                    sh[0] = t4.x; sh[1] = t4.y; sh[2] = t4.z; sh[3] = t4.w;
                }
                auto unit_mma = [&](const int ubk, rc_f32x4_t& aE, rc_f32x4_t& aO) {
                    aE = rc_f32x4_t{0.f, 0.f, 0.f, 0.f}; aO = rc_f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int kb = 0; kb < KB; ++kb) {
                        const int o = ubk ^ (kb << 6);
                        const uint4 bE = *reinterpret_cast<const uint4*>(rc_smem + offA0 + o);
                        const uint4 bO = *reinterpret_cast<const uint4*>(rc_smem + offA1 + o);
                        aE = rc_mfma(afr[kb], bE, aE);
                        aO = rc_mfma(afr[kb], bO, aO);
                    }
                };
                auto unit_act = [&](const int uzk, const unsigned msk, const rc_f32x4_t& aE, const rc_f32x4_t& aO) {
                    unsigned d[4];
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        float vE = aE[j], vO = aO[j];
                        if constexpr (ROUND) {        // y1 as a stored bf16 tensor would read back
                            const unsigned pk = pk_bf16(vE, vO);
                            vE = __uint_as_float(pk << 16); vO = __uint_as_float(pk & 0xffff0000u);
                        }
                        const float hE = fmaf(vE, sc[j], sh[j]), hO = fmaf(vO, sc[j], sh[j]);
                        d[j] = pk_bf16(hE * sigmoidf_(hE), hO * sigmoidf_(hO)) & msk;
                    }
                    *reinterpret_cast<uint4*>(rc_smem + uzk) = make_uint4(d[0], d[1], d[2], d[3]);
                };
                rc_f32x4_t accE[2], accO[2];
                if (nu > 0) unit_mma(ub[0], accE[0], accO[0]);
#pragma unroll
                for (int k = 0; k < MU; ++k) {
                    if (k < nu) {
                        if (k + 1 < MU && k + 1 < nu) unit_mma(ub[k + 1 < MU ? k + 1 : k], accE[(k + 1) & 1], accO[(k + 1) & 1]);
                        unit_act(uz[k], um[k], accE[k & 1], accO[k & 1]);
                    }
                }
            }
            RC_STAMP(0);
            rc_wait_vm0();              // this wave's blob (and, at a tile's first slice, a0) DMA has landed
            RC_STAMP(1);
            rc_lds_barrier();           // ... everybody's; the stencil tile is complete; a0 / blob reads of phase A are done
            RC_STAMP(2);
            if (last_slice && ntile < ntiles) issue_a0(ntile);
            // ---------------- phase B: 3x3 stencil on the pair-packed tile, y2 store, BN2 sums
            {
                uint4 wp[3][ST == 1 ? 4 : 2];
#pragma unroll
                for (int dy = 0; dy < 3; ++dy)
#pragma unroll
                    for (int k = 0; k < (ST == 1 ? 4 : 2); ++k)
                        wp[dy][k] = *reinterpret_cast<const uint4*>(blobp + WP_OFF + ((dy * 4 + k) * 64 + cv * 4) * 4);
                rc_f2_t st0[2] = {rc_f2_t{0.f, 0.f}, rc_f2_t{0.f, 0.f}}, st1[2] = {rc_f2_t{0.f, 0.f}, rc_f2_t{0.f, 0.f}};
                unsigned char* outs = outt + s * 128;              // this slice's 128-byte channel segment (uniform)
#pragma unroll
                for (int k = 0; k < MI; ++k) {
                    if (!iv[k]) continue;
                    float acc0[4] = {0.f, 0.f, 0.f, 0.f}, acc1[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int dy = 0; dy < 3; ++dy) {
                        const unsigned wa[4] = {wp[dy][0].x, wp[dy][0].y, wp[dy][0].z, wp[dy][0].w};
                        const unsigned wb[4] = {wp[dy][1].x, wp[dy][1].y, wp[dy][1].z, wp[dy][1].w};
                        const uint4 p0 = *reinterpret_cast<const uint4*>(rc_smem + zb[k][0] + dy * rowb);
                        const uint4 p1 = *reinterpret_cast<const uint4*>(rc_smem + zb[k][1] + dy * rowb);
                        const unsigned x0[4] = {p0.x, p0.y, p0.z, p0.w}, x1[4] = {p1.x, p1.y, p1.z, p1.w};
                        if constexpr (ST == 1) {
                            const unsigned wc[4] = {wp[dy][ST == 1 ? 2 : 0].x, wp[dy][ST == 1 ? 2 : 0].y, wp[dy][ST == 1 ? 2 : 0].z, wp[dy][ST == 1 ? 2 : 0].w};
                            const unsigned wd[4] = {wp[dy][ST == 1 ? 3 : 1].x, wp[dy][ST == 1 ? 3 : 1].y, wp[dy][ST == 1 ? 3 : 1].z, wp[dy][ST == 1 ? 3 : 1].w};
#pragma unroll
                            for (int qq = 0; qq < 4; ++qq) {
                                acc0[qq] = rc_dot2(x0[qq], wa[qq], acc0[qq]);
                                acc0[qq] = rc_dot2(x1[qq], wb[qq], acc0[qq]);
                                acc1[qq] = rc_dot2(x0[qq], wc[qq], acc1[qq]);
                                acc1[qq] = rc_dot2(x1[qq], wd[qq], acc1[qq]);
                            }
                        } else {
                            // stride 2: output ox reads staged columns 2ox .. 2ox+2 = P[ox] and the low half of P[ox+1]
                            const uint4 p2 = *reinterpret_cast<const uint4*>(rc_smem + zb[k][NPB - 1] + dy * rowb);
                            const unsigned x2[4] = {p2.x, p2.y, p2.z, p2.w};
#pragma unroll
                            for (int qq = 0; qq < 4; ++qq) {
                                acc0[qq] = rc_dot2(x0[qq], wa[qq], acc0[qq]);
                                acc0[qq] = rc_dot2(x1[qq], wb[qq], acc0[qq]);
                                acc1[qq] = rc_dot2(x1[qq], wa[qq], acc1[qq]);
                                acc1[qq] = rc_dot2(x2[qq], wb[qq], acc1[qq]);
                            }
                        }
                    }
                    // BN2 sums over the values as stored (bf16-rounded), like the materialised path
                    const uint2 pk0 = make_uint2(pk_bf16(acc0[0], acc0[1]), pk_bf16(acc0[2], acc0[3]));
                    *reinterpret_cast<uint2*>(outs + opix[k]) = pk0;
                    rc_f2_t r0 = rc_f2_t{__uint_as_float(pk0.x << 16), __uint_as_float(pk0.x & 0xffff0000u)};
                    rc_f2_t r1 = rc_f2_t{__uint_as_float(pk0.y << 16), __uint_as_float(pk0.y & 0xffff0000u)};
                    st0[0] += r0; st0[1] += r1; st1[0] += r0 * r0; st1[1] += r1 * r1;
                    if (iodd[k]) {
                        const uint2 pk1 = make_uint2(pk_bf16(acc1[0], acc1[1]), pk_bf16(acc1[2], acc1[3]));
                        *reinterpret_cast<uint2*>(outs + opix[k] + (unsigned)a.E * 2u) = pk1;
                        r0 = rc_f2_t{__uint_as_float(pk1.x << 16), __uint_as_float(pk1.x & 0xffff0000u)};
                        r1 = rc_f2_t{__uint_as_float(pk1.y << 16), __uint_as_float(pk1.y & 0xffff0000u)};
                        st0[0] += r0; st0[1] += r1; st1[0] += r0 * r0; st1[1] += r1 * r1;
                    }
                    if (k & 1) __builtin_amdgcn_sched_barrier(0);   // at most two items' tile reads in flight (registers)
                }
                if (a.stats) {
                    // fold the four pixel lanes of a wave that share a channel vector with two transposing lane swaps
                    // (v_permlane16_swap / v_permlane32_swap: 12 VALU instructions for the 8 sums), then park the wave's
                    // 128 partial sums in its own LDS slot; LDS float atomics here cost ~180 cycles per instruction
                    unsigned v[8] = {__float_as_uint(st0[0].x), __float_as_uint(st0[0].y), __float_as_uint(st0[1].x), __float_as_uint(st0[1].y),
                                     __float_as_uint(st1[0].x), __float_as_uint(st1[0].y), __float_as_uint(st1[1].x), __float_as_uint(st1[1].y)};
                    float c4[4], d2[2];
#pragma unroll
                    for (int k = 0; k < 4; ++k) {       // 16-lane rows 0/2 keep the sums of v[k], rows 1/3 those of v[k+4]
                        const rc_u32x2_t r = __builtin_amdgcn_permlane16_swap(v[k], v[k + 4], false, false);
                        c4[k] = __uint_as_float(r.x) + __uint_as_float(r.y);
                    }
#pragma unroll
                    for (int k = 0; k < 2; ++k) {       // rows 0,1 keep c4[k], rows 2,3 keep c4[k+2]
                        const rc_u32x2_t r = __builtin_amdgcn_permlane32_swap(__float_as_uint(c4[k]), __float_as_uint(c4[k + 2]), false, false);
                        d2[k] = __uint_as_float(r.x) + __uint_as_float(r.y);
                    }
                    // lane (row r, cv) now holds value index (r&1)*4 + (r>>1)*2 + {0,1} of channel vector cv
                    const int r = lane >> 4;
                    float* slot = reinterpret_cast<float*>(rc_smem + offW) + wave * 128 + (r & 1) * 64 + cv * 4 + (r >> 1) * 2;
                    *reinterpret_cast<float2*>(slot) = make_float2(d2[0], d2[1]);
                }
            }
            RC_STAMP(3);
            if (last_slice) rc_wait_vm0();      // the next tile's a0 must have landed before its first phase A
            RC_STAMP(4);
            rc_lds_barrier();
            RC_STAMP(5);
            if (a.stats && tid < 128) {         // fold the waves' slots into the workgroup's running sums (one owner per address)
                const float* sl = reinterpret_cast<const float*>(rc_smem + offW) + tid;
                float acc = 0.f;
#pragma unroll
                for (int w = 0; w < NW; ++w) acc += sl[w * 128];
                float* dst = lstat + (tid >> 6) * a.E + s * 64 + (tid & 63);
                *dst += acc;
            }
        }
    }
#ifdef RC_PROFILE
    if (lane == 0 && blockIdx.x < 256 && wave < 8) {
        pacc[6] = (unsigned long long)it;
        for (int i = 0; i < 8; ++i) rc_prof[(blockIdx.x * 8 + wave) * 8 + i] = pacc[i];
    }
#endif
    if (a.stats) {
        __syncthreads();
        DET_ENTER();
        for (int i = tid; i < 2 * a.E; i += NT) {
            const int which = i >= a.E ? 1 : 0;
            stat_add(a.stats, (int)(blockIdx.x % DWN_NREP), a.E, which, i - which * a.E, lstat[i]);
        }
        DET_EXIT();
    }
}

// ------------------------------------------------------------------------------------------------
// blob builder: one workgroup per 64-channel slice
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void dws_rc_prep_kernel(const float* __restrict__ w1, const float* __restrict__ wdw,
                                                           const float* __restrict__ coef, int E, int CIN, unsigned char* blob) {
    const int slice = blockIdx.x, tid = threadIdx.x;
    const int CH = CIN / 8, W1B = 64 * CIN * 2;
    const int BLOB = ((W1B + 3072 + 512 + 1023) / 1024) * 1024;
    unsigned char* bp = blob + (i64)slice * BLOB;
    for (int idx = tid; idx < 64 * CH; idx += 256) {
        const int row = idx / CH, cphys = idx % CH;
        const int c = cphys ^ (CIN == 64 ? ((row >> 1) & 7) : (row & 15));
        const float* src = w1 + (i64)(slice * 64 + row) * CIN + c * 8;
        float v[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] = src[i];
        *reinterpret_cast<uint4*>(bp + idx * 16) = pack16<bf16_t>(v);
    }
    for (int idx = tid; idx < 12 * 64; idx += 256) {
        const int k = idx / 64, ch = idx % 64;
        const int dy = k >> 2, combo = k & 3;
        const int c = slice * 64 + ch;
        const float w0 = wdw[(i64)(dy * 3 + 0) * E + c], w1v = wdw[(i64)(dy * 3 + 1) * E + c], w2 = wdw[(i64)(dy * 3 + 2) * E + c];
        unsigned val;
        if (combo == 0) val = rc_pack2(w0, w1v);
        else if (combo == 1) val = rc_pack2(w2, 0.f);
        else if (combo == 2) val = rc_pack2(0.f, w0);
        else val = rc_pack2(w1v, w2);
        *reinterpret_cast<unsigned*>(bp + W1B + idx * 4) = val;
    }
    for (int idx = tid; idx < 128; idx += 256) {
        const int ch = idx & 63;
        *reinterpret_cast<float*>(bp + W1B + 3072 + idx * 4) = coef[(i64)(idx >> 6) * E + slice * 64 + ch];
    }
    for (int idx = W1B + 3072 + 512 + tid * 4; idx < BLOB; idx += 256 * 4) *reinterpret_cast<unsigned*>(bp + idx) = 0u;
}

// ------------------------------------------------------------------------------------------------
// launchers / C-ABI
// ------------------------------------------------------------------------------------------------
#define RC_LDS_MAX (160 * 1024)
#define RC_NT 768          // threads per workgroup (one workgroup per CU: the LDS tiles decide)
#define RC_MU_A 5          // geometry A: many pair tiles, one pixel pair per thread (wide stride-2 planes)
#define RC_MI_A 1
#define RC_MU_B 4          // geometry B
#define RC_MI_B(cin) ((cin) == 64 ? 3 : 2)
static size_t rc_fwd_lds_bytes(int Cin, int E, int Win, int stride, int R, int* FP_out) {
    const int Wpp = (Win + 3) >> 1;
    const int F = ((R - 1) * stride + 3) * Wpp;
    const int FP = (F + 15) & ~15;
    if (FP_out) *FP_out = FP;
    return (size_t)2 * FP * Cin * 2 + (size_t)FP * 256 + (size_t)2 * rc_blob_bytes_c(Cin) + (size_t)8 * E + (size_t)(RC_NT / 64) * 512;
}

// ONE predicate for "a band of R output rows can be launched": the tile must fit the LDS *and* one of the two compiled
// register geometries — (MU_A pair tiles per wave, MI_A pixel pairs per thread) or (MU_B, MI_B).  rc_supported() (what the
// block forward asks before it drops conv_pw) and the launcher both use it, so a plane the launcher would refuse (e.g.
// Cin = 64, stride 1, Win = 130: the LDS fits, no geometry does) falls back to conv_pw + stencil instead of failing.
static bool rc_fits(int Cin, int E, int Win, int Wout, int stride, int R, bool* few) {
    int FPq = 0;
    if (rc_fwd_lds_bytes(Cin, E, Win, stride, R, &FPq) > RC_LDS_MAX) return false;
    const int units = ((FPq >> 4) + RC_NT / 256 - 1) / (RC_NT / 256);
    const int items = (R * ((Wout + 1) >> 1) + RC_NT / 16 - 1) / (RC_NT / 16);
    if (units <= RC_MU_B && items <= RC_MI_B(Cin)) { if (few) *few = false; return true; }
    if (units <= RC_MU_A && items <= RC_MI_A) { if (few) *few = true; return true; }
    return false;
}

static bool rc_supported(int dtype, int Cin, int E, int ks, int stride, int Hin, int Win) {
    if (dtype != DWN_BF16 || ks != 3 || (stride != 1 && stride != 2)) return false;
    if ((Cin != 64 && Cin != 128) || E % 64 || E <= 0) return false;
    if (Hin < 1 || Win < 2) return false;
    return rc_fits(Cin, E, Win, (Win - 1) / stride + 1, stride, 1, nullptr);
}

#ifdef RC_PROFILE
extern "C" int dwn_rc_prof_read(unsigned long long* host_out) {
    return (int)hipMemcpyFromSymbol(host_out, HIP_SYMBOL(rc_prof), sizeof(unsigned long long) * 256 * 8 * 8);
}
#endif

extern "C" {

size_t dwn_dw_spatial_rc_blob_bytes(int E, int Cin) { return (size_t)(E / 64) * rc_blob_bytes_c(Cin); }

int dwn_dw_spatial_rc_supported(int dtype, int Cin, int E, int ks, int stride, int Hin, int Win) {
    return rc_supported(dtype, Cin, E, ks, stride, Hin, Win) ? 1 : 0;
}

int dwn_dw_spatial_rc_prep(const float* w_pw, const float* w_dws, const float* bn1_coef, int E, int Cin, void* blob,
                           int device, void* stream) {
    hipError_t e = hipSetDevice(device);
    if (e != hipSuccess) return dwn_set_error((int)e, hipGetErrorString(e));
    if (!w_pw || !w_dws || !bn1_coef || !blob) return dwn_set_error(-1, "dw_spatial_rc_prep: null pointer");
    if ((Cin != 64 && Cin != 128) || E % 64 || E <= 0) return dwn_set_error(-3, "dw_spatial_rc_prep: Cin must be 64 / 128 and E a multiple of 64");
    hipLaunchKernelGGL(dws_rc_prep_kernel, dim3(E / 64), dim3(256), 0, (hipStream_t)stream, w_pw, w_dws, bn1_coef, E, Cin,
                       (unsigned char*)blob);
    DWN_CHECK_LAUNCH();
    return 0;
}

int dwn_dw_spatial_fwd_rc(const dwn_dw_spatial_rc_fwd_args* ap, int device, void* stream) {
    hipError_t e = hipSetDevice(device);
    if (e != hipSuccess) return dwn_set_error((int)e, hipGetErrorString(e));
    const dwn_dw_spatial_rc_fwd_args& x = *ap;
    hipStream_t s = (hipStream_t)stream;
    if (!x.a0 || !x.blob || !x.out) return dwn_set_error(-1, "dw_spatial_fwd_rc: null pointer");
    if (!rc_supported(DWN_BF16, x.Cin, x.E, 3, x.stride, x.Hin, x.Win))
        return dwn_set_error(-3, "dw_spatial_fwd_rc: unsupported configuration (bf16, 3x3, stride 1/2, Cin 64/128, E % 64 == 0)");
    if (x.Hout != (x.Hin - 1) / x.stride + 1 || x.Wout != (x.Win - 1) / x.stride + 1)
        return dwn_set_error(-2, "dw_spatial_fwd_rc: Hout/Wout inconsistent with stride");
    if (x.a0_ld % 8 || x.a0_ld < x.Cin) return dwn_set_error(-2, "dw_spatial_fwd_rc: a0_ld must be a multiple of 8 and >= Cin");
    RcFwd k;
    k.a0 = (const bf16_t*)x.a0; k.a0_ld = x.a0_ld; k.blob = (const unsigned char*)x.blob; k.out = (bf16_t*)x.out;
    k.stats = x.stats; k.planes = x.planes; k.Hin = x.Hin; k.Win = x.Win; k.Hout = x.Hout; k.Wout = x.Wout; k.E = x.E;
    k.round_y1 = x.round_y1;
    // among the band heights that fit (rc_fits) the largest wins, split evenly over the plane
    auto fits = [&](int R, bool* few) { return rc_fits(x.Cin, x.E, x.Win, x.Wout, x.stride, R, few); };
    int R = x.rows_band;
    bool few_items = false;
    if (R <= 0) {
        R = 0;
        for (int r = 1; r <= x.Hout; ++r) { bool f; if (fits(r, &f)) R = r; }
        if (R == 0) return dwn_set_error(-5, "dw_spatial_fwd_rc: plane too wide for the LDS tile");
        int nb = (x.Hout + R - 1) / R;                 // even split: no ragged last band
        int Re = (x.Hout + nb - 1) / nb;
        bool f;
        if (fits(Re, &f)) R = Re;
    }
    if (R > x.Hout) R = x.Hout;
    if (!fits(R, &few_items)) return dwn_set_error(-5, "dw_spatial_fwd_rc: rows_band does not fit the LDS tile / register geometry");
    int FP = 0;
    const size_t lds = rc_fwd_lds_bytes(x.Cin, x.E, x.Win, x.stride, R, &FP);
    k.R = R; k.FP = FP;
    const int nbands = (x.Hout + R - 1) / R;
    const i64 ntiles = (i64)x.planes * nbands;
    if (ntiles <= 0) return 0;
#define RC_FWD_LAUNCH4(CIN_, ST_, RND_, MU_, MI_) do { \
        auto kern = dw_spatial_fwd_rc_kernel<CIN_, ST_, RC_NT, RND_, MU_, MI_>; \
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) \
            (void)hipGetLastError(); \
        int bpc = 0; \
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&bpc, kern, RC_NT, lds) != hipSuccess || bpc < 1) { (void)hipGetLastError(); bpc = 1; } \
        const i64 cap = (i64)256 * bpc; \
        dim3 grid((unsigned)(ntiles < cap ? ntiles : cap)); \
        hipLaunchKernelGGL(kern, grid, dim3(RC_NT), lds, s, k); } while (0)
#define RC_FWD_LAUNCH3(CIN_, ST_, MU_, MI_) do { if (x.round_y1) RC_FWD_LAUNCH4(CIN_, ST_, true, MU_, MI_); else RC_FWD_LAUNCH4(CIN_, ST_, false, MU_, MI_); } while (0)
#define RC_FWD_LAUNCH(CIN_, ST_) do { if (few_items) RC_FWD_LAUNCH3(CIN_, ST_, RC_MU_A, RC_MI_A); else RC_FWD_LAUNCH3(CIN_, ST_, RC_MU_B, RC_MI_B(CIN_)); } while (0)
    if (x.Cin == 64) { if (x.stride == 1) RC_FWD_LAUNCH(64, 1); else RC_FWD_LAUNCH(64, 2); }
    else { if (x.stride == 1) RC_FWD_LAUNCH(128, 1); else RC_FWD_LAUNCH(128, 2); }
#undef RC_FWD_LAUNCH4
#undef RC_FWD_LAUNCH3
#undef RC_FWD_LAUNCH
    DWN_CHECK_LAUNCH();
    return 0;
}

}  // extern "C"
