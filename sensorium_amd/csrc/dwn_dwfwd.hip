// spat_covn_dw forward (reference: src/models/dwiseneuro.py:96-102), bf16 storage, 3x3, stride 1 or 2 — chained row-walk kernels.
//
//   y2 = dwS * SiLU(BN1(y1))   (+ Σy2, Σy2² for BatchNorm-2)
//
// Same arithmetic as dw_spatial_fwd_pair_kernel (activated input staged x-pair-packed in LDS, v_dot2c_f32_bf16 taps, bf16
// stencil weights) with the loop organisation of the row-walk backward kernels (dwn_dwbwd.hip): a thread stages its own
// pixel-pair column(s) row after row (constant halo mask, addresses advance by constants), then walks down the output rows
// of its output pair column with the tile rows it needs in a sliding register window; planes narrower than 32 output pixels
// sit side by side in one tile.  The stencils are VALU-issue-bound, and a quarter of the pair kernel's instructions were
// flat-index decodes.  y2 is bit-identical to the pair kernel's.
#include "dwn_internal.h"
#include <stdlib.h>

extern __shared__ __attribute__((aligned(16))) unsigned char wf_smem[];


#ifndef WF_PRIO
#define WF_PRIO 1          // issue priority of the staging / rebuild phase (0 = off)
#endif
typedef float wf_f2_t __attribute__((ext_vector_type(2)));
typedef __attribute__((ext_vector_type(2))) __bf16 wf_bf16x2_t;
typedef unsigned wf_u32x2_t __attribute__((ext_vector_type(2)));

#ifndef WF_MINW
#define WF_MINW 3
#endif
#define WF_MINW_RC4 3      // ... chunks of 4 input rows fit three
#ifndef WF_MINW_RC
#define WF_MINW_RC 2       // the y1-rebuilding form keeps 32 registers of W1 fragments: chunks of 8 input rows need two waves per SIMD
#endif
typedef __attribute__((ext_vector_type(8))) short wf_bf16x8_t;
typedef __attribute__((ext_vector_type(4))) float wf_f32x4_t;
__device__ __forceinline__ wf_f32x4_t wf_mfma(const uint4& a, const uint4& b, const wf_f32x4_t& c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(wf_bf16x8_t, a), __builtin_bit_cast(wf_bf16x8_t, b), c, 0, 0, 0);
}
// workgroup -> (plane-group start, channel slice); XCD = true: the slices of a plane group are consecutive workgroups of one XCD
// (dwn_dwbwd.hip wk_block: what they all read — the block input a0 — is fetched into that L2 once).  gridDim.x % 8 == 0.
struct WfBlk { int x, y; };
template <bool XCD>
__device__ __forceinline__ WfBlk wf_block() {
    WfBlk r;
    if constexpr (XCD) {
        const int gx = (int)gridDim.x, ns = (int)gridDim.y;
        const int b = (int)blockIdx.y * gx + (int)blockIdx.x;
        const int xcd = b & 7, i = b >> 3;
        r.y = i % ns; r.x = (i / ns) * 8 + xcd;
    } else {
        r.x = (int)blockIdx.x; r.y = (int)blockIdx.y;
    }
    return r;
}

__device__ __forceinline__ float wf_dot2(unsigned a, unsigned b, float c) {
    return __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(wf_bf16x2_t, a), __builtin_bit_cast(wf_bf16x2_t, b), c, false);
}
// first tap of an accumulator: the three-operand form with the constant 0 as addend (hipcc selects v_dot2c_f32_bf16, which
// accumulates in place, for the builtin and spends a v_mov on zeroing every accumulator first: 8 of a walk step's 79 instructions)
__device__ __forceinline__ float wf_dot2z(unsigned a, unsigned b) {
    float r;
    asm("v_dot2_f32_bf16 %0, %1, %2, 0" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ void wf_unpack(const uint2& r, wf_f2_t& lo, wf_f2_t& hi) {
    lo = wf_f2_t{__uint_as_float(r.x << 16), __uint_as_float(r.x & 0xffff0000u)};
    hi = wf_f2_t{__uint_as_float(r.y << 16), __uint_as_float(r.y & 0xffff0000u)};
}
__device__ __forceinline__ uint2 wf_ld8(const bf16_t* p) { return *reinterpret_cast<const uint2*>(p); }

// workgroup barrier for LDS hand-offs only (__syncthreads() also drains vmcnt: the write acknowledgements of every y2 store)
__device__ __forceinline__ void wf_lds_barrier() {
#ifdef WK_FULL_BARRIER
    __syncthreads();
#else
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
#endif
}

// ------------------------------------------------------------------------------------------------
// CHAINED rows.  LPW = output pixel pairs per plane row (Wout == 2*LPW in {32, 16, 8}); NG = 16/LPW planes side by side in one
// tile; the staged tile has NPC = ST*LPW + 1 pair columns k = (wi = 2k-1, wi = 2k).
// A workgroup walks a whole plane group top to bottom in chunks of RB output rows and keeps the activated tile as a RING of
// row slots: the halo rows a chunk needs are the previous chunk's last rows, still in LDS, so no input row is fetched or
// activated twice, and a chunk's rows are fetched in ONE batch (one dependent round trip per chunk).  y2 is bit-identical to
// the pair kernel's (same dot2 order).
//   stride 1: chunk c stages input rows s .. s+RB-1 (s = c*RB) and produces output rows s-1 .. s+RB-2; ring of RB+2 rows
//   stride 2: chunk c stages input rows 2s .. 2s+2RB-1 and produces output rows s .. s+RB-1;          ring of 2RB+1 rows
// ring slot of input row r = (r + 1) mod RQ; row -1 (slot 0) is the zero row above the plane.
// ------------------------------------------------------------------------------------------------
// CIN > 0: the input is NOT read — a.in.p is ignored and the activated rows are REBUILT from the block input a0 (a.a0, CIN channels)
// as SiLU(BN1(a0 . W1^T)) on the matrix cores, BatchNorm-1 applied to the fp32 accumulators (round 6: the product is NOT rounded to
// bf16 first — these are the values the Gram-matrix statistics describe, and the backward stencils rebuild the same accumulators).
// The MFMA runs with the PIXELS as its A operand (16 consecutive pixels of the plane-group row) and W1 rows as B in the order
// 4 col + n, so a lane's accumulators are y1[pixels 4g .. 4g+3][channels 4 col .. 4 col + 3]: two complete x-pairs of four
// consecutive channels — two ds_write_b128, no lane exchange.  For that the ring holds EVEN-aligned pairs in this form,
// Q_k = (x = 2k, x = 2k+1) in pair column k + 1, with an all-zero column on either side of a plane (NPC = W/2 + 2); the walk reads
// three pair columns per row (stride 1: Q_{jj-1}, Q_jj, Q_{jj+1}; stride 2: Q_{2jj-1}, Q_{2jj}, Q_{2jj+1}), four dot products per
// tap row and channel as before.  Pair columns are 64 dwords apart: a ds_write_b128 group (8 lanes x 16 B) and a ds_read_b128
// group (16 lanes: 4 + 4 + 8 of two pair columns) each cover distinct banks.
template <int ST, int LPW, int RB, int CIN>
__global__ __launch_bounds__(256, CIN > 0 ? (ST * RB <= 4 && CIN <= 64 ? WF_MINW_RC4 : WF_MINW_RC) : WF_MINW) void dw_spatial_fwd_chain_kernel(const DwSpatialFwd a) {
    typedef bf16_t T;
    constexpr bool QL = CIN > 0;                         // even-aligned pairs Q_k = (2k, 2k+1) + a zero column on either side
    constexpr int NT = 256, CS = 64, NG = 16 / LPW, NPC = ST * LPW + (QL ? 2 : 1);
    constexpr int PS = 64;                               // dwords between consecutive pairs of a ring row
    constexpr int NWC = ST == 1 ? 4 : 2;
    constexpr int NR = ST * RB;                          // input rows staged per chunk
    constexpr int RQ = ST == 1 ? RB + 2 : 2 * RB + 1;
    constexpr int rowdw = NPC * PS;
    constexpr int KB = CIN > 0 ? CIN / 32 : 1;
    __shared__ float lstat[2 * CS];
    __shared__ __attribute__((aligned(16))) float lcoef[2 * CS];             // BatchNorm-1 scale, shift (re-read per phase: registers)
    // rebuilt form: every coefficient TWICE, (s_c, s_c) — read as the register pairs of the pixel-pair-packed math.  A splat built in
    // registers is folded by hipcc into v_pk_fma_f32 ... op_sel:[1,0,1] (both lanes read the pair's high register): the instruction
    // class of the round-5 corruption, refused by tools/check_isa.py
    __shared__ __attribute__((aligned(16))) float lcoef2[QL ? 4 * CS : 4];
    const int tid = threadIdx.x, lane = tid & 63;
    const int cv = tid & 15, pl = tid >> 4;
    const int grp = pl / LPW, jj = pl % LPW;
    const WfBlk blk = wf_block<(CIN > 0)>();
    const int c0 = blk.y * CS;
    const int chan = c0 + cv * 4;
    const bool chan_ok = chan < a.C;
    const int chs = chan_ok ? chan : 0;
    if (tid < 2 * CS) {
        lstat[tid] = 0.f;
        const int c = c0 + (tid & (CS - 1));
        const float v = c < a.C ? (tid < CS ? a.in.v1[c] : a.in.v2[c]) : 0.f;
        lcoef[tid] = v;
        if constexpr (QL) { lcoef2[2 * tid] = v; lcoef2[2 * tid + 1] = v; }
    }
    __syncthreads();

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
                if constexpr (QL && ST == 1) {
                    // E = (Qa.hi, Qc.lo), Qb:  out[2jj] = E.(w0,0) + Qb.(w1,w2),  out[2jj+1] = Qb.(w0,w1) + E.(0,w2)
                    wp[dy][0][q] = pk_bf16(w[dy * 3 + 0][q], 0.f);
                    wp[dy][1][q] = pk_bf16(w[dy * 3 + 1][q], w[dy * 3 + 2][q]);
                    wp[dy][2][q] = pk_bf16(w[dy * 3 + 0][q], w[dy * 3 + 1][q]);
                    wp[dy][3][q] = pk_bf16(0.f, w[dy * 3 + 2][q]);
                } else if constexpr (QL) {
                    // pairs Qa, Qb, Qc:  out[2jj] = Qa.(0,w0) + Qb.(w1,w2),  out[2jj+1] = Qb.(0,w0) + Qc.(w1,w2)
                    wp[dy][0][q] = pk_bf16(0.f, w[dy * 3 + 0][q]);
                    wp[dy][1][q] = pk_bf16(w[dy * 3 + 1][q], w[dy * 3 + 2][q]);
                } else {
                    wp[dy][0][q] = pk_bf16(w[dy * 3 + 0][q], w[dy * 3 + 1][q]);
                    wp[dy][1][q] = pk_bf16(w[dy * 3 + 2][q], 0.f);
                    if constexpr (ST == 1) {
                        wp[dy][2][q] = pk_bf16(0.f, w[dy * 3 + 0][q]);
                        wp[dy][3][q] = pk_bf16(w[dy * 3 + 1][q], w[dy * 3 + 2][q]);
                    }
                }
            }
    }
    float st0[4] = {0.f, 0.f, 0.f, 0.f}, st1[4] = {0.f, 0.f, 0.f, 0.f};            // BatchNorm-2 sums of this thread's 4 channels

    const int Hin = a.Hin, Win = a.Win, Hout = a.Hout, Wout = a.Wout;
    const int ngroups = (a.planes + NG - 1) / NG;
    const int nchunks = ST == 1 ? (Hin + RB) / RB : (Hout + RB - 1) / RB;
    const T* inp = reinterpret_cast<const T*>(a.in.p);
    T* outp = reinterpret_cast<T*>(a.out);
    unsigned* tile = reinterpret_cast<unsigned*>(wf_smem);        // ring: [NG][RQ][NPC][64] dwords
    unsigned* tplane = tile + grp * RQ * rowdw + cv * 4;
    static_assert(CIN == 0 || CIN == 64 || CIN == 128, "the y1-rebuilding form is built for 64 / 128 input channels");
    const unsigned inrow = (unsigned)Win * (unsigned)a.in.ld, outrow = (unsigned)Wout * (unsigned)a.C;
    // staging roles: stride 1 = the walk role (4 channels, own output pair column); stride 2 = 8 channels per lane, one lane
    // per input pair column (16-byte loads)
    constexpr int SLW = 2 * LPW;
    const int scv = tid & 7, spl = tid >> 3;
    const int sgrp = ST == 1 ? grp : spl / SLW, kc = ST == 1 ? jj : spl % SLW;
    const int sch = ST == 1 ? chan : c0 + scv * 8;
    const int schs = sch < a.C ? sch : 0;
    unsigned* splane_t = ST == 1 ? tplane : tile + sgrp * RQ * rowdw + scv * 8;
    const unsigned cmask = (kc > 0 ? 0x0000ffffu : 0u) | 0xffff0000u;
    const unsigned colhi = (unsigned)(2 * kc) * (unsigned)a.in.ld;
    const unsigned lodelta = kc > 0 ? (unsigned)a.in.ld : 0u;
    const unsigned collast = (unsigned)(Win - 1) * (unsigned)a.in.ld;
    constexpr int NLC = ST == 1 ? LPW : SLW;                      // staging lanes per plane row
    constexpr int NEX = (NR + NLC - 1) / NLC;                     // halo-column rows a lane stages per chunk

    // ---- rebuilt input (CIN > 0): lane (lr, lg) of an MFMA = (pixel lr of the 16-pixel tile, k group lg) for the a0 operand (A),
    // (W1 row 4 lr + n of channel tile n, k group lg) for W1 (B); accumulator register j of tile n = y1[pixel 4 lg + j][channel 4 lr + n]
    const int lr = lane & 15, lg = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // W1 fragments of this workgroup's 64 channels: registers for CIN = 64 (32 VGPRs); for CIN = 128 a copy of the slice in LDS behind
    // the ring (16 KB, 16-byte chunk c of row r at chunk c ^ ((r >> 2) & 15): conflict-free ds_read_b128 of a fragment)
    constexpr bool W1_LDS = CIN > 64;
    constexpr unsigned RING_BYTES = (unsigned)NG * RQ * rowdw * 4u;
    uint4 wfr[W1_LDS ? 1 : 4][KB];
    if constexpr (CIN > 0) {
        const T* w1 = reinterpret_cast<const T*>(a.w1);
        if constexpr (W1_LDS) {
            constexpr int CH = CIN / 8;
            for (int i = tid; i < 64 * CH; i += NT) {
                const int r = i / CH, c = i % CH;
                const int ch = c0 + r;
                *reinterpret_cast<uint4*>(wf_smem + RING_BYTES + (r * CH + (c ^ ((r >> 2) & 15))) * 16) =
                    *reinterpret_cast<const uint4*>(w1 + (i64)(ch < a.C ? ch : c0) * CIN + 8 * c);
            }
            __syncthreads();
        } else {
#pragma unroll
            for (int n = 0; n < 4; ++n) {
                const int ch = c0 + 4 * lr + n;
#pragma unroll
                for (int kb = 0; kb < KB; ++kb)
                    wfr[n][kb] = *reinterpret_cast<const uint4*>(w1 + (i64)(ch < a.C ? ch : c0) * CIN + 8 * lg + 32 * kb);
            }
        }
        // the zero columns on either side of every plane, all ring slots (never written again)
        for (int i = tid; i < NG * RQ * 2 * 16; i += NT) {
            const int c16 = i & 15, side = (i >> 4) & 1, rs = i >> 5;
            *reinterpret_cast<uint4*>(reinterpret_cast<unsigned*>(wf_smem) + (rs * NPC + (side ? NPC - 1 : 0)) * PS + 4 * c16) = make_uint4(0, 0, 0, 0);
        }
    }
    auto w1frag = [&](const int n, const int kb) -> uint4 {
        if constexpr (W1_LDS) {
            const int r = 4 * lr + n;
            return *reinterpret_cast<const uint4*>(wf_smem + RING_BYTES + (r * (CIN / 8) + ((lg + 4 * kb) ^ lr)) * 16);
        } else {
            return wfr[n][kb];
        }
    };
    (void)lr; (void)lg; (void)wave;

    // one plane-group row of a0 as MFMA A-operand fragments: [16-pixel tile t][k step]
    constexpr int NTL = 2 * ST;                  // 16-pixel tiles per plane-group row (32 ST pixels)
    constexpr int WIN = 2 * ST * LPW;            // pixels per plane row
    uint4 pf[NTL][KB];
    auto load_row = [&](const int pg_, const int hi, uint4 (&fr)[NTL][KB]) {
        if constexpr (CIN > 0) {
            const T* a0p = reinterpret_cast<const T*>(a.a0);
            const unsigned a0ld = (unsigned)a.a0_ld;
#pragma unroll
            for (int t = 0; t < NTL; ++t) {
                const int gxp = 16 * t + lr;             // this lane's pixel of the plane-group row
                const int g = gxp / WIN, x = gxp % WIN;
                const int pln = pg_ * NG + g;
                const T* src = a0p + ((i64)(pln < a.planes ? pln : 0) * a.Hin + (hi < a.Hin ? hi : a.Hin - 1)) * a.Win * (i64)a0ld +
                               (unsigned)x * a0ld + 8 * lg;     // rows / planes past the end: any valid address (masked below)
#pragma unroll
                for (int kb = 0; kb < KB; ++kb) fr[t][kb] = *reinterpret_cast<const uint4*>(src + 32 * kb);
            }
        }
    };
    auto rebuild_row = [&](const int pg_, const int hi, int sl, const uint4 (&fr)[NTL][KB]) {
        if constexpr (CIN > 0) {
            unsigned* tile_ = reinterpret_cast<unsigned*>(wf_smem);
            sl = sl >= RQ ? sl - RQ : sl;
            // BatchNorm-1 (scale, scale) / (shift, shift) of channels 4 lr + n (re-read per row: registers)
            const float4 sa = *reinterpret_cast<const float4*>(&lcoef2[8 * lr]), sb = *reinterpret_cast<const float4*>(&lcoef2[8 * lr + 4]);
            const float4 ta = *reinterpret_cast<const float4*>(&lcoef2[2 * CS + 8 * lr]), tb = *reinterpret_cast<const float4*>(&lcoef2[2 * CS + 8 * lr + 4]);
            const wf_f2_t bsc[4] = {wf_f2_t{sa.x, sa.y}, wf_f2_t{sa.z, sa.w}, wf_f2_t{sb.x, sb.y}, wf_f2_t{sb.z, sb.w}};
            const wf_f2_t bsh[4] = {wf_f2_t{ta.x, ta.y}, wf_f2_t{ta.z, ta.w}, wf_f2_t{tb.x, tb.y}, wf_f2_t{tb.z, tb.w}};
#pragma unroll
            for (int t = 0; t < NTL; ++t) {
                const int gx0 = 16 * t + 4 * lg;         // first of this lane's four pixels (a quad never straddles planes: 4 | WIN)
                const int g = gx0 / WIN, x0 = gx0 % WIN;
                const unsigned okmask = (pg_ * NG + g < a.planes && hi < a.Hin) ? 0xffffffffu : 0u;     // rows below / planes past the end: zeros
                unsigned* dst = tile_ + ((g * RQ + sl) * NPC + (x0 >> 1) + 1) * PS + 4 * lr;
                wf_f32x4_t acc[4];
#pragma unroll
                for (int n = 0; n < 4; ++n) {
                    acc[n] = wf_f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int kb = 0; kb < KB; ++kb) acc[n] = wf_mfma(fr[t][kb], w1frag(n, kb), acc[n]);
                }
                // BatchNorm-1 + SiLU on the accumulators, packed over PIXEL pairs (registers j, j + 1 of one channel) with the channel's
                // scale / shift duplicated in a register pair read from LDS: no op_sel on a low lane (tools/check_isa.py gates this object)
                unsigned q0[4], q1[4];
#pragma unroll
                for (int n = 0; n < 4; ++n) {
                    const wf_f2_t s2 = bsc[n], t2 = bsh[n];
                    const wf_f2_t h0 = wf_f2_t{acc[n][0], acc[n][1]} * s2 + t2, h1 = wf_f2_t{acc[n][2], acc[n][3]} * s2 + t2;
                    const wf_f2_t z0 = h0 * sigmoid2f_(h0), z1 = h1 * sigmoid2f_(h1);
                    q0[n] = pk_bf16(z0.x, z0.y) & okmask;                    // Q pair (x0, x0 + 1)
                    q1[n] = pk_bf16(z1.x, z1.y) & okmask;                    // Q pair (x0 + 2, x0 + 3)
                }
                *reinterpret_cast<uint4*>(dst) = make_uint4(q0[0], q0[1], q0[2], q0[3]);
                *reinterpret_cast<uint4*>(dst + PS) = make_uint4(q1[0], q1[1], q1[2], q1[3]);
            }
        }
    };
    (void)pf;

    for (int pg = blk.x; pg < ngroups; pg += gridDim.x) {
        const int plane = pg * NG + grp;
        const bool pvalid = plane < a.planes && chan_ok;
        const int splane = pg * NG + sgrp;
        const bool svalid = splane < a.planes && sch < a.C;
        const T* in0 = inp + (i64)(splane < a.planes ? splane : 0) * Hin * Win * a.in.ld + schs;
        T* out0 = outp + (i64)(plane < a.planes ? plane : 0) * Hout * Wout * a.C + chan + (unsigned)(2 * jj) * (unsigned)a.C;
        // row -1 of the ring (slot 0): zeros
        if constexpr (QL) {
            for (int i = tid; i < NG * NPC * 16; i += NT) {
                const int c16 = i & 15, col = (i >> 4) % NPC, gq = (i >> 4) / NPC;
                *reinterpret_cast<uint4*>(tile + (gq * RQ * NPC + col) * PS + 4 * c16) = make_uint4(0, 0, 0, 0);
            }
        } else {
            unsigned* z = splane_t + kc * PS;
            if constexpr (ST == 1) *reinterpret_cast<uint4*>(z) = make_uint4(0, 0, 0, 0);
            else { reinterpret_cast<uint4*>(z)[0] = make_uint4(0, 0, 0, 0); reinterpret_cast<uint4*>(z)[1] = make_uint4(0, 0, 0, 0); }
            if (kc == 0) {
                unsigned* zl = splane_t + (NPC - 1) * PS;
                if constexpr (ST == 1) *reinterpret_cast<uint4*>(zl) = make_uint4(0, 0, 0, 0);
                else { reinterpret_cast<uint4*>(zl)[0] = make_uint4(0, 0, 0, 0); reinterpret_cast<uint4*>(zl)[1] = make_uint4(0, 0, 0, 0); }
            }
        }
        int slot_s = 1;                                  // ring slot of input row ST*s
        if constexpr (CIN > 0 && NR == 4) load_row(pg, wave, pf);
        for (int chunk = 0; chunk < nchunks; ++chunk) {
            const int s = chunk * RB;
            const int hi_s = ST * s;                     // first input row staged by this chunk
            // ---------------- stage SiLU(BN1(y1)) rows hi_s .. hi_s + NR - 1, x-pair-packed, into their ring slots
            // (at raised issue priority: the other workgroups of the CU are mostly in their walk, whose LDS reads and dot products
            // can wait — this phase ends in the barrier all four waves of this workgroup need; measured -4 ... -7 % on the rebuilt form)
            __builtin_amdgcn_s_setprio(WF_PRIO);
            if constexpr (CIN > 0) {
                // ---------------- rebuild SiLU(BN1(y1)) of input rows hi_s .. hi_s + NR - 1 from a0: wave w takes rows w, w + 4, ...
                if constexpr (NR == 4) {
                    rebuild_row(pg, hi_s + wave, slot_s + wave, pf);                   // fragments fetched one chunk ahead
                    if (chunk + 1 < nchunks) load_row(pg, hi_s + NR + wave, pf);       // in flight under the walk below
                } else {
                    for (int u = wave; u < NR; u += 4) {
                        load_row(pg, hi_s + u, pf);
                        rebuild_row(pg, hi_s + u, slot_s + u, pf);
                    }
                }
            } else if constexpr (ST == 1) {
                const float4 s4 = *reinterpret_cast<const float4*>(&lcoef[cv * 4]), t4 = *reinterpret_cast<const float4*>(&lcoef[CS + cv * 4]);
                const wf_f2_t bs2[2] = {wf_f2_t{s4.x, s4.y}, wf_f2_t{s4.z, s4.w}}, bt2[2] = {wf_f2_t{t4.x, t4.y}, wf_f2_t{t4.z, t4.w}};
                auto act_pack = [&](const uint2& rlo, const uint2& rhi) {
                    wf_f2_t y0, y1v, z[2][2];
                    wf_unpack(rlo, y0, y1v);
                    {
                        const wf_f2_t h0 = y0 * bs2[0] + bt2[0], h1 = y1v * bs2[1] + bt2[1];
                        z[0][0] = h0 * sigmoid2f_(h0);
                        z[0][1] = h1 * sigmoid2f_(h1);
                    }
                    wf_unpack(rhi, y0, y1v);
                    {
                        const wf_f2_t h0 = y0 * bs2[0] + bt2[0], h1 = y1v * bs2[1] + bt2[1];
                        z[1][0] = h0 * sigmoid2f_(h0);
                        z[1][1] = h1 * sigmoid2f_(h1);
                    }
                    return make_uint4(pk_bf16(z[0][0].x, z[1][0].x), pk_bf16(z[0][0].y, z[1][0].y), pk_bf16(z[0][1].x, z[1][1].x), pk_bf16(z[0][1].y, z[1][1].y));
                };
                uint2 rr[NR][2], er[NEX];
                bool ok[NR], eok[NEX];
#pragma unroll
                for (int u = 0; u < NR; ++u) {
                    const int hi = hi_s + u;
                    ok[u] = svalid && hi < Hin;
                    const unsigned off = ok[u] ? (unsigned)hi * inrow + colhi : lodelta;
                    rr[u][1] = wf_ld8(in0 + off);
                    rr[u][0] = wf_ld8(in0 + off - lodelta);
                }
#pragma unroll
                for (int e = 0; e < NEX; ++e) {
                    const int u = kc + e * NLC;
                    eok[e] = svalid && u < NR && hi_s + u < Hin;
                    er[e] = wf_ld8(in0 + (eok[e] ? (unsigned)(hi_s + u) * inrow + collast : 0u));
                }
                int sl = slot_s;
#pragma unroll
                for (int u = 0; u < NR; ++u) {
                    uint4 o = act_pack(rr[u][0], rr[u][1]);
                    const unsigned m = ok[u] ? cmask : 0u;
                    o.x &= m; o.y &= m; o.z &= m; o.w &= m;
                    *reinterpret_cast<uint4*>(splane_t + kc * PS + sl * rowdw) = o;
                    sl = sl + 1 == RQ ? 0 : sl + 1;
                }
#pragma unroll
                for (int e = 0; e < NEX; ++e) {
                    const int u = kc + e * NLC;
                    if (u < NR) {
                        uint4 o = act_pack(er[e], er[e]);
                        const unsigned m = eok[e] ? 0x0000ffffu : 0u;
                        o.x &= m; o.y &= m; o.z &= m; o.w &= m;
                        int se = slot_s + u; se = se >= RQ ? se - RQ : se;
                        *reinterpret_cast<uint4*>(splane_t + (NPC - 1) * PS + se * rowdw) = o;
                    }
                }
            } else {
                wf_f2_t s8[4], t8[4];
                {
                    const float4 sa = *reinterpret_cast<const float4*>(&lcoef[scv * 8]), sb = *reinterpret_cast<const float4*>(&lcoef[scv * 8 + 4]);
                    const float4 ta = *reinterpret_cast<const float4*>(&lcoef[CS + scv * 8]), tb = *reinterpret_cast<const float4*>(&lcoef[CS + scv * 8 + 4]);
                    s8[0] = wf_f2_t{sa.x, sa.y}; s8[1] = wf_f2_t{sa.z, sa.w}; s8[2] = wf_f2_t{sb.x, sb.y}; s8[3] = wf_f2_t{sb.z, sb.w};
                    t8[0] = wf_f2_t{ta.x, ta.y}; t8[1] = wf_f2_t{ta.z, ta.w}; t8[2] = wf_f2_t{tb.x, tb.y}; t8[3] = wf_f2_t{tb.z, tb.w};
                }
                auto act_pack8 = [&](const uint4& rlo, const uint4& rhi, const unsigned m, unsigned* dst) {
                    const unsigned wa[4] = {rlo.x, rlo.y, rlo.z, rlo.w}, wb[4] = {rhi.x, rhi.y, rhi.z, rhi.w};
                    unsigned o[8];
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const wf_f2_t ya = wf_f2_t{__uint_as_float(wa[i] << 16), __uint_as_float(wa[i] & 0xffff0000u)};
                        const wf_f2_t yb = wf_f2_t{__uint_as_float(wb[i] << 16), __uint_as_float(wb[i] & 0xffff0000u)};
                        const wf_f2_t ha = ya * s8[i] + t8[i], hb = yb * s8[i] + t8[i];
                        const wf_f2_t za = ha * sigmoid2f_(ha);
                        const wf_f2_t zb = hb * sigmoid2f_(hb);
                        o[2 * i] = pk_bf16(za.x, zb.x) & m;
                        o[2 * i + 1] = pk_bf16(za.y, zb.y) & m;
                    }
                    reinterpret_cast<uint4*>(dst)[0] = make_uint4(o[0], o[1], o[2], o[3]);
                    reinterpret_cast<uint4*>(dst)[1] = make_uint4(o[4], o[5], o[6], o[7]);
                };
                uint4 rr[NR][2], er[NEX];
                bool ok[NR], eok[NEX];
#pragma unroll
                for (int u = 0; u < NR; ++u) {
                    const int hi = hi_s + u;
                    ok[u] = svalid && hi < Hin;
                    const unsigned off = ok[u] ? (unsigned)hi * inrow + colhi : lodelta;
                    rr[u][1] = *reinterpret_cast<const uint4*>(in0 + off);
                    rr[u][0] = *reinterpret_cast<const uint4*>(in0 + off - lodelta);
                }
#pragma unroll
                for (int e = 0; e < NEX; ++e) {
                    const int u = kc + e * NLC;
                    eok[e] = svalid && u < NR && hi_s + u < Hin;
                    er[e] = *reinterpret_cast<const uint4*>(in0 + (eok[e] ? (unsigned)(hi_s + u) * inrow + collast : 0u));
                }
                int sl = slot_s;
#pragma unroll
                for (int u = 0; u < NR; ++u) {
                    act_pack8(rr[u][0], rr[u][1], ok[u] ? cmask : 0u, splane_t + kc * PS + sl * rowdw);
                    sl = sl + 1 == RQ ? 0 : sl + 1;
                }
#pragma unroll
                for (int e = 0; e < NEX; ++e) {
                    const int u = kc + e * NLC;
                    if (u < NR) {
                        int se = slot_s + u; se = se >= RQ ? se - RQ : se;
                        act_pack8(er[e], er[e], eok[e] ? 0x0000ffffu : 0u, splane_t + (NPC - 1) * PS + se * rowdw);
                    }
                }
            }
            __builtin_amdgcn_s_setprio(0);
            wf_lds_barrier();
            // ---------------- walk down this chunk's output rows of the thread's output pair column (outputs 2jj, 2jj+1)
            const int o_lo = ST == 1 ? (s > 0 ? s - 1 : 0) : s;
            const int o_hi = ST == 1 ? (s + RB - 1 < Hout ? s + RB - 1 : Hout) : (s + RB < Hout ? s + RB : Hout);      // exclusive
            if (pvalid && o_lo < o_hi) {
                const unsigned* tc = tplane + (ST == 1 ? jj : 2 * jj) * PS;       // first ring pair of this output pair
                auto finish = [&](const int oy, const float* acc0, const float* acc1) {
                    T* dst = out0 + (unsigned)oy * outrow;
                    // round per channel as a PIXEL pair (x = 2jj, 2jj+1): the BatchNorm-2 sums of the values as stored are then two dot
                    // products per channel (pair . (1, 1) and pair . pair) instead of unpack + packed add + packed fma, and the stores'
                    // channel-paired dwords are one v_perm each (round 6: 16 single-rate instructions where there were 20, 8 of them packed)
                    unsigned q[4];
#pragma unroll
                    for (int i = 0; i < 4; ++i) q[i] = pk_bf16(acc0[i], acc1[i]);
                    const uint2 pk0 = make_uint2(__builtin_amdgcn_perm(q[1], q[0], 0x05040100u), __builtin_amdgcn_perm(q[3], q[2], 0x05040100u));
                    const uint2 pk1 = make_uint2(__builtin_amdgcn_perm(q[1], q[0], 0x07060302u), __builtin_amdgcn_perm(q[3], q[2], 0x07060302u));
                    *reinterpret_cast<uint2*>(dst) = pk0;
                    *reinterpret_cast<uint2*>(dst + a.C) = pk1;
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        st0[i] = wf_dot2(q[i], 0x3f803f80u, st0[i]);
                        st1[i] = wf_dot2(q[i], q[i], st1[i]);
                    }
                };
                if constexpr (ST == 1 && !QL) {
                    // ring slots of input rows o_lo - 1, o_lo, o_lo + 1
                    int sl0 = slot_s + (o_lo - s) - 1; sl0 = sl0 < 0 ? sl0 + RQ : sl0;
                    const int sl1 = sl0 + 1 == RQ ? 0 : sl0 + 1;
                    int sl2 = sl1 + 1 == RQ ? 0 : sl1 + 1;
                    uint4 tw[3][2];
                    tw[0][0] = *reinterpret_cast<const uint4*>(tc + sl0 * rowdw); tw[0][1] = *reinterpret_cast<const uint4*>(tc + sl0 * rowdw + PS);
                    tw[1][0] = *reinterpret_cast<const uint4*>(tc + sl1 * rowdw); tw[1][1] = *reinterpret_cast<const uint4*>(tc + sl1 * rowdw + PS);
                    auto row_step = [&](const int oy, const uint4 (&t0)[2], const uint4 (&t1)[2], uint4 (&t2)[2]) {
                        t2[0] = *reinterpret_cast<const uint4*>(tc + sl2 * rowdw);
                        t2[1] = *reinterpret_cast<const uint4*>(tc + sl2 * rowdw + PS);
                        sl2 = sl2 + 1 == RQ ? 0 : sl2 + 1;
                        float acc0[4], acc1[4];
#pragma unroll
                        for (int dy = 0; dy < 3; ++dy) {
                            const uint4 p0 = dy == 0 ? t0[0] : dy == 1 ? t1[0] : t2[0];
                            const uint4 p1 = dy == 0 ? t0[1] : dy == 1 ? t1[1] : t2[1];
                            const unsigned x0[4] = {p0.x, p0.y, p0.z, p0.w}, x1[4] = {p1.x, p1.y, p1.z, p1.w};
#pragma unroll
                            for (int q = 0; q < 4; ++q) {
                                acc0[q] = dy == 0 ? wf_dot2z(x0[q], wp[dy][0][q]) : wf_dot2(x0[q], wp[dy][0][q], acc0[q]);
                                acc0[q] = wf_dot2(x1[q], wp[dy][1][q], acc0[q]);
                                acc1[q] = dy == 0 ? wf_dot2z(x0[q], wp[dy][NWC - 2][q]) : wf_dot2(x0[q], wp[dy][NWC - 2][q], acc1[q]);
                                acc1[q] = wf_dot2(x1[q], wp[dy][NWC - 1][q], acc1[q]);
                            }
                        }
                        finish(oy, acc0, acc1);
                    };
                    for (int oy = o_lo; oy < o_hi; oy += 3) {
                        row_step(oy, tw[0], tw[1], tw[2]);
                        if (oy + 1 < o_hi) row_step(oy + 1, tw[1], tw[2], tw[0]);
                        if (oy + 2 < o_hi) row_step(oy + 2, tw[2], tw[0], tw[1]);
                    }
                } else if constexpr (ST == 1) {
                    // even-aligned pairs (rebuilt input): a row needs Q_{jj-1}.hi, Q_jj, Q_{jj+1}.lo — the two outer halves are kept as ONE
                    // dword per channel, E = (Q_{jj-1}.hi, Q_{jj+1}.lo) (one v_perm per channel and row; 8 instead of 12 window registers
                    // per row):  out[2jj] = E.(w0,0) + Q_jj.(w1,w2),  out[2jj+1] = Q_jj.(w0,w1) + E.(0,w2)
                    int sl0 = slot_s + (o_lo - s) - 1; sl0 = sl0 < 0 ? sl0 + RQ : sl0;
                    const int sl1 = sl0 + 1 == RQ ? 0 : sl0 + 1;
                    int sl2 = sl1 + 1 == RQ ? 0 : sl1 + 1;
                    uint4 tw[3][2];                            // [row][E, Q_jj]
                    auto ld_row = [&](const int sl, uint4 (&t)[2]) {
                        const uint4 qa = *reinterpret_cast<const uint4*>(tc + sl * rowdw);
                        t[1] = *reinterpret_cast<const uint4*>(tc + sl * rowdw + PS);
                        const uint4 qc = *reinterpret_cast<const uint4*>(tc + sl * rowdw + 2 * PS);
                        t[0] = make_uint4(__builtin_amdgcn_perm(qc.x, qa.x, 0x05040302u), __builtin_amdgcn_perm(qc.y, qa.y, 0x05040302u),
                                          __builtin_amdgcn_perm(qc.z, qa.z, 0x05040302u), __builtin_amdgcn_perm(qc.w, qa.w, 0x05040302u));
                    };
                    ld_row(sl0, tw[0]); ld_row(sl1, tw[1]);
                    auto row_step = [&](const int oy, const uint4 (&t0)[2], const uint4 (&t1)[2], uint4 (&t2)[2]) {
                        ld_row(sl2, t2);
                        sl2 = sl2 + 1 == RQ ? 0 : sl2 + 1;
                        float acc0[4], acc1[4];
#pragma unroll
                        for (int dy = 0; dy < 3; ++dy) {
                            const uint4 p0 = dy == 0 ? t0[0] : dy == 1 ? t1[0] : t2[0];
                            const uint4 p1 = dy == 0 ? t0[1] : dy == 1 ? t1[1] : t2[1];
                            const unsigned x0[4] = {p0.x, p0.y, p0.z, p0.w}, x1[4] = {p1.x, p1.y, p1.z, p1.w};
#pragma unroll
                            for (int q = 0; q < 4; ++q) {
                                acc0[q] = dy == 0 ? wf_dot2z(x0[q], wp[dy][0][q]) : wf_dot2(x0[q], wp[dy][0][q], acc0[q]);
                                acc0[q] = wf_dot2(x1[q], wp[dy][1][q], acc0[q]);
                                acc1[q] = dy == 0 ? wf_dot2z(x1[q], wp[dy][2][q]) : wf_dot2(x1[q], wp[dy][2][q], acc1[q]);
                                acc1[q] = wf_dot2(x0[q], wp[dy][3][q], acc1[q]);
                            }
                        }
                        finish(oy, acc0, acc1);
                    };
                    for (int oy = o_lo; oy < o_hi; oy += 3) {
                        row_step(oy, tw[0], tw[1], tw[2]);
                        if (oy + 1 < o_hi) row_step(oy + 1, tw[1], tw[2], tw[0]);
                        if (oy + 2 < o_hi) row_step(oy + 2, tw[2], tw[0], tw[1]);
                    }
                } else {
                    // output row o reads input rows 2o-1, 2o, 2o+1 = ring slots sa, sa+1, sa+2 (mod RQ), sa = slot_s - 1 + 2(o - s)
                    int sa = slot_s - 1; sa = sa < 0 ? sa + RQ : sa;
                    uint4 ta[3], tb[3], tcw[3];
#pragma unroll
                    for (int m = 0; m < 3; ++m) ta[m] = *reinterpret_cast<const uint4*>(tc + sa * rowdw + m * PS);
                    auto row_step = [&](const int oy, const uint4 (&r0)[3], uint4 (&r1)[3], uint4 (&r2)[3]) {
                        const int sb = sa + 1 >= RQ ? sa + 1 - RQ : sa + 1;
                        const int sc = sb + 1 >= RQ ? sb + 1 - RQ : sb + 1;
#pragma unroll
                        for (int m = 0; m < 3; ++m) {
                            r1[m] = *reinterpret_cast<const uint4*>(tc + sb * rowdw + m * PS);
                            r2[m] = *reinterpret_cast<const uint4*>(tc + sc * rowdw + m * PS);
                        }
                        sa = sc;
                        float acc0[4], acc1[4];
#pragma unroll
                        for (int dy = 0; dy < 3; ++dy) {
                            const uint4 p0 = dy == 0 ? r0[0] : dy == 1 ? r1[0] : r2[0];
                            const uint4 p1 = dy == 0 ? r0[1] : dy == 1 ? r1[1] : r2[1];
                            const uint4 p2 = dy == 0 ? r0[2] : dy == 1 ? r1[2] : r2[2];
                            const unsigned x0[4] = {p0.x, p0.y, p0.z, p0.w}, x1[4] = {p1.x, p1.y, p1.z, p1.w}, x2[4] = {p2.x, p2.y, p2.z, p2.w};
#pragma unroll
                            for (int q = 0; q < 4; ++q) {
                                acc0[q] = dy == 0 ? wf_dot2z(x0[q], wp[dy][0][q]) : wf_dot2(x0[q], wp[dy][0][q], acc0[q]);
                                acc0[q] = wf_dot2(x1[q], wp[dy][1][q], acc0[q]);
                                acc1[q] = dy == 0 ? wf_dot2z(x1[q], wp[dy][0][q]) : wf_dot2(x1[q], wp[dy][0][q], acc1[q]);
                                acc1[q] = wf_dot2(x2[q], wp[dy][1][q], acc1[q]);
                            }
                        }
                        finish(oy, acc0, acc1);
                    };
                    for (int oy = o_lo; oy < o_hi; oy += 2) {
                        row_step(oy, ta, tb, tcw);
                        if (oy + 1 < o_hi) row_step(oy + 1, tcw, tb, ta);
                    }
                }
            }
            wf_lds_barrier();
            slot_s += NR; slot_s = slot_s >= RQ ? slot_s - RQ : slot_s;
        }
    }
    if (a.stats) {
        const unsigned v[8] = {__float_as_uint(st0[0]), __float_as_uint(st0[1]), __float_as_uint(st0[2]), __float_as_uint(st0[3]),
                               __float_as_uint(st1[0]), __float_as_uint(st1[1]), __float_as_uint(st1[2]), __float_as_uint(st1[3])};
        float c4[4], d2[2];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const wf_u32x2_t r = __builtin_amdgcn_permlane16_swap(v[k], v[k + 4], false, false);
            c4[k] = __uint_as_float(r.x) + __uint_as_float(r.y);
        }
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const wf_u32x2_t r = __builtin_amdgcn_permlane32_swap(__float_as_uint(c4[k]), __float_as_uint(c4[k + 2]), false, false);
            d2[k] = __uint_as_float(r.x) + __uint_as_float(r.y);
        }
        const int r = lane >> 4;                          // lane row r holds value index (r&1)*4 + (r>>1)*2 + {0,1}
        DET_WAVES_BEGIN
#pragma unroll
        for (int k = 0; k < 2; ++k) atomicAdd(&lstat[(r & 1) * CS + cv * 4 + (r >> 1) * 2 + k], d2[k]);
        DET_WAVES_END
        __syncthreads();
        DET_ENTER();
        if (tid < 2 * CS) {
            const int which = tid / CS, c = c0 + tid % CS;
            if (c < a.C) stat_add(a.stats, (int)(blk.x % DWN_NREP), a.C, which, c, lstat[tid]);
        }
    }
    DET_EXIT();
}

// ------------------------------------------------------------------------------------------------
// launcher
// ------------------------------------------------------------------------------------------------
bool dw_spatial_fwd_walk_supported(const DwSpatialFwd& a, int dtype) {
    if (a.impl == 1 || dtype != DWN_BF16 || a.ks != 3 || a.C % 8) return false;      // impl 1: the pair / generic kernels (tests)
    if (a.stride != 1 && a.stride != 2) return false;
    if (a.Wout != 32 && a.Wout != 16 && a.Wout != 8) return false;
    if (a.Win != a.Wout * a.stride || a.Hout != (a.Hin - 1) / a.stride + 1) return false;
    if ((i64)a.Hin * a.Win * a.in.ld >= (1ll << 31)) return false;
    return true;
}

// rebuilt-input mode (a.a0 != NULL): bf16, Cin = 64 / 128, whole 64-channel slices (an MFMA needs every lane of the wave)
bool dw_spatial_fwd_rc_walk_supported(const DwSpatialFwd& a, int dtype) {
    if (!dw_spatial_fwd_walk_supported(a, dtype)) return false;
    if ((a.Cin != 64 && a.Cin != 128) || a.C % 64) return false;
    if (a.a0_ld % 8 || a.a0_ld < a.Cin) return false;
    if ((i64)a.Hin * a.Win * a.a0_ld >= (1ll << 31)) return false;
    return true;
}

template <int ST, int LPW, int RB, int CIN>
static int launch_fc(const DwSpatialFwd& a, hipStream_t s) {
    constexpr int NG = 16 / LPW, NPC = ST * LPW + (CIN > 0 ? 2 : 1), RQ = ST == 1 ? RB + 2 : 2 * RB + 1;
    const size_t lds = (size_t)NG * RQ * NPC * 256 + (CIN > 64 ? (size_t)64 * CIN * 2 : 0);
    auto kern = dw_spatial_fwd_chain_kernel<ST, LPW, RB, CIN>;
    if (lds > 48 * 1024 && hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
        (void)hipGetLastError();
    int bpc = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&bpc, kern, 256, lds) != hipSuccess || bpc < 1) { (void)hipGetLastError(); bpc = 2; }
    const int slices = (a.C + 63) / 64;
    const i64 work = (a.planes + NG - 1) / NG;
    i64 gx = (256 * bpc) / slices;
    if (gx < 1) gx = 1;
    if (gx > work) gx = work;
    if (CIN > 0) gx = gx >= 8 ? (gx & ~(i64)7) : 8;   // wf_block<true>: the slices of a plane group share an XCD
    hipLaunchKernelGGL(kern, dim3((unsigned)gx, slices), dim3(256), lds, s, a);
    DWN_CHECK_LAUNCH();
    return 0;
}
// output rows per chunk: a.rows_band, or the measured best at the metric shapes (tools/fwd_chain_check.py).  The rebuilt-input
// form stages whole input rows per wave: chunks of 4 or 8 input rows
template <int ST, int LPW, int CIN>
static int launch_fc_rb(const DwSpatialFwd& a, hipStream_t s) {
    if constexpr (CIN > 0) {
        const int rb = a.rows_band > 0 ? a.rows_band : (ST == 1 ? 4 : 2);
        if constexpr (ST == 1) {
            if (rb <= 4) return launch_fc<1, LPW, 4, CIN>(a, s);
            return launch_fc<1, LPW, 8, CIN>(a, s);
        } else {
            if (rb <= 2) return launch_fc<2, LPW, 2, CIN>(a, s);
            return launch_fc<2, LPW, 4, CIN>(a, s);
        }
    } else {
        const int rb = a.rows_band > 0 ? a.rows_band : (ST == 1 ? (LPW == 16 ? 4 : 6) : (LPW == 4 ? 1 : 2));
        if constexpr (ST == 1) {
            if (rb <= 2) return launch_fc<1, LPW, 2, 0>(a, s);
            if (rb <= 4) return launch_fc<1, LPW, 4, 0>(a, s);
            if (rb <= 6) return launch_fc<1, LPW, 6, 0>(a, s);
            return launch_fc<1, LPW, 8, 0>(a, s);
        } else {
            if (rb <= 1) return launch_fc<2, LPW, 1, 0>(a, s);
            if (rb <= 2) return launch_fc<2, LPW, 2, 0>(a, s);
            if (rb <= 3) return launch_fc<2, LPW, 3, 0>(a, s);
            return launch_fc<2, LPW, 4, 0>(a, s);
        }
    }
}

template <int CIN>
static int launch_fwd_walk_c(const DwSpatialFwd& a, hipStream_t s) {
    if (a.stride == 1) {
        if (a.Wout == 32) return launch_fc_rb<1, 16, CIN>(a, s);
        if (a.Wout == 16) return launch_fc_rb<1, 8, CIN>(a, s);
        return launch_fc_rb<1, 4, CIN>(a, s);
    }
    if (a.Wout == 32) return launch_fc_rb<2, 16, CIN>(a, s);
    if (a.Wout == 16) return launch_fc_rb<2, 8, CIN>(a, s);
    return launch_fc_rb<2, 4, CIN>(a, s);
}

int launch_dw_spatial_fwd_walk(const DwSpatialFwd& a, hipStream_t s) {
    if (a.a0) {
        if (!dw_spatial_fwd_rc_walk_supported(a, DWN_BF16) || !a.w1)
            return dwn_set_error(-3, "dw_spatial_fwd: rebuilt-input mode needs bf16, Cin = 64 / 128, C % 64 == 0, w1 and a row-walk plane width");
        return a.Cin == 64 ? launch_fwd_walk_c<64>(a, s) : launch_fwd_walk_c<128>(a, s);
    }
    return launch_fwd_walk_c<0>(a, s);
}
