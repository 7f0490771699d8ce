// spat_covn_dw backward (reference: src/models/dwiseneuro.py:96-102), bf16 storage, 3x3 — "row walk" kernels.
//
//   dh1 = (dwS^T dL/dy2) * SiLU'(BN1(y1)),  dW += z1 (x) dL/dy2,  Σdh1, Σdh1·ŷ1      with dL/dy2 = A1*dh2 + A2*y2 + A3
//
// Same arithmetic as dw_spatial_bwd_pair_kernel (x-pair-packed gradient tile in LDS, v_dot2c_f32_bf16 taps) but organised
// so that almost no integer index math is left in the loops — on this part the stencils are VALU-issue-bound (SQ counters:
// one VALU instruction per 6-8 cycles and SIMD at 3.5 TB/s), and 40 % of the pair kernel's instructions were flat-index
// decodes (exact division, per-pixel bounds, masks):
//   * a thread owns ONE pixel-pair column of its plane and walks down the rows: addresses advance by constants, the column's
//     halo mask is a per-thread constant, row validity is a scalar;
//   * the three gradient-tile rows a centre row needs live in a register window that slides by one row per step (2 LDS
//     reads per step instead of 6);
//   * planes narrower than 32 pixels put several planes side by side in one tile (16 / (W/2) lane groups), so the 9x16 and
//     5x8 planes of the deep blocks fill every lane and one barrier serves a whole multi-plane tile;
//   * the weight-gradient partials are folded across a wave's four pixel lanes with v_permlane16/32_swap before the
//     (slow) LDS float atomics.
// dh1 is bit-identical to the pair kernel's (same dot2 order); dW and the BatchNorm sums differ in summation order only.
#include "dwn_internal.h"
#include <stdlib.h>
#include <type_traits>

#ifndef WK_PRIO
#define WK_PRIO 1          // issue priority of the staging / rebuild phase (0 = off)
#endif
#ifndef WK_MINW
#define WK_MINW 3          // waves per SIMD the walk kernels are compiled for (three 256-thread workgroups per CU)
#endif
#ifndef WK_MINW2
#define WK_MINW2 3         // ... the stride-2 kernel (a thread owns four pixels: more registers)
#endif
#ifndef WK_MINW_RC
#define WK_MINW_RC 2       // the y1-rebuilding forms keep 32 registers of W1 fragments: two workgroups per CU
#endif
#ifndef WK_MINW2_RC
#define WK_MINW2_RC 2      // ... its y1-rebuilding form (W1 fragments + accumulators on top)
#endif
#ifndef WK_S1C_PF
#define WK_S1C_PF 1        // stride-1 y1-rebuilding form, gradient rows of the NEXT chunk fetched under the current chunk's walk: 0 = never,
                           // 1 = at 128 input channels (measured -11 %: 284 -> 252 us at 1024 planes of 9 x 16, E = 896), 2 = always
                           // (64 input channels: 430 vs 429 us — that form is not waiting for these loads)
#endif

extern __shared__ __attribute__((aligned(16))) unsigned char wk_smem[];

typedef float wk_f2_t __attribute__((ext_vector_type(2)));
typedef __attribute__((ext_vector_type(8))) short wk_bf16x8_t;
typedef __attribute__((ext_vector_type(4))) float wk_f32x4_t;
__device__ __forceinline__ wk_f32x4_t wk_mfma(const uint4& a, const uint4& b, const wk_f32x4_t& c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(wk_bf16x8_t, a), __builtin_bit_cast(wk_bf16x8_t, b), c, 0, 0, 0);
}
typedef __attribute__((ext_vector_type(2))) __bf16 wk_bf16x2_t;
typedef unsigned wk_u32x2_t __attribute__((ext_vector_type(2)));

__device__ __forceinline__ float wk_dot2(unsigned a, unsigned b, float c) {
    return __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(wk_bf16x2_t, a), __builtin_bit_cast(wk_bf16x2_t, b), c, false);
}
// first tap of an accumulator: the three-operand form with the constant 0 as addend (hipcc selects v_dot2c_f32_bf16, which
// accumulates in place, for the builtin and spends a v_mov on zeroing every accumulator: 8 - 16 instructions per walk step)
__device__ __forceinline__ float wk_dot2z(unsigned a, unsigned b) {
    float r;
    asm("v_dot2_f32_bf16 %0, %1, %2, 0" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ void wk_unpack(const uint2& r, wk_f2_t& lo, wk_f2_t& hi) {
    lo = wk_f2_t{__uint_as_float(r.x << 16), __uint_as_float(r.x & 0xffff0000u)};
    hi = wk_f2_t{__uint_as_float(r.y << 16), __uint_as_float(r.y & 0xffff0000u)};
}
__device__ __forceinline__ uint2 wk_ld8(const bf16_t* p) { return *reinterpret_cast<const uint2*>(p); }

// fold n (even) per-thread partial sums over the 4 lanes {l, l+16, l+32, l+48} of a wave: afterwards lane row r = lane/16
// holds the totals of value indices j*4 + sel(r) ... see the caller; two transposing swaps halve the value count each.
// in: v[4k] ; out: o[k] where o[j] (j < k) is the total of v[4*?]: value index = j + k*((r&1)*2 + (r>>1))  (documented below)
template <int K>
__device__ __forceinline__ void wk_fold4(const float* v, float* o, int /*lane*/) {
    // v has 4K values; step 1 pairs (v[i], v[i + 2K]), step 2 pairs (c[i], c[i + K])
    float c[2 * K];
#pragma unroll
    for (int i = 0; i < 2 * K; ++i) {
        const wk_u32x2_t r = __builtin_amdgcn_permlane16_swap(__float_as_uint(v[i]), __float_as_uint(v[i + 2 * K]), false, false);
        c[i] = __uint_as_float(r.x) + __uint_as_float(r.y);      // rows 0/2: sums of v[i]; rows 1/3: sums of v[i + 2K]
    }
#pragma unroll
    for (int i = 0; i < K; ++i) {
        const wk_u32x2_t r = __builtin_amdgcn_permlane32_swap(__float_as_uint(c[i]), __float_as_uint(c[i + K]), false, false);
        o[i] = __uint_as_float(r.x) + __uint_as_float(r.y);      // rows 0,1: totals of c[i]; rows 2,3: totals of c[i + K]
    }
    // lane row r holds o[i] = total of value index i + K*(r>>1) + 2K*(r&1)
}

// ------------------------------------------------------------------------------------------------
// Gradient tile staging shared by both strides.  The tile covers output rows ho_lo .. ho_lo + rows_q - 1 of NG planes,
// x-pair-packed: tile[plane][row][k][c] = (g[wo = 2k-1], g[wo = 2k]) with g = A1*dh2 + A2*y2 + A3 rounded to bf16 and zero
// outside the plane.  A thread stages its own pair column k = jj for every row (constant halo mask, addresses advance by
// one output row), then the extra last column k = LPW = Wout/2 is spread over the rows.
// ------------------------------------------------------------------------------------------------
template <int LPW>
__device__ __forceinline__ void wk_stage(const DwSpatialBwd& a, unsigned* tile, const int grp, const int jj, const int cv,
                                         const int chs, const int psafe, const bool pvalid, const int ho_lo, const int rows_q,
                                         const int rows_qmax) {
    typedef bf16_t T;
    constexpr int CS = 64, Wqp = LPW + 1;
    const int Hout = a.Hout, Wout = a.Wout;
    const int rowdw = Wqp * CS;
    unsigned* tcol = tile + (grp * rows_qmax * Wqp + jj) * CS + cv * 4;
    const unsigned cmask = (jj > 0 ? 0x0000ffffu : 0u) | 0xffff0000u;          // pair jj = (wo = 2jj-1, wo = 2jj)
    const unsigned dyrow = (unsigned)Wout * (unsigned)a.dy.ld;                  // elements per gradient row
    float a1[4], a2[4], a3[4];
    ldc4(a.dy.v1 + chs, a1); ldc4(a.dy.v2 + chs, a2); ldc4(a.dy.v3 + chs, a3);
    const wk_f2_t a1v[2] = {wk_f2_t{a1[0], a1[1]}, wk_f2_t{a1[2], a1[3]}}, a2v[2] = {wk_f2_t{a2[0], a2[1]}, wk_f2_t{a2[2], a2[3]}};
    const wk_f2_t a3v[2] = {wk_f2_t{a3[0], a3[1]}, wk_f2_t{a3[2], a3[3]}};
    const i64 pbase = (i64)psafe * Hout * Wout * a.dy.ld + chs;
    const T* dp0 = reinterpret_cast<const T*>(a.dy.p) + pbase;
    const T* dq0 = reinterpret_cast<const T*>(a.dy.q) + pbase;
    auto affine_pack = [&](const uint2& plo, const uint2& qlo, const uint2& phi, const uint2& qhi) {
        wk_f2_t p0, p1, q0, q1, gl0, gl1, gh0, gh1;
        wk_unpack(plo, p0, p1); wk_unpack(qlo, q0, q1);
        gl0 = a1v[0] * p0 + (a2v[0] * q0 + a3v[0]); gl1 = a1v[1] * p1 + (a2v[1] * q1 + a3v[1]);
        wk_unpack(phi, p0, p1); wk_unpack(qhi, q0, q1);
        gh0 = a1v[0] * p0 + (a2v[0] * q0 + a3v[0]); gh1 = a1v[1] * p1 + (a2v[1] * q1 + a3v[1]);
        return make_uint4(pk_bf16(gl0.x, gh0.x), pk_bf16(gl0.y, gh0.y), pk_bf16(gl1.x, gh1.x), pk_bf16(gl1.y, gh1.y));
    };
    // own column: pixels (ho, 2jj-1) and (ho, 2jj) for every staged row, 4 rows of loads in flight
    constexpr int NB = 4;
    const unsigned colhi = (unsigned)(2 * jj) * (unsigned)a.dy.ld;
    const unsigned lodelta = jj > 0 ? (unsigned)a.dy.ld : 0u;          // jj == 0: the low pixel is the halo (masked)
    for (int r0 = 0; r0 < rows_q; r0 += NB) {
        uint2 rp[NB][2], rq[NB][2];
        bool ok[NB];
#pragma unroll
        for (int u = 0; u < NB; ++u) {
            const int ho = ho_lo + r0 + u;
            ok[u] = pvalid && r0 + u < rows_q && (unsigned)ho < (unsigned)Hout;
            const unsigned off = ok[u] ? (unsigned)ho * dyrow + colhi : lodelta;
            rp[u][1] = wk_ld8(dp0 + off); rq[u][1] = wk_ld8(dq0 + off);
            rp[u][0] = wk_ld8(dp0 + off - lodelta); rq[u][0] = wk_ld8(dq0 + off - lodelta);
        }
#pragma unroll
        for (int u = 0; u < NB; ++u) {
            if (r0 + u < rows_q) {
                uint4 o = affine_pack(rp[u][0], rq[u][0], rp[u][1], rq[u][1]);
                const unsigned m = ok[u] ? cmask : 0u;
                o.x &= m; o.y &= m; o.z &= m; o.w &= m;
                *reinterpret_cast<uint4*>(tcol + (r0 + u) * rowdw) = o;
            }
        }
    }
    // last pair column (wo = Wout-1, halo): row r = jj, jj + LPW, ... of this plane
    const unsigned collast = (unsigned)(Wout - 1) * (unsigned)a.dy.ld;
    for (int r = jj; r < rows_q; r += LPW) {
        const int ho = ho_lo + r;
        const bool okr = pvalid && (unsigned)ho < (unsigned)Hout;
        const unsigned off = okr ? (unsigned)ho * dyrow + collast : 0u;
        const uint2 p = wk_ld8(dp0 + off), q = wk_ld8(dq0 + off);
        uint4 o = affine_pack(p, q, p, q);
        const unsigned m = okr ? 0x0000ffffu : 0u;
        o.x &= m; o.y &= m; o.z &= m; o.w &= m;
        *reinterpret_cast<uint4*>(tile + ((grp * rows_qmax + r) * Wqp + LPW) * CS + cv * 4) = o;
    }
}

// ------------------------------------------------------------------------------------------------
// stride 1, CHAINED rows.  LPW = pixel pairs per plane row (Win == 2*LPW in {32, 16, 8}); NG = 16/LPW planes side by side in one
// tile.  dh1 is bit-identical to the pair kernel's (same dot2 order).  A banded version of this kernel made ~14 DEPENDENT memory
// round trips per 9-row tile and idled on latency; here:
//   * a workgroup walks a WHOLE plane group top to bottom in chunks of RB rows and keeps the gradient tile as a RING of
//     RB + 2 row slots: the two halo rows a chunk needs are the previous chunk's last rows, still in LDS -> no halo re-read;
//   * the chunk's y1 rows arrive by LDS-DMA (global_load_lds_dwordx4, no registers), issued BEFORE the staging loads so both
//     ride the same round trip; a wave reads back only the 1 KB blocks it fetched itself (pixels 8w..8w+7 of the 32-pixel
//     group row), placed 0,2,1,3,4,6,5,7 inside the block so that the two pixel lanes of a 32-lane group hit different
//     bank halves;
//   * staging fetches the chunk's RB rows (+ the halo column) in ONE batch.
// One dependent round trip per RB rows.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ void wk_glds16(const void* gsrc, unsigned lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}
__device__ __forceinline__ void wk_wait_vm0() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
// workgroup barrier for LDS hand-offs only: __syncthreads() also drains vmcnt, i.e. waits for the write acknowledgements of
// every dh1 store issued so far — per chunk, for nothing (nobody reads dh1 back)
__device__ __forceinline__ void wk_lds_barrier() {
#ifdef WK_FULL_BARRIER
    __syncthreads();
#else
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
#endif
}


// workgroup -> (plane-group start, channel slice).  XCD-aware form: the slices of one plane group are consecutive workgroups of ONE
// XCD (workgroups are dealt round-robin over the 8 XCDs), so what they all read (the block input a0, when y1 is rebuilt) is
// fetched into that XCD's L2 once (measured, profiles/r5_l2share_pmc.json: fabric reads halve).  Needs gridDim.x % 8 == 0.
struct WkBlk { int x, y; };
template <bool XCD>
__device__ __forceinline__ WkBlk wk_block() {
    WkBlk r;
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

// CIN > 0: y1 is NOT read — a thread's y1 values are REBUILT from the block input a0 (a.a0, Cin = CIN channels, seven times
// narrower) with v_mfma_f32_16x16x32_bf16 and W1 (a.w1, as rounded to bf16), inside the walk and without an LDS round trip (round 6;
// the stride-2 kernel's scheme): the MFMA runs with the PIXELS as its A operand — tile row 4 g + j = (pixel pair g of this wave's four,
// row parity j >> 1, x parity j & 1) of two consecutive rows — and W1 rows as B in the order 4 cv + n, so accumulator register j of
// channel tile n in lane (cv, g) IS y1[row r + (j >> 1)][x = 2 jj + (j & 1)][channel 4 cv + n]: the thread's own 2 x 4 values of two
// consecutive walk rows, as fp32 accumulators (not rounded to bf16: the forward stencil activates the same accumulators and the
// Gram-matrix statistics describe them).  A wave fetches its a0 fragments one chunk ahead (each tile's registers are refilled as
// soon as its MFMAs are issued) and keeps its W1 fragments in registers for the whole launch.  The slices of a plane group run on
// one XCD (wk_block), so a0 crosses the fabric once and the other slices find it in that L2.
template <int LPW, int RB, int CIN>
__global__ __launch_bounds__(256, CIN > 0 ? WK_MINW_RC : WK_MINW) void dw_spatial_bwd_s1c_kernel(const DwSpatialBwd a) {
    typedef bf16_t T;
    constexpr int NT = 256, CS = 64, NG = 16 / LPW, Wqp = LPW + 1, RQ = RB + 2, W = 2 * LPW;
    constexpr int KB = CIN > 0 ? CIN / 32 : 1;
    constexpr int rowdw = Wqp * CS;                                         // dwords between ring rows of one plane
    constexpr unsigned RING_BYTES = (unsigned)NG * RQ * Wqp * CS * 4u;
    constexpr int NEX = (RB + LPW - 1) / LPW;                              // halo-column rows a thread stages per chunk
    __shared__ float lstat[2 * CS];
    __shared__ __attribute__((aligned(16))) unsigned lwp[3 * 4 * CS];        // packed weights [dy][combo][channel]
    __shared__ __attribute__((aligned(16))) float lcoef[5 * CS];             // bn1 scale, shift; BatchNorm-2 backward A1, A2, A3 (re-read per phase: registers)
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int cv = tid & 15, pl = tid >> 4;
    const int grp = pl / LPW, jj = pl % LPW;
    const WkBlk blk = wk_block<(CIN > 0)>();
    const int c0 = blk.y * CS;
    const int chan = c0 + cv * 4;
    const bool chan_ok = chan < a.C;
    const int chs = chan_ok ? chan : 0;
    if (tid < 2 * CS) lstat[tid] = 0.f;
    for (int i = tid; i < 5 * CS; i += NT) {
        const int which = i / CS, c = c0 + i % CS;
        const float* src = which == 0 ? a.y1.v1 : which == 1 ? a.y1.v2 : which == 2 ? a.dy.v1 : which == 3 ? a.dy.v2 : a.dy.v3;
        lcoef[i] = c < a.C ? src[c] : 0.f;
    }
    for (int i = tid; i < 3 * CS; i += NT) {
        const int dy = i / CS, cc = i % CS, c = c0 + cc;
        float w0 = 0.f, w1 = 0.f, w2 = 0.f;
        if (c < a.C) { w0 = a.w[(i64)(dy * 3 + 0) * a.C + c]; w1 = a.w[(i64)(dy * 3 + 1) * a.C + c]; w2 = a.w[(i64)(dy * 3 + 2) * a.C + c]; }
        lwp[(dy * 4 + 0) * CS + cc] = pk_bf16(w2, w1);
        lwp[(dy * 4 + 1) * CS + cc] = pk_bf16(w0, 0.f);
        lwp[(dy * 4 + 2) * CS + cc] = pk_bf16(0.f, w2);
        lwp[(dy * 4 + 3) * CS + cc] = pk_bf16(w1, w0);
    }
    __syncthreads();

    float dwp[9][4];
#pragma unroll
    for (int k = 0; k < 9; ++k) { dwp[k][0] = dwp[k][1] = dwp[k][2] = dwp[k][3] = 0.f; }
    wk_f2_t sp0[2] = {wk_f2_t{0.f, 0.f}, wk_f2_t{0.f, 0.f}}, sp1[2] = {wk_f2_t{0.f, 0.f}, wk_f2_t{0.f, 0.f}};

    const int Hin = a.Hin;                          // stride 1: Hout == Hin, Wout == Win == W
    const int ngroups = (a.planes + NG - 1) / NG;
    const int nchunks = (Hin + RB) / RB;            // chunks cover rows 0 .. Hin (row Hin = the zero row below the plane)
    T* dhp = reinterpret_cast<T*>(a.dh1);
    const T* y1p = reinterpret_cast<const T*>(a.y1.p);
    unsigned* tile = reinterpret_cast<unsigned*>(wk_smem);        // ring: [NG][RQ][Wqp][64] dwords
    unsigned* tcol = tile + (grp * RQ * Wqp + jj) * CS + cv * 4;              // this thread's pair column, ring slot 0
    unsigned* tlast = tile + (grp * RQ * Wqp + LPW) * CS + cv * 4;           // the halo pair column (wo = W-1 | outside), slot 0
    const unsigned y1row = (unsigned)W * (unsigned)a.y1.ld, dhrow = (unsigned)W * (unsigned)a.C;
    const unsigned dyrow = (unsigned)W * (unsigned)a.dy.ld;
    // ---- y1 by LDS-DMA: lane -> (pixel slot, 16-byte channel chunk) of this wave's 1 KB block of a group row
    const int dslot = lane >> 3, c16 = lane & 7;
    const int dq = (dslot & 4) | ((dslot & 1) << 1) | ((dslot >> 1) & 1);     // pixel of the block that lives in slot dslot
    const int dp = wave * 8 + dq;                                              // pixel of the 32-pixel group row
    const int dgrp = dp / W, dx = dp % W;
    const int dce = (c0 + c16 * 8 < a.C) ? c0 + c16 * 8 : c0;                   // channel tail: any valid address (never read back)
    const unsigned lds_y1 = (unsigned)(size_t)wk_smem + RING_BYTES + (unsigned)wave * 1024u;
    // ... and where this thread finds its two pixels (2jj, 2jj+1) of a row in that block: slots s0 and s0 + 2
    const int jq = pl & 3;
    const unsigned char* yld = wk_smem + RING_BYTES + wave * 1024 + (((jq & 1) + ((jq >> 1) << 2)) * 128) + cv * 8;
    // ---- rebuilt y1 (CIN > 0): lane (lr, lg): A-operand row lr = (pixel pair lr >> 2 of this wave, row parity (lr >> 1) & 1, x parity
    // lr & 1), k group lg; B-operand column lr = W1 row c0 + 4 lr + n of channel tile n
    const int lr = lane & 15, lg = lane >> 4;
    // W1 fragments: registers for CIN = 64; for CIN = 128 a swizzled copy of the 64-row slice in LDS behind the ring
    // (16-byte chunk c of row r at chunk c ^ ((r >> 2) & (chunks per row - 1)): conflict-free ds_read_b128 of a fragment at 128
    // input channels, two-way at 64 — eight reads per two walk rows)
    // PF (rebuilt form): the NEXT chunk's gradient rows are loaded before the current chunk's walk and stay in registers under it —
    // the kernel runs two workgroups per CU and was latency-bound (vector ALUs 52 % busy, a third of the wave cycles waiting): with
    // the loads a chunk ahead a workgroup never waits for HBM at the top of a chunk.  The 36 registers come from the W1 fragments,
    // which then live in LDS for 64 input channels too
    constexpr bool PF = (CIN > 64 && WK_S1C_PF >= 1) || (CIN > 0 && WK_S1C_PF >= 2);
    constexpr bool W1_LDS = CIN > 64 || PF;
    constexpr unsigned W1_OFF = CIN > 0 ? RING_BYTES : RING_BYTES + (unsigned)RB * 4096u;
    uint4 wfr[W1_LDS ? 1 : 4][KB];
    const T* a0src0 = nullptr;
    if constexpr (CIN > 0) {
        const T* w1 = reinterpret_cast<const T*>(a.w1);
        if constexpr (W1_LDS) {
            constexpr int CH = CIN / 8;
            for (int i = tid; i < 64 * CH; i += NT) {
                const int r = i / CH, c = i % CH;
                const int ch = c0 + r;
                *reinterpret_cast<uint4*>(wk_smem + W1_OFF + (r * CH + (c ^ ((r >> 2) & (CH - 1)))) * 16) =
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
    }
    auto w1frag = [&](const int n, const int kb) -> uint4 {
        if constexpr (W1_LDS) {
            const int r = 4 * lr + n;
            return *reinterpret_cast<const uint4*>(wk_smem + W1_OFF + (r * (CIN / 8) + ((lg + 4 * kb) ^ (lr & (CIN / 8 - 1)))) * 16);
        } else {
            return wfr[n][kb];
        }
    };
    (void)lr; (void)lg;
    // staging constants
    const unsigned cmask = (jj > 0 ? 0x0000ffffu : 0u) | 0xffff0000u;          // pair jj = (wo = 2jj-1, wo = 2jj)
    const unsigned colhi = (unsigned)(2 * jj) * (unsigned)a.dy.ld;
    const unsigned lodelta = jj > 0 ? (unsigned)a.dy.ld : 0u;          // jj == 0: the low pixel is the halo (masked)
    const unsigned collast = (unsigned)(W - 1) * (unsigned)a.dy.ld;

    // the staged rows of one chunk, raw: RB rows x (low, high) pixel of the thread's pair column + the halo column's rows
    uint2 st_rp[RB][2], st_rq[RB][2], st_ep[NEX], st_eq[NEX];
    auto stage_issue = [&](const T* dp0_, const T* dq0_, const int s_, const bool pv) {
#pragma unroll
        for (int u = 0; u < RB; ++u) {
            const int ho = s_ + u;
            const unsigned off = (pv && ho < Hin) ? (unsigned)ho * dyrow + colhi : lodelta;      // rows past the plane: any valid address (masked)
            st_rp[u][1] = wk_ld8(dp0_ + off); st_rq[u][1] = wk_ld8(dq0_ + off);
            st_rp[u][0] = wk_ld8(dp0_ + off - lodelta); st_rq[u][0] = wk_ld8(dq0_ + off - lodelta);
        }
#pragma unroll
        for (int e = 0; e < NEX; ++e) {
            const int ho = s_ + jj + e * LPW;
            const unsigned off = (pv && jj + e * LPW < RB && ho < Hin) ? (unsigned)ho * dyrow + collast : 0u;
            st_ep[e] = wk_ld8(dp0_ + off); st_eq[e] = wk_ld8(dq0_ + off);
        }
    };
    for (int pg = blk.x; pg < ngroups; pg += gridDim.x) {
        const int plane = pg * NG + grp;
        const bool pvalid = plane < a.planes && chan_ok;
        const int psafe = plane < a.planes ? plane : 0;
        const i64 pbase = (i64)psafe * Hin * W * a.dy.ld + chs;
        const T* dp0 = reinterpret_cast<const T*>(a.dy.p) + pbase;
        const T* dq0 = reinterpret_cast<const T*>(a.dy.q) + pbase;
        // the group this workgroup takes next (prefetch target of this group's last chunk)
        const int pgn = pg + (int)gridDim.x < ngroups ? pg + (int)gridDim.x : pg;
        const bool pvalid_n = pgn * NG + grp < a.planes && chan_ok;
        const i64 pbase_n = (i64)(pgn * NG + grp < a.planes ? pgn * NG + grp : 0) * Hin * W * a.dy.ld + chs;
        const int dplane = pg * NG + dgrp < a.planes ? pg * NG + dgrp : 0;
        const T* ysrc0 = y1p + ((i64)dplane * Hin * W + dx) * a.y1.ld + dce;
        if constexpr (CIN > 0) {
            const int tpl = wave * 4 + (lr >> 2);                                   // this lane's tile pixel: pair column of the group row
            const int tplane = pg * NG + tpl / LPW < a.planes ? pg * NG + tpl / LPW : 0;
            a0src0 = reinterpret_cast<const T*>(a.a0) + ((i64)tplane * Hin * W + 2 * (tpl % LPW) + (lr & 1)) * a.a0_ld + 8 * lg;
        }
        const i64 prow = (i64)psafe * Hin * W;
        const T* y10 = y1p + prow * a.y1.ld;      // (unused for loads: y1 comes from LDS)
        (void)y10;
        T* dh0 = dhp + prow * a.C + chan + (unsigned)(2 * jj) * (unsigned)a.C;
        // row -1 of the ring (slot 0) is the zero row above the plane
        *reinterpret_cast<uint4*>(tcol) = make_uint4(0, 0, 0, 0);
        if (jj == 0) *reinterpret_cast<uint4*>(tlast) = make_uint4(0, 0, 0, 0);
        int slot_s = 1;                            // ring slot of gradient row s = chunk * RB   (slot(r) = (r + 1) mod RQ)
        auto stage_finish = [&](const int s_, const int slot_, const bool pv) {
            const float4 c1 = *reinterpret_cast<const float4*>(&lcoef[2 * CS + cv * 4]);
            const float4 c2 = *reinterpret_cast<const float4*>(&lcoef[3 * CS + cv * 4]);
            const float4 c3 = *reinterpret_cast<const float4*>(&lcoef[4 * CS + cv * 4]);
            const wk_f2_t a1v[2] = {wk_f2_t{c1.x, c1.y}, wk_f2_t{c1.z, c1.w}}, a2v[2] = {wk_f2_t{c2.x, c2.y}, wk_f2_t{c2.z, c2.w}};
            const wk_f2_t a3v[2] = {wk_f2_t{c3.x, c3.y}, wk_f2_t{c3.z, c3.w}};
            auto affine_pack = [&](const uint2& plo, const uint2& qlo, const uint2& phi, const uint2& qhi) {
                wk_f2_t p0, p1, q0, q1, gl0, gl1, gh0, gh1;
                wk_unpack(plo, p0, p1); wk_unpack(qlo, q0, q1);
                gl0 = a1v[0] * p0 + (a2v[0] * q0 + a3v[0]); gl1 = a1v[1] * p1 + (a2v[1] * q1 + a3v[1]);
                wk_unpack(phi, p0, p1); wk_unpack(qhi, q0, q1);
                gh0 = a1v[0] * p0 + (a2v[0] * q0 + a3v[0]); gh1 = a1v[1] * p1 + (a2v[1] * q1 + a3v[1]);
                return make_uint4(pk_bf16(gl0.x, gh0.x), pk_bf16(gl0.y, gh0.y), pk_bf16(gl1.x, gh1.x), pk_bf16(gl1.y, gh1.y));
            };
            int sl = slot_;
#pragma unroll
            for (int u = 0; u < RB; ++u) {
                uint4 o = affine_pack(st_rp[u][0], st_rq[u][0], st_rp[u][1], st_rq[u][1]);
                const unsigned m = (pv && s_ + u < Hin) ? cmask : 0u;
                o.x &= m; o.y &= m; o.z &= m; o.w &= m;
                *reinterpret_cast<uint4*>(tcol + sl * rowdw) = o;
                sl = sl + 1 == RQ ? 0 : sl + 1;
            }
#pragma unroll
            for (int e = 0; e < NEX; ++e) {
                const int u = jj + e * LPW;
                if (u < RB) {
                    uint4 o = affine_pack(st_ep[e], st_eq[e], st_ep[e], st_eq[e]);
                    const unsigned m = (pv && s_ + u < Hin) ? 0x0000ffffu : 0u;
                    o.x &= m; o.y &= m; o.z &= m; o.w &= m;
                    int se = slot_ + u; se = se >= RQ ? se - RQ : se;
                    *reinterpret_cast<uint4*>(tlast + se * rowdw) = o;
                }
            }
        };
        uint4 afr[RB / 2][KB];                     // rebuilt form: a0 fragments of the chunk's rows, two rows per MFMA pixel tile
        auto load_afr_t = [&](const int s_, const int t) {        // tile t of chunk s_: rows s_ - 1 + 2 t, s_ + 2 t
            if constexpr (CIN > 0) {
                const unsigned a0row = (unsigned)W * (unsigned)a.a0_ld;
                int row = s_ - 1 + 2 * t + ((lr >> 1) & 1);                 // rows outside the plane: any valid address (never used)
                row = row < 0 ? 0 : (row >= Hin ? Hin - 1 : row);
#pragma unroll
                for (int kb = 0; kb < KB; ++kb)
                    afr[t][kb] = *reinterpret_cast<const uint4*>(a0src0 + (unsigned)row * a0row + 32 * kb);
            }
        };
        auto load_afr = [&](const int s_) {
            if constexpr (CIN > 0) {
#pragma unroll
                for (int t = 0; t < RB / 2; ++t) load_afr_t(s_, t);
            }
        };
        load_afr(0);
        if constexpr (PF) { if (pg == blk.x) stage_issue(dp0, dq0, 0, pvalid); }      // (later groups: issued under the previous group's last walk)
        for (int chunk = 0; chunk < nchunks; ++chunk) {
            const int s = chunk * RB;
            // staging + rebuild at raised issue priority: the CU's other workgroups are mostly in their walk, which can wait; this
            // phase ends in the barrier all four waves need (stand-alone -1.3 %; the forward kernel's gain is larger)
            __builtin_amdgcn_s_setprio(WK_PRIO);
            // ---------------- y1 rows s-1 .. s+RB-2 -> LDS (DMA, this wave's pixels only) — or the a0 rows they are rebuilt from
            // (rebuilt form: the fragments of this chunk were fetched under the previous chunk's walk)
            if constexpr (CIN == 0) {
#pragma unroll
                for (int i = 0; i < RB; ++i) {
                    const int row = s - 1 + i;
                    if ((unsigned)row < (unsigned)Hin)
                        wk_glds16(ysrc0 + (unsigned)row * y1row, (unsigned)__builtin_amdgcn_readfirstlane((int)(lds_y1 + (unsigned)i * 4096u)));
                }
            }
            // ---------------- stage dL/dy2 rows s .. s+RB-1 (BatchNorm-backward affine), x-pair-packed, into their ring slots
            if constexpr (!PF) stage_issue(dp0, dq0, s, pvalid);
            stage_finish(s, slot_s, pvalid);
            if constexpr (CIN == 0) wk_wait_vm0();  // this wave's y1 blocks have landed (it is their only reader)
            __builtin_amdgcn_s_setprio(0);
            wk_lds_barrier();
            if constexpr (PF) {
                // the next chunk's gradient rows (this group's, or the first chunk of the group this workgroup takes next; after the last
                // chunk of its last group: the same addresses again, unused) — in flight under the walk below
                const bool last = chunk + 1 == nchunks;
                stage_issue(reinterpret_cast<const T*>(a.dy.p) + (last ? pbase_n : pbase), reinterpret_cast<const T*>(a.dy.q) + (last ? pbase_n : pbase),
                            last ? 0 : s + RB, last ? pvalid_n : pvalid);
            }
            // ---------------- walk rows s-1 .. s+RB-2 of this thread's pixel-pair column
            const int r_lo = s > 0 ? s - 1 : 0;
            const int r_hi = s + RB - 1 < Hin ? s + RB - 1 : Hin;          // exclusive
            const float4 s4 = *reinterpret_cast<const float4*>(&lcoef[cv * 4]), t4 = *reinterpret_cast<const float4*>(&lcoef[CS + cv * 4]);
            const wk_f2_t bs2[2] = {wk_f2_t{s4.x, s4.y}, wk_f2_t{s4.z, s4.w}}, bt2[2] = {wk_f2_t{t4.x, t4.y}, wk_f2_t{t4.z, t4.w}};
            // one row of this thread's pixel pair: y[h][q] = y1 of pixel h, channel pair q (fp32)
            auto row_math = [&](const int r, const uint4 (&g0)[2], const uint4 (&g1)[2], const uint4 (&g2)[2], const wk_f2_t (&y)[2][2]) {
                wk_f2_t z1[2][2], dsl[2][2];
#pragma unroll
                for (int h = 0; h < 2; ++h) {
#pragma unroll
                    for (int q = 0; q < 2; ++q) {
                        const wk_f2_t hh = y[h][q] * bs2[q] + bt2[q];
                        const wk_f2_t sg = sigmoid2f_(hh);
                        z1[h][q] = hh * sg;
                        dsl[h][q] = sg * (1.0f + hh * (1.0f - sg));
                    }
                }
                const unsigned Z[4] = {pk_bf16(z1[0][0].x, z1[1][0].x), pk_bf16(z1[0][0].y, z1[1][0].y),
                                       pk_bf16(z1[0][1].x, z1[1][1].x), pk_bf16(z1[0][1].y, z1[1][1].y)};
                float dz0[4] = {0.f, 0.f, 0.f, 0.f}, dz1[4] = {0.f, 0.f, 0.f, 0.f};       // (wk_dot2z here costs 30 spilled registers)
#pragma unroll
                for (int dy = 0; dy < 3; ++dy) {
                    __builtin_amdgcn_sched_barrier(0);        // one stencil row's weight vectors live at a time (registers)
                    // (reading the tap weights one stencil row ahead, oldest gradient row first — 16 more registers, W1 moved to LDS for
                    // them — measured 431 vs 416 us on the rebuilt form: no gain, removed)
                    const uint4 G0 = dy == 0 ? g2[0] : dy == 1 ? g1[0] : g0[0];
                    const uint4 G1 = dy == 0 ? g2[1] : dy == 1 ? g1[1] : g0[1];
                    const uint4 Wa = *reinterpret_cast<const uint4*>(&lwp[(dy * 4 + 0) * CS + cv * 4]);
                    const uint4 Wb = *reinterpret_cast<const uint4*>(&lwp[(dy * 4 + 1) * CS + cv * 4]);
                    const uint4 Wc = *reinterpret_cast<const uint4*>(&lwp[(dy * 4 + 2) * CS + cv * 4]);
                    const uint4 Wd = *reinterpret_cast<const uint4*>(&lwp[(dy * 4 + 3) * CS + cv * 4]);
                    const unsigned ga[4] = {G0.x, G0.y, G0.z, G0.w}, gb[4] = {G1.x, G1.y, G1.z, G1.w};
                    const unsigned wa[4] = {Wa.x, Wa.y, Wa.z, Wa.w}, wb[4] = {Wb.x, Wb.y, Wb.z, Wb.w};
                    const unsigned wc[4] = {Wc.x, Wc.y, Wc.z, Wc.w}, wd[4] = {Wd.x, Wd.y, Wd.z, Wd.w};
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        dz0[q] = wk_dot2(ga[q], wa[q], dz0[q]);
                        dz0[q] = wk_dot2(gb[q], wb[q], dz0[q]);
                        dz1[q] = wk_dot2(ga[q], wc[q], dz1[q]);
                        dz1[q] = wk_dot2(gb[q], wd[q], dz1[q]);
                        const unsigned gm = __builtin_amdgcn_alignbit(gb[q], ga[q], 16);      // (G0.hi, G1.lo)
                        dwp[dy * 3 + 2][q] = wk_dot2(Z[q], ga[q], dwp[dy * 3 + 2][q]);
                        dwp[dy * 3 + 1][q] = wk_dot2(Z[q], gm, dwp[dy * 3 + 1][q]);
                        dwp[dy * 3 + 0][q] = wk_dot2(Z[q], gb[q], dwp[dy * 3 + 0][q]);
                    }
                }
                T* dst = dh0 + (unsigned)r * dhrow;
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const float* dz = h == 0 ? dz0 : dz1;
                    const wk_f2_t d0 = wk_f2_t{dz[0], dz[1]} * dsl[h][0], d1 = wk_f2_t{dz[2], dz[3]} * dsl[h][1];
                    const uint2 packed = make_uint2(pk_bf16(d0.x, d0.y), pk_bf16(d1.x, d1.y));
                    *reinterpret_cast<uint2*>(dst + h * a.C) = packed;
                    wk_f2_t r0, r1;
                    wk_unpack(packed, r0, r1);                    // statistics of the values as stored
                    sp0[0] += r0; sp0[1] += r1;
                    sp1[0] += r0 * y[h][0];
                    sp1[1] += r1 * y[h][1];
                }
            };
            if constexpr (CIN > 0) {
                // rows ri = 0 .. RB-1 (r = s - 1 + ri), unrolled: the window rotation and the MFMA tile of a row are compile-time; a row
                // outside [r_lo, r_hi) (row -1 of the first chunk, rows past the plane in the last) skips its math, not its window read
                int sl = slot_s - 2; sl = sl < 0 ? sl + RQ : sl;                 // ring slot of gradient row s - 2
                uint4 gw[3][2];
#pragma unroll
                for (int m = 0; m < 2; ++m) {
                    gw[m][0] = *reinterpret_cast<const uint4*>(tcol + sl * rowdw); gw[m][1] = *reinterpret_cast<const uint4*>(tcol + sl * rowdw + CS);
                    sl = sl + 1 == RQ ? 0 : sl + 1;
                }
                wk_f32x4_t acc[4];
#pragma unroll
                for (int ri = 0; ri < RB; ++ri) {
                    const int r = s - 1 + ri;
                    uint4 (&g0)[2] = gw[ri % 3], (&g1)[2] = gw[(ri + 1) % 3], (&g2)[2] = gw[(ri + 2) % 3];
                    g2[0] = *reinterpret_cast<const uint4*>(tcol + sl * rowdw);
                    g2[1] = *reinterpret_cast<const uint4*>(tcol + sl * rowdw + CS);
                    sl = sl + 1 == RQ ? 0 : sl + 1;
                    if ((ri & 1) == 0) {
                        // y1 of rows r, r + 1 for this wave's four pixel pairs: acc[n][2 dr + h] = y1[row r + dr][x = 2 jj + h][channel 4 cv + n]
#pragma unroll
                        for (int n = 0; n < 4; ++n) {
                            acc[n] = wk_f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                            for (int kb = 0; kb < KB; ++kb) acc[n] = wk_mfma(afr[ri >> 1][kb], w1frag(n, kb), acc[n]);
                        }
                        if (chunk + 1 < nchunks) load_afr_t(s + RB, ri >> 1);      // the next chunk's rows of this tile: in flight under the walk
                    }
                    if (pvalid && r >= r_lo && r < r_hi) {
                        const int d = 2 * (ri & 1);
                        const wk_f2_t y[2][2] = {{wk_f2_t{acc[0][d], acc[1][d]}, wk_f2_t{acc[2][d], acc[3][d]}},
                                                 {wk_f2_t{acc[0][d + 1], acc[1][d + 1]}, wk_f2_t{acc[2][d + 1], acc[3][d + 1]}}};
                        row_math(r, g0, g1, g2, y);
                    }
                }
            } else if (pvalid && r_lo < r_hi) {
                // ring slots of gradient rows r_lo - 1, r_lo, r_lo + 1
                int sl0 = slot_s + (r_lo - s) - 1; sl0 = sl0 < 0 ? sl0 + RQ : sl0;
                int sl1 = sl0 + 1 == RQ ? 0 : sl0 + 1;
                int sl2 = sl1 + 1 == RQ ? 0 : sl1 + 1;
                uint4 gw[3][2];
                gw[0][0] = *reinterpret_cast<const uint4*>(tcol + sl0 * rowdw); gw[0][1] = *reinterpret_cast<const uint4*>(tcol + sl0 * rowdw + CS);
                gw[1][0] = *reinterpret_cast<const uint4*>(tcol + sl1 * rowdw); gw[1][1] = *reinterpret_cast<const uint4*>(tcol + sl1 * rowdw + CS);
                auto row_step = [&](const int r, uint4 (&g0)[2], uint4 (&g1)[2], uint4 (&g2)[2]) {
                    g2[0] = *reinterpret_cast<const uint4*>(tcol + sl2 * rowdw);
                    g2[1] = *reinterpret_cast<const uint4*>(tcol + sl2 * rowdw + CS);
                    sl2 = sl2 + 1 == RQ ? 0 : sl2 + 1;
                    const unsigned char* yr = yld + (r - (s - 1)) * 4096;
                    const uint2 ry[2] = {*reinterpret_cast<const uint2*>(yr), *reinterpret_cast<const uint2*>(yr + 256)};
                    wk_f2_t y[2][2];
                    wk_unpack(ry[0], y[0][0], y[0][1]);
                    wk_unpack(ry[1], y[1][0], y[1][1]);
                    row_math(r, g0, g1, g2, y);
                };
                for (int r = r_lo; r < r_hi; r += 3) {
                    row_step(r, gw[0], gw[1], gw[2]);
                    if (r + 1 < r_hi) row_step(r + 1, gw[1], gw[2], gw[0]);
                    if (r + 2 < r_hi) row_step(r + 2, gw[2], gw[0], gw[1]);
                }
            }
            wk_lds_barrier();
            slot_s += RB; slot_s = slot_s >= RQ ? slot_s - RQ : slot_s;
        }
    }
    // ---------------- weight gradient: fold the wave's four pixel lanes (36 values -> 9 per lane), then LDS / global atomics
    float* lw = reinterpret_cast<float*>(wk_smem);               // [9][64], the ring is dead
    for (int i = tid; i < 9 * CS; i += NT) lw[i] = 0.f;
    __syncthreads();
    {
        float v[36], o[9];
#pragma unroll
        for (int k = 0; k < 9; ++k)
#pragma unroll
            for (int q = 0; q < 4; ++q) v[k * 4 + q] = dwp[k][q];
        wk_fold4<9>(v, o, lane);
        const int r = lane >> 4;
        DET_WAVES_BEGIN
#pragma unroll
        for (int i = 0; i < 9; ++i) {
            const int vi = i + 9 * (r >> 1) + 18 * (r & 1);          // value index = tap*4 + q
            atomicAdd(&lw[(vi >> 2) * CS + cv * 4 + (vi & 3)], o[i]);
        }
        DET_WAVES_END
    }
    __syncthreads();
    DET_ENTER();
    for (int i = tid; i < 9 * CS; i += NT) {
        const int k = i / CS, c = c0 + i % CS;
        if (c < a.C) atomicAdd(&a.dw[(i64)c * 9 + k], lw[i]);
    }
    if (a.stats) {
        float bm[4], bi[4];
        ldc4(a.y1.v3 + chs, bm); ldc4(a.y1.v4 + chs, bi);
        float v[8] = {sp0[0].x, sp0[0].y, sp0[1].x, sp0[1].y,
                      bi[0] * fmaf(-bm[0], sp0[0].x, sp1[0].x), bi[1] * fmaf(-bm[1], sp0[0].y, sp1[0].y),
                      bi[2] * fmaf(-bm[2], sp0[1].x, sp1[1].x), bi[3] * fmaf(-bm[3], sp0[1].y, sp1[1].y)}, o[2];
        wk_fold4<2>(v, o, lane);
        const int r = lane >> 4;
        DET_WAVES_BEGIN
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int vi = i + 2 * (r >> 1) + 4 * (r & 1);           // 0..3: Σdh1 of channel vi; 4..7: Σdh1·ŷ1 of channel vi-4
            atomicAdd(&lstat[(vi >> 2) * CS + cv * 4 + (vi & 3)], o[i]);
        }
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
// stride 2.  A thread owns one QUAD of input pixels (columns 4m .. 4m+3) of an input row and walks down the rows.
// With the output-column pairs Gq[k] = (g[wo = 2k-1], g[wo = 2k]) and M = (g[wo = 2m], g[wo = 2m+1]) = align(Gq[m], Gq[m+1]),
// a tap row (dy, output row ho) contributes
//   dz[4m]   = dot2(M, (w1, 0))      dz[4m+1] = dot2(M, (w2, w0))      dz[4m+2] = dot2(M, (0, w1))      dz[4m+3] = dot2(Gq[m+1], (w2, w0))
//   dW[dy][1] += dot2((z0, z2), M)   dW[dy][2] += dot2((z1, z3), M)    dW[dy][0] += dot2((z1, z3), Gq[m+1])
// Even input rows meet one tap row (dy = 1, ho = hi/2), odd rows two (dy = 0 at ho = (hi+1)/2, dy = 2 at ho = (hi-1)/2):
// the two gradient rows in use slide down in registers (one tile row read per two input rows).
// LPW = quads per input row = output pairs per row (Win == 4*LPW in {64, 32, 16}); NG = 16/LPW planes per tile.
// ------------------------------------------------------------------------------------------------
// CIN > 0: y1 is rebuilt from a0 (see the stride-1 kernel).  Here the MFMA runs with the PIXELS as its A operand (the wave's 16
// pixels of the input row: four quads) and W1 rows as B in the order 4 cv + n, so that accumulator register j of channel tile n
// in lane (cv, quad) IS y1[pixel j of the thread's quad][channel 4 cv + n]: the thread's own 4 x 4 values, no LDS round trip.
template <int LPW, int CIN>
__global__ __launch_bounds__(256, CIN > 0 ? WK_MINW2_RC : WK_MINW2) void dw_spatial_bwd_s2_kernel(const DwSpatialBwd a, const int R, const int rows_qmax) {
    typedef bf16_t T;
    constexpr int NT = 256, CS = 64, NG = 16 / LPW, Wqp = LPW + 1;
    constexpr int KB = CIN > 0 ? CIN / 32 : 1;
    __shared__ float lstat[2 * CS];
    __shared__ __attribute__((aligned(16))) unsigned lwp[3 * 3 * CS];        // packed weights [dy][combo][channel]
    __shared__ __attribute__((aligned(16))) float lcoef[2 * CS];             // BatchNorm-1 scale, shift (re-read per row: registers)
    const int tid = threadIdx.x, lane = tid & 63;
    const int cv = tid & 15, pl = tid >> 4;
    const int grp = pl / LPW, jj = pl % LPW;
    const WkBlk blk = wk_block<(CIN > 0)>();
    const int c0 = blk.y * CS;
    const int chan = c0 + cv * 4;
    const bool chan_ok = chan < a.C;
    const int chs = chan_ok ? chan : 0;
    if (tid < 2 * CS) {
        lstat[tid] = 0.f;
        const int c = c0 + (tid & (CS - 1));
        lcoef[tid] = c < a.C ? (tid < CS ? a.y1.v1[c] : a.y1.v2[c]) : 0.f;
    }
    for (int i = tid; i < 3 * CS; i += NT) {
        const int dy = i / CS, cc = i % CS, c = c0 + cc;
        float w0 = 0.f, w1 = 0.f, w2 = 0.f;
        if (c < a.C) { w0 = a.w[(i64)(dy * 3 + 0) * a.C + c]; w1 = a.w[(i64)(dy * 3 + 1) * a.C + c]; w2 = a.w[(i64)(dy * 3 + 2) * a.C + c]; }
        lwp[(dy * 3 + 0) * CS + cc] = pk_bf16(w1, 0.f);
        lwp[(dy * 3 + 1) * CS + cc] = pk_bf16(w2, w0);
        lwp[(dy * 3 + 2) * CS + cc] = pk_bf16(0.f, w1);
    }
    __syncthreads();

        // B operand: column cv of channel tile n = W1 row c0 + 4 cv + n, k group lane >> 4.  Registers for CIN = 64; for CIN = 128 a
    // swizzled LDS copy of the slice behind the gradient tile (chunk c of row r at c ^ ((r >> 2) & 15))
    constexpr bool W1_LDS = CIN > 64;
    const unsigned W1_OFF = (unsigned)rows_qmax * (unsigned)(NG * Wqp * 256);
    uint4 wfr[W1_LDS ? 1 : 4][KB];
    if constexpr (CIN > 0) {
        const T* w1 = reinterpret_cast<const T*>(a.w1);
        if constexpr (W1_LDS) {
            constexpr int CH = CIN / 8;
            for (int i = tid; i < 64 * CH; i += NT) {
                const int r = i / CH, c = i % CH;
                const int ch = c0 + r;
                *reinterpret_cast<uint4*>(wk_smem + W1_OFF + (r * CH + (c ^ ((r >> 2) & 15))) * 16) =
                    *reinterpret_cast<const uint4*>(w1 + (i64)(ch < a.C ? ch : c0) * CIN + 8 * c);
            }
            __syncthreads();
        } else {
#pragma unroll
            for (int n = 0; n < 4; ++n) {
                const int ch = chs + n;
#pragma unroll
                for (int kb = 0; kb < KB; ++kb)
                    wfr[n][kb] = *reinterpret_cast<const uint4*>(w1 + (i64)(ch < a.C ? ch : chs) * CIN + 8 * (lane >> 4) + 32 * kb);
            }
        }
    }
    auto w1frag = [&](const int n, const int kb) -> uint4 {
        if constexpr (W1_LDS) {
            const int r = 4 * cv + n;
            return *reinterpret_cast<const uint4*>(wk_smem + W1_OFF + (r * (CIN / 8) + (((lane >> 4) + 4 * kb) ^ cv)) * 16);
        } else {
            return wfr[n][kb];
        }
    };
    // A operand: tile pixel lane & 15 = (quad (lane & 15) >> 2 of this wave, pixel (lane & 15) & 3)
    const int apl = (tid >> 6) * 4 + ((lane & 15) >> 2);
    const int agrp = apl / LPW, ajj = apl % LPW;
    (void)agrp; (void)ajj;
    float dwp[9][4];
#pragma unroll
    for (int k = 0; k < 9; ++k) { dwp[k][0] = dwp[k][1] = dwp[k][2] = dwp[k][3] = 0.f; }
    wk_f2_t sp0[2] = {wk_f2_t{0.f, 0.f}, wk_f2_t{0.f, 0.f}}, sp1[2] = {wk_f2_t{0.f, 0.f}, wk_f2_t{0.f, 0.f}};   // Σdh1, Σdh1·y1

    const int Win = a.Win, Hin = a.Hin;             // Win == 4*LPW; R is even: bands start on even input rows
    const int nbands = (Hin + R - 1) / R;
    const int ngroups = (a.planes + NG - 1) / NG;
    const int ntiles = ngroups * nbands;
    T* dhp = reinterpret_cast<T*>(a.dh1);
    const T* y1p = reinterpret_cast<const T*>(a.y1.p);
    unsigned* tile = reinterpret_cast<unsigned*>(wk_smem);        // [NG][rows_qmax][Wqp][64] dwords
    const int rowdw = Wqp * CS;
    const unsigned* tcol = tile + (grp * rows_qmax * Wqp + jj) * CS + cv * 4;
    const unsigned y1row = (unsigned)Win * (unsigned)a.y1.ld, dhrow = (unsigned)Win * (unsigned)a.C;

    for (int tile_id = blk.x; tile_id < ntiles; tile_id += gridDim.x) {
        const int pg = tile_id / nbands, band = tile_id - pg * nbands;
        const int plane = pg * NG + grp;
        const bool pvalid = plane < a.planes && chan_ok;
        const int psafe = plane < a.planes ? plane : 0;
        const int hi0 = band * R;
        const int nri = (Hin - hi0 < R) ? Hin - hi0 : R;
        const int ho_lo = hi0 >> 1;
        const int rows_q = ((hi0 + nri) >> 1) - ho_lo + 1;
        __builtin_amdgcn_s_setprio(WK_PRIO);             // staging at raised issue priority (see the stride-1 kernel)
        wk_stage<LPW>(a, tile, grp, jj, cv, chs, psafe, pvalid, ho_lo, rows_q, rows_qmax);
        __builtin_amdgcn_s_setprio(0);
        __syncthreads();
        if (pvalid) {
            const i64 prow = (i64)plane * Hin * Win;
            const T* y10 = y1p + prow * a.y1.ld + chs + (unsigned)(hi0 * Win + 4 * jj) * (unsigned)a.y1.ld;
            T* dh0 = dhp + prow * a.C + chan + (unsigned)(hi0 * Win + 4 * jj) * (unsigned)a.C;
            const T* a0t = nullptr;
            if constexpr (CIN > 0) {
                const int aplane = pg * NG + agrp < a.planes ? pg * NG + agrp : 0;
                a0t = reinterpret_cast<const T*>(a.a0) + ((i64)aplane * Hin * Win + (unsigned)(hi0 * Win + 4 * ajj + (lane & 3))) * a.a0_ld + 8 * (lane >> 4);
            }
            const unsigned a0row = (unsigned)Win * (unsigned)(CIN > 0 ? a.a0_ld : 0);
            uint4 anext[KB];                              // a0 fragments one row ahead (rebuilt-y1 form)
            if constexpr (CIN > 0) {
#pragma unroll
                for (int kb = 0; kb < KB; ++kb) anext[kb] = *reinterpret_cast<const uint4*>(a0t + 32 * kb);
            }
            uint4 gA[2], gB[2];                          // gradient rows in use: pairs jj, jj+1
            gA[0] = *reinterpret_cast<const uint4*>(tcol); gA[1] = *reinterpret_cast<const uint4*>(tcol + CS);
            // one input row: NTAP tap rows (dy, gradient row) — even rows (dy 1, gcur), odd rows (dy 0, gnext) and (dy 2, gcur)
            auto row_step = [&](const int iy, auto odd_c, const uint4 (&gcur)[2], const uint4 (&gnext)[2]) {
                constexpr bool ODD = decltype(odd_c)::value;
                uint2 ry[4];
                uint4 afr[KB];
                if constexpr (CIN > 0) {
                    // this row's a0 fragments were fetched during the previous row step; the next row's go out now (rows past the band:
                    // the clamped address of the last row again — never used)
                    const int iyn = iy + 1 < nri ? iy + 1 : iy;
#pragma unroll
                    for (int kb = 0; kb < KB; ++kb) {
                        afr[kb] = anext[kb];
                        anext[kb] = *reinterpret_cast<const uint4*>(a0t + (unsigned)iyn * a0row + 32 * kb);
                    }
                } else {
                    // (fetching these one row ahead, as the rebuilt form does with its a0 fragments, changes nothing here — 1047 / 518 / 292
                    // vs 1044 / 519 / 279 us at two waves per SIMD, 1057 / 540 / 291 at three: this form moves 5 TB/s and is bound by that)
                    const T* yn = y10 + (unsigned)iy * y1row;
#pragma unroll
                    for (int p = 0; p < 4; ++p) ry[p] = wk_ld8(yn + p * a.y1.ld);       // in flight under the data-gradient taps
                }
                // 1) data-gradient taps (need only the gradient rows): dz of the four pixels
                float dz[4][4];
                // first tap in the three-operand form where registers allow (the stored-y1 form already spills at its budget)
                if constexpr (CIN == 0) {
#pragma unroll
                    for (int p = 0; p < 4; ++p) dz[p][0] = dz[p][1] = dz[p][2] = dz[p][3] = 0.f;
                }
                auto tap_dz = [&](const int dy, const uint4 (&g)[2], const bool first_) {
                    const bool first = first_ && CIN > 0;
                    const uint4 W0 = *reinterpret_cast<const uint4*>(&lwp[(dy * 3 + 0) * CS + cv * 4]);
                    const uint4 W1 = *reinterpret_cast<const uint4*>(&lwp[(dy * 3 + 1) * CS + cv * 4]);
                    const uint4 W2 = *reinterpret_cast<const uint4*>(&lwp[(dy * 3 + 2) * CS + cv * 4]);
                    const unsigned ga[4] = {g[0].x, g[0].y, g[0].z, g[0].w}, gb[4] = {g[1].x, g[1].y, g[1].z, g[1].w};
                    const unsigned w0[4] = {W0.x, W0.y, W0.z, W0.w}, w1[4] = {W1.x, W1.y, W1.z, W1.w}, w2[4] = {W2.x, W2.y, W2.z, W2.w};
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const unsigned gm = __builtin_amdgcn_alignbit(gb[q], ga[q], 16);      // (g[wo=2m], g[wo=2m+1])
                        dz[0][q] = first ? wk_dot2z(gm, w0[q]) : wk_dot2(gm, w0[q], dz[0][q]);
                        dz[1][q] = first ? wk_dot2z(gm, w1[q]) : wk_dot2(gm, w1[q], dz[1][q]);
                        dz[2][q] = first ? wk_dot2z(gm, w2[q]) : wk_dot2(gm, w2[q], dz[2][q]);
                        dz[3][q] = first ? wk_dot2z(gb[q], w1[q]) : wk_dot2(gb[q], w1[q], dz[3][q]);
                    }
                };
                if constexpr (ODD) { tap_dz(0, gnext, true); __builtin_amdgcn_sched_barrier(0); tap_dz(2, gcur, false); } else { tap_dz(1, gcur, true); }
                __builtin_amdgcn_sched_barrier(0);
                wk_f32x4_t acc[4];
                if constexpr (CIN > 0) {
                    // y1 of the thread's quad: acc[n][p] = y1[pixel p][channel 4 cv + n] — the fp32 accumulators themselves (round 6: not
                    // rounded to bf16; the forward stencil activates the same accumulators, the Gram statistics describe them)
#pragma unroll
                    for (int n = 0; n < 4; ++n) {
                        acc[n] = wk_f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                        for (int kb = 0; kb < KB; ++kb) acc[n] = wk_mfma(afr[kb], w1frag(n, kb), acc[n]);
                    }
                }
                (void)acc;
                // 2) activation, pixel by pixel (order 0, 2, 1, 3 so that (z0, z2) and (z1, z3) pack as soon as possible);
                //    each pixel is finished at once: dh1 = dz * SiLU', store, BatchNorm-backward sums
                T* dst = dh0 + (unsigned)iy * dhrow;
                const float4 s4 = *reinterpret_cast<const float4*>(&lcoef[cv * 4]), t4 = *reinterpret_cast<const float4*>(&lcoef[CS + cv * 4]);
                const wk_f2_t bs2[2] = {wk_f2_t{s4.x, s4.y}, wk_f2_t{s4.z, s4.w}}, bt2[2] = {wk_f2_t{t4.x, t4.y}, wk_f2_t{t4.z, t4.w}};
                wk_f2_t zk[2][2];
                unsigned Zev[4], Zod[4];
#pragma unroll
                for (int pp = 0; pp < 4; ++pp) {
                    const int p = pp == 0 ? 0 : pp == 1 ? 2 : pp == 2 ? 1 : 3;
                    wk_f2_t y[2], z[2], dsl[2];
                    if constexpr (CIN > 0) { y[0] = wk_f2_t{acc[0][p], acc[1][p]}; y[1] = wk_f2_t{acc[2][p], acc[3][p]}; }
                    else wk_unpack(ry[p], y[0], y[1]);
#pragma unroll
                    for (int q = 0; q < 2; ++q) {
                        const wk_f2_t hh = y[q] * bs2[q] + bt2[q];
                        const wk_f2_t sg = sigmoid2f_(hh);
                        z[q] = hh * sg;
                        dsl[q] = sg * (1.0f + hh * (1.0f - sg));
                    }
                    const wk_f2_t d0 = wk_f2_t{dz[p][0], dz[p][1]} * dsl[0], d1 = wk_f2_t{dz[p][2], dz[p][3]} * dsl[1];
                    const uint2 packed = make_uint2(pk_bf16(d0.x, d0.y), pk_bf16(d1.x, d1.y));
                    *reinterpret_cast<uint2*>(dst + p * a.C) = packed;
                    wk_f2_t r0, r1;
                    wk_unpack(packed, r0, r1);                    // statistics of the values as stored
                    sp0[0] += r0; sp0[1] += r1;
                    sp1[0] += r0 * y[0];
                    sp1[1] += r1 * y[1];
                    __builtin_amdgcn_sched_barrier(0);
                    if ((pp & 1) == 0) { zk[0][0] = z[0]; zk[0][1] = z[1]; }
                    else {
                        unsigned* Zp = pp == 1 ? Zev : Zod;
                        Zp[0] = pk_bf16(zk[0][0].x, z[0].x); Zp[1] = pk_bf16(zk[0][0].y, z[0].y);
                        Zp[2] = pk_bf16(zk[0][1].x, z[1].x); Zp[3] = pk_bf16(zk[0][1].y, z[1].y);
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
                // 3) weight-gradient taps
                auto tap_dw = [&](const int dy, const uint4 (&g)[2]) {
                    const unsigned ga[4] = {g[0].x, g[0].y, g[0].z, g[0].w}, gb[4] = {g[1].x, g[1].y, g[1].z, g[1].w};
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const unsigned gm = __builtin_amdgcn_alignbit(gb[q], ga[q], 16);
                        dwp[dy * 3 + 1][q] = wk_dot2(Zev[q], gm, dwp[dy * 3 + 1][q]);
                        dwp[dy * 3 + 2][q] = wk_dot2(Zod[q], gm, dwp[dy * 3 + 2][q]);
                        dwp[dy * 3 + 0][q] = wk_dot2(Zod[q], gb[q], dwp[dy * 3 + 0][q]);
                    }
                };
                if constexpr (ODD) { tap_dw(0, gnext); tap_dw(2, gcur); } else { tap_dw(1, gcur); }
            };
            using T_ = std::true_type;
            using F_ = std::false_type;
            // input rows in pairs (even, odd); gA = gradient row of the even row, gB = the next one
            for (int iy = 0; iy < nri; iy += 2) {
                row_step(iy, F_{}, gA, gB);
                if (iy + 1 < nri) {
                    gB[0] = *reinterpret_cast<const uint4*>(tcol + ((iy >> 1) + 1) * rowdw);
                    gB[1] = *reinterpret_cast<const uint4*>(tcol + ((iy >> 1) + 1) * rowdw + CS);
                    row_step(iy + 1, T_{}, gA, gB);
                    gA[0] = gB[0]; gA[1] = gB[1];
                }
            }
        }
        __syncthreads();
    }
    // ---------------- weight gradient: fold the wave's four pixel lanes (36 values -> 9 per lane), then LDS / global atomics
    float* lw = reinterpret_cast<float*>(wk_smem);               // [9][64], the gradient tile is dead
    for (int i = tid; i < 9 * CS; i += NT) lw[i] = 0.f;
    __syncthreads();
    {
        float v[36], o[9];
#pragma unroll
        for (int k = 0; k < 9; ++k)
#pragma unroll
            for (int q = 0; q < 4; ++q) v[k * 4 + q] = dwp[k][q];
        wk_fold4<9>(v, o, lane);
        const int r = lane >> 4;
        DET_WAVES_BEGIN
#pragma unroll
        for (int i = 0; i < 9; ++i) {
            const int vi = i + 9 * (r >> 1) + 18 * (r & 1);          // value index = tap*4 + q
            atomicAdd(&lw[(vi >> 2) * CS + cv * 4 + (vi & 3)], o[i]);
        }
        DET_WAVES_END
    }
    __syncthreads();
    DET_ENTER();
    for (int i = tid; i < 9 * CS; i += NT) {
        const int k = i / CS, c = c0 + i % CS;
        if (c < a.C) atomicAdd(&a.dw[(i64)c * 9 + k], lw[i]);
    }
    if (a.stats) {
        float bm[4], bi[4];
        ldc4(a.y1.v3 + chs, bm); ldc4(a.y1.v4 + chs, bi);
        float v[8] = {sp0[0].x, sp0[0].y, sp0[1].x, sp0[1].y,
                      bi[0] * fmaf(-bm[0], sp0[0].x, sp1[0].x), bi[1] * fmaf(-bm[1], sp0[0].y, sp1[0].y),
                      bi[2] * fmaf(-bm[2], sp0[1].x, sp1[1].x), bi[3] * fmaf(-bm[3], sp0[1].y, sp1[1].y)}, o[2];
        wk_fold4<2>(v, o, lane);
        const int r = lane >> 4;
        DET_WAVES_BEGIN
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int vi = i + 2 * (r >> 1) + 4 * (r & 1);
            atomicAdd(&lstat[(vi >> 2) * CS + cv * 4 + (vi & 3)], o[i]);
        }
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
#ifndef WK_LDS_BUDGET
#define WK_LDS_BUDGET (50 * 1024)          // three workgroups per CU
#endif

bool dw_spatial_bwd_walk_supported(const DwSpatialBwd& a, int dtype) {
    if (a.impl == 1 || dtype != DWN_BF16 || a.ks != 3 || a.C % 8) return false;      // impl 1: the pair / generic kernels (tests)
    if (a.stride == 1) {
        if (a.Win != 32 && a.Win != 16 && a.Win != 8) return false;
        if (a.Hout != a.Hin || a.Wout != a.Win) return false;
        if ((i64)a.Hin * a.Win * (a.dy.ld > a.y1.ld ? a.dy.ld : a.y1.ld) >= (1ll << 31)) return false;
        return true;
    }
    if (a.stride == 2) {
        if (a.Win != 64 && a.Win != 32 && a.Win != 16) return false;
        if (a.Hout != (a.Hin - 1) / 2 + 1 || a.Wout != a.Win / 2) return false;
        if ((i64)a.Hin * a.Win * (a.dy.ld > a.y1.ld ? a.dy.ld : a.y1.ld) >= (1ll << 31)) return false;
        return true;
    }
    return false;
}

// rebuilt-y1 mode (a.a0 != NULL): bf16, Cin = 64 (one or two... k-steps of 32 per MFMA tile; W1 fragments live in registers),
// whole 64-channel slices (an MFMA needs every lane of the wave: no channel tail)
bool dw_spatial_bwd_rc_supported(const DwSpatialBwd& a, int dtype) {
    if (!dw_spatial_bwd_walk_supported(a, dtype)) return false;
    if ((a.Cin != 64 && a.Cin != 128) || a.C % 64) return false;
    if (a.a0_ld % 8 || a.a0_ld < a.Cin) return false;
    if ((i64)a.Hin * a.Win * a.a0_ld >= (1ll << 31)) return false;
    return true;
}

template <int LPW, int CIN>
static int launch_s2(const DwSpatialBwd& a, hipStream_t s) {
    constexpr int NG = 16 / LPW, Wqp = LPW + 1;
    const size_t rowb = (size_t)NG * Wqp * 256;
    // rebuilt-y1 form: two workgroups per CU (registers), so the tile may be larger
    const size_t budget = CIN > 0 ? (size_t)(WK_LDS_BUDGET * 3 / 2) : (size_t)WK_LDS_BUDGET;
    int R = a.rows_band;                              // input rows per band, even
    if (R <= 0) {
        R = 2;
        while (R < a.Hin && (size_t)((R + 2) / 2 + 1) * rowb <= budget) R += 2;
        const int nb = (a.Hin + R - 1) / R;
        R = (a.Hin + nb - 1) / nb;                    // even split
    }
    R = (R + 1) & ~1;
    if (R > a.Hin) R = (a.Hin + 1) & ~1;
    const int rows_qmax = R / 2 + 1;
    size_t lds = (size_t)rows_qmax * rowb + (CIN > 64 ? (size_t)64 * CIN * 2 : 0);
    if (lds < 9 * 64 * sizeof(float)) lds = 9 * 64 * sizeof(float);
    if (lds > 150 * 1024) return dwn_set_error(-5, "dw_spatial_bwd: rows_band too large for the LDS tile");
    auto kern = dw_spatial_bwd_s2_kernel<LPW, CIN>;
    if (lds > 48 * 1024 && hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
        (void)hipGetLastError();
    int bpc = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&bpc, kern, 256, lds) != hipSuccess || bpc < 1) { (void)hipGetLastError(); bpc = 2; }
    const int slices = (a.C + 63) / 64;
    const int nbands = (a.Hin + R - 1) / R;
    const i64 work = (i64)((a.planes + NG - 1) / NG) * nbands;
    i64 gx = (256 * bpc) / slices;
    if (gx < 1) gx = 1;
    if (gx > work) gx = work;
    if (CIN > 0) gx = gx >= 8 ? (gx & ~(i64)7) : 8;   // wk_block<true>: the slices of a tile share an XCD (a workgroup past the work just exits)
    hipLaunchKernelGGL(kern, dim3((unsigned)gx, slices), dim3(256), lds, s, a, R, rows_qmax);
    DWN_CHECK_LAUNCH();
    return 0;
}

template <int LPW, int RB, int CIN>
static int launch_s1c(const DwSpatialBwd& a, hipStream_t s) {
    constexpr int NG = 16 / LPW, Wqp = LPW + 1;
    // ring + (stored form) the chunk's y1 rows by LDS-DMA / (rebuilt form, 128 input channels) the W1 slice
    const size_t lds = (size_t)NG * (RB + 2) * Wqp * 256 + (CIN > 0 ? ((CIN > 64 || WK_S1C_PF >= 2) ? (size_t)64 * CIN * 2 : 0) : (size_t)RB * 4096);
    auto kern = dw_spatial_bwd_s1c_kernel<LPW, RB, CIN>;
    if (lds > 48 * 1024 && hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
        (void)hipGetLastError();
    int bpc = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&bpc, kern, 256, lds) != hipSuccess || bpc < 1) { (void)hipGetLastError(); bpc = 2; }
    const int slices = (a.C + 63) / 64;
    const i64 work = (a.planes + NG - 1) / NG;
    i64 gx = (256 * bpc) / slices;
    if (gx < 1) gx = 1;
    if (gx > work) gx = work;
    if (CIN > 0) gx = gx >= 8 ? (gx & ~(i64)7) : 8;
    hipLaunchKernelGGL(kern, dim3((unsigned)gx, slices), dim3(256), lds, s, a);
    DWN_CHECK_LAUNCH();
    return 0;
}
// chained stride-1 kernel: rows per chunk = a.rows_band, or the measured best at the metric shapes (tools/bwd_chain_check.py):
// 4 rows per chunk at 18x32 planes (three workgroups per CU), 2 at 9x16 and 5x8
template <int LPW, int CIN>
static int launch_s1c_rb(const DwSpatialBwd& a, hipStream_t s) {
    const int rb = a.rows_band > 0 ? a.rows_band : (LPW == 16 ? 4 : 2);
    if (rb <= 2) return launch_s1c<LPW, 2, CIN>(a, s);
    if (rb <= 4) return launch_s1c<LPW, 4, CIN>(a, s);
    if (rb <= 6) return launch_s1c<LPW, 6, CIN>(a, s);
    return launch_s1c<LPW, 8, CIN>(a, s);
}

template <int CIN>
static int launch_walk_c(const DwSpatialBwd& a, hipStream_t s) {
    if (a.stride == 1) {
        if (a.Win == 32) return launch_s1c_rb<16, CIN>(a, s);
        if (a.Win == 16) return launch_s1c_rb<8, CIN>(a, s);
        return launch_s1c_rb<4, CIN>(a, s);
    }
    if (a.stride == 2) {
        if (a.Win == 64) return launch_s2<16, CIN>(a, s);
        if (a.Win == 32) return launch_s2<8, CIN>(a, s);
        return launch_s2<4, CIN>(a, s);
    }
    return dwn_set_error(-3, "dw_spatial_bwd_walk: unsupported configuration");
}

int launch_dw_spatial_bwd_walk(const DwSpatialBwd& a, hipStream_t s) {
    if (a.a0) {
        if (!dw_spatial_bwd_rc_supported(a, DWN_BF16) || !a.w1)
            return dwn_set_error(-3, "dw_spatial_bwd: rebuilt-y1 mode needs bf16, Cin = 64 / 128, C % 64 == 0, w1 and a row-walk plane width");
        return a.Cin == 64 ? launch_walk_c<64>(a, s) : launch_walk_c<128>(a, s);
    }
    return launch_walk_c<0>(a, s);
}
