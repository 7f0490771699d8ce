// "XL" NN GEMM for the shapes where the 128x128 / 4-wave kernel of dwn_gemm.hip is bound by its own load path rather than
// by HBM or the matrix cores (round 3): C[M][N] = A[M][K] . B[N][K]^T, bf16 operands, plain loads, store (+ BatchNorm
// Σ/Σ²) epilogue — conv_pw of the 256-channel blocks (K = 256, N = 1792: src/models/dwiseneuro.py:91), the cortex layers
// and their data gradients (:207), the readout data gradient (:276 backward).
//
// Why another kernel: at K = 256 the 128x128 tile re-fills 128 KB of LDS per 4 k-steps of 32 MFMAs per wave with a one-k-tile
// prefetch: a tile took 8.6 us of which 1.2 us were MFMAs — four dependent load round trips.  Here
//   * the tile is 256 rows x BN (256 or 128) columns, 8 waves (2x4 or 4x2, 128x64 or 64x64 per wave): half the LDS fill
//     bytes per FLOP of two 128x128 tiles;
//   * both operands arrive by LDS-DMA (global_load_lds_dwordx4) into a two-stage ring, the loads of k-step s+1 — also
//     across tile boundaries — issued before the MFMAs of k-step s, the XOR bank swizzle applied on the source side;
//   * a workgroup owns one N-tile and a contiguous range of M-tiles (BatchNorm sums stay in registers until the end);
//   * the epilogue re-uses the stage that was just multiplied: each wave turns 8 KB of it into four 16x64 groups at a time and
//     leaves as whole 128-byte row segments while the next tile's first k-step is already in flight.
// Measured (tools/xl_check.py, profiles): 147456 x 1792 x 256: 321 -> 219 us (618 TFLOP/s), 40960 rows: 80 -> 60 us; in the step
// pw_fwd 1.82 -> 1.73 ms.  What bounds it now (in-kernel s_memtime stamps per phase): the LDS-DMA fill rate of a CU.  A k-step
// moves 64 KB into LDS; the second-dispatched waves spend 2200-3900 cycles ISSUING their eight global_load_lds (queue back-
// pressure), the first-dispatched ones wait 1100-2400 cycles for theirs to land, the 64 MFMAs per wave take 1600 — i.e.
// ~64 KB per ~3 us per CU = the ~25 GB/s per CU LDS-DMA cadence MI355X_MICROARCH.md lists for a streaming fill.  Ablation builds
// agree (no MFMAs / no stores / no A loads / no B loads: 200 / 187 / 181 / 194 of 243 us).  Fewer fill bytes per FLOP would need
// a tile that does not fit the LDS; the next lever is keeping the weight operand in registers across k-steps.
// MFMA operand roles as in dwn_gemm.hip (weights = A operand): acc[i][j][r] = C[m = i*16 + lr][n = j*16 + 4*lg + r].
#include "dwn_internal.h"
#include <stdlib.h>

typedef __attribute__((ext_vector_type(8))) short xl_bf16x8_t;
typedef __attribute__((ext_vector_type(4))) float xl_f32x4_t;

extern __shared__ __attribute__((aligned(16))) unsigned char xl_smem[];
__device__ __attribute__((aligned(16))) unsigned xl_zero_src[4] = {0u, 0u, 0u, 0u};      // source of k chunks past K

static __device__ __forceinline__ void xl_glds16(const void* gsrc, unsigned lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}
static __device__ __forceinline__ void xl_lds_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}
static __device__ __forceinline__ void xl_wait_vm0() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

struct XlArgs {
    const bf16_t* A; i64 lda;
    const bf16_t* B; i64 ldb;
    bf16_t* C; i64 ldc;
    int M, N, K, groups;
    double* stats; int stat_nchan;
};

template <int BN>
__global__ __launch_bounds__(512, 2) void gemm_nn_xl_kernel(const XlArgs g) {
    constexpr int BM = 256, BK = 64;
    constexpr int WN = BN / 64;                         // waves along N (4 or 2)
    constexpr int WM = 8 / WN;                          // waves along M (2 or 4)
    constexpr int RM = BM / WM / 16, RN = 4;            // 16x16 tiles per wave: 8x4 or 4x4
    constexpr unsigned A_BYTES = BM * 128u, B_BYTES = BN * 128u, STG = A_BYTES + B_BYTES;
    constexpr unsigned OFF_RED = 2 * STG;               // (the epilogue slabs alias the stage that was just multiplied)
    constexpr int GPP = STG >= 8 * 8192 ? 4 : 2;       // 16-row groups a wave's slab holds per epilogue pass (2 KB each)
    static_assert(STG >= 8 * GPP * 2048 && RM % GPP == 0, "a stage holds one epilogue slab per wave");
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lr = lane & 15, lg = lane >> 4;
    const int wm = wave / WN, wn = wave % WN;
    const int m_base = wm * (BM / WM), n_base = wn * 64;
    const unsigned lds0 = (unsigned)(size_t)xl_smem;
    float* lred = reinterpret_cast<float*>(xl_smem + OFF_RED);

    const int ntn = (g.N + BN - 1) / BN, ntm = (g.M + BM - 1) / BM;
    const int units = ntn * g.groups;
    const int G = (int)gridDim.x;                       // a multiple of 8: blocks b and b+8 share an XCD
    const int lid = ((int)blockIdx.x & 7) * (G >> 3) + ((int)blockIdx.x >> 3);
    const int nranges = G / units > 0 ? (G / units < ntm ? G / units : ntm) : 1;
    const int nitems = units * nranges;
    const int nk = (g.K + BK - 1) / BK;
    const int rin = lane >> 3, cch = lane & 7;          // DMA role: row inside an 8-row block, 16-byte chunk

    for (int item = lid; item < nitems; item += G) {
        const int unit = item % units, mr = item / units;      // consecutive lids (one XCD) share the M-range: A rows hit its L2
        const int grp = unit / ntn, nt = unit % ntn;
        const int n0 = nt * BN;
        const int mt_beg = (int)((i64)mr * ntm / nranges), mt_end = (int)((i64)(mr + 1) * ntm / nranges);
        if (mt_beg >= mt_end) continue;
        const bf16_t* Ag = g.A + (i64)grp * g.K;
        const bf16_t* Bg = g.B + (i64)grp * g.N * g.ldb;
        bf16_t* Cg = g.C + (i64)grp * g.N;
        const int nsteps = (mt_end - mt_beg) * nk;

        // Both operands by LDS-DMA one k-step ahead.  (A register-staged TWO k-steps ahead was tried: 231 -> 222 us at 147456 x
        // 1792 x 256 for 32 more VGPRs and a ds_write pass — in-kernel stamps show the wait for the loads is only ~170 of a
        // step's ~5200 cycles; what costs is issuing them, the epilogue and the barrier skew.)
        // Issue cost: 900-1200 cycles per step went into 8 load instructions' address arithmetic (runtime divisions by nk, 64-bit
        // row products, tail selects): full tiles now use one base pointer + a row-block stride formed once per item (B) / tile
        // (A), the (k-step, tile) pairs advance as counters, ragged tiles and the K tail take a uniform slow branch.
        constexpr int A_CH = BM / 64, B_CH = BN / 64;                        // LDS-DMA instructions per wave and operand
        const int klane = (cch ^ rin) << 3;                                  // source-side swizzle: LDS chunk cch holds k-chunk cch ^ (row & 7)
        const bool nfull = n0 + BN <= g.N;
        const bf16_t* b0 = Bg + (i64)(n0 + wave * 8 + rin < g.N ? n0 + wave * 8 + rin : g.N - 1) * g.ldb + klane;
        const i64 bstride = 64 * g.ldb, astride = 64 * g.lda;
        const bf16_t* a0p = Ag;
        int aptr_mt = -1;
        bool mfull = false;
        auto issue = [&](const int ks, const int mt, const int stage) {
            const unsigned sa = lds0 + (unsigned)stage * STG + (unsigned)wave * 1024u;
            const int k0 = ks * BK;
            if (mt != aptr_mt) {
                aptr_mt = mt; mfull = mt * BM + BM <= g.M;
                const int m = mt * BM + wave * 8 + rin < g.M ? mt * BM + wave * 8 + rin : g.M - 1;
                a0p = Ag + (i64)m * g.lda + klane;
            }
            const bool kfull = k0 + BK <= g.K;
            if (mfull && kfull) {
#pragma unroll
                for (int r = 0; r < A_CH; ++r)
                    xl_glds16(a0p + r * astride + k0, (unsigned)__builtin_amdgcn_readfirstlane((int)(sa + (unsigned)r * 8192u)));
            } else {
                const bool kok = k0 + klane < g.K;
#pragma unroll
                for (int r = 0; r < A_CH; ++r) {
                    int m = mt * BM + (r * 8 + wave) * 8 + rin;
                    m = m < g.M ? m : g.M - 1;
                    xl_glds16(kok ? (const void*)(Ag + (i64)m * g.lda + klane + k0) : (const void*)xl_zero_src,
                              (unsigned)__builtin_amdgcn_readfirstlane((int)(sa + (unsigned)r * 8192u)));
                }
            }
            if (nfull && kfull) {
#pragma unroll
                for (int r = 0; r < B_CH; ++r)
                    xl_glds16(b0 + r * bstride + k0, (unsigned)__builtin_amdgcn_readfirstlane((int)(sa + A_BYTES + (unsigned)r * 8192u)));
            } else {
                const bool kok = k0 + klane < g.K;
#pragma unroll
                for (int r = 0; r < B_CH; ++r) {
                    int n = n0 + (r * 8 + wave) * 8 + rin;
                    n = n < g.N ? n : g.N - 1;
                    xl_glds16(kok ? (const void*)(Bg + (i64)n * g.ldb + klane + k0) : (const void*)xl_zero_src,
                              (unsigned)__builtin_amdgcn_readfirstlane((int)(sa + A_BYTES + (unsigned)r * 8192u)));
                }
            }
        };
#define XL_ADV(KS, MT) do { if (++(KS) == nk) { (KS) = 0; ++(MT); } } while (0)

        xl_f32x4_t acc[RM][RN];
#pragma unroll
        for (int i = 0; i < RM; ++i)
#pragma unroll
            for (int j = 0; j < RN; ++j) acc[i][j] = xl_f32x4_t{0.f, 0.f, 0.f, 0.f};
        float s0[RN][4], s1[RN][4];
#pragma unroll
        for (int j = 0; j < RN; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) { s0[j][r] = 0.f; s1[j][r] = 0.f; }
        if (g.stats && tid < 2 * BN) lred[tid] = 0.f;

        int ks0 = 0, mt0 = mt_beg;                              // (k-step, tile) of the current step and of step+1
        int ks1 = 0, mt1 = mt_beg; XL_ADV(ks1, mt1);
        issue(0, mt_beg, 0);
        xl_wait_vm0();
        xl_lds_barrier();
        for (int step = 0; step < nsteps; ++step) {
            // the loads of step+1 are issued BETWEEN this step's MFMA groups, one instruction at a time: issued in one burst
            // at the top of the step, the second-dispatched waves sat 2200-3900 cycles in queue back-pressure before their
            // first MFMA (in-order issue) while the first-dispatched ones idled at the barrier (in-kernel stamps)
            bool fast = false;
            const bf16_t *pa = nullptr, *pb = nullptr;
            unsigned sdst = 0;
            if (step + 1 < nsteps) {
                const int k0 = ks1 * BK;
                if (mt1 != aptr_mt) {
                    aptr_mt = mt1; mfull = mt1 * BM + BM <= g.M;
                    const int m = mt1 * BM + wave * 8 + rin < g.M ? mt1 * BM + wave * 8 + rin : g.M - 1;
                    a0p = Ag + (i64)m * g.lda + klane;
                }
                fast = mfull && nfull && k0 + BK <= g.K;
                if (fast) { pa = a0p + k0; pb = b0 + k0; sdst = lds0 + (unsigned)((step + 1) & 1) * STG + (unsigned)wave * 1024u; }
                else issue(ks1, mt1, (step + 1) & 1);                    // ragged tile / K tail: everything up front
            }
            const unsigned char* tA = xl_smem + (step & 1) * STG;
            const unsigned char* tB = tA + A_BYTES;
            constexpr int NL = A_CH + B_CH, SP = (2 * RM) / NL;           // loads per wave and step, MFMA groups between two of them
#pragma unroll
            for (int kb = 0; kb < 2; ++kb) {
                const int chunk = kb * 4 + lg;
                uint4 af[RM], bfr[RN];
#pragma unroll
                for (int j = 0; j < RN; ++j) {
                    const int row = n_base + j * 16 + lr;
                    bfr[j] = *reinterpret_cast<const uint4*>(tB + row * 128 + ((chunk ^ (row & 7)) << 4));
                }
                // activation fragments two 16-row groups ahead of their MFMAs (all RM of them live at once spill at 256 VGPRs)
                auto lda = [&](const int i) {
                    const int row = m_base + i * 16 + lr;
                    return *reinterpret_cast<const uint4*>(tA + row * 128 + ((chunk ^ (row & 7)) << 4));
                };
                af[0] = lda(0);
                af[1] = lda(1);
#pragma unroll
                for (int i = 0; i < RM; ++i) {
                    if (i + 2 < RM) af[i + 2] = lda(i + 2);
#pragma unroll
                    for (int j = 0; j < RN; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(xl_bf16x8_t, bfr[j]),
                                                                            __builtin_bit_cast(xl_bf16x8_t, af[i]), acc[i][j], 0, 0, 0);
                    const int slot = kb * RM + i;
                    if (slot % SP == 0 && slot / SP < NL) {
                        const int q = slot / SP;
                        if (fast) {
                            if (q < A_CH) xl_glds16(pa + q * astride, (unsigned)__builtin_amdgcn_readfirstlane((int)(sdst + (unsigned)q * 8192u)));
                            else xl_glds16(pb + (q - A_CH) * bstride, (unsigned)__builtin_amdgcn_readfirstlane((int)(sdst + A_BYTES + (unsigned)(q - A_CH) * 8192u)));
                        }
                    }
                    __builtin_amdgcn_sched_barrier(0);          // pins the fragment-read / MFMA / load-issue interleave
                }
            }
            if (ks0 == nk - 1) {
                // ---------------- epilogue of tile mt0.  The stage just multiplied is free once every wave is past its MFMAs:
                // each wave turns 8 KB of it into four 16-row x 64-column groups (128-byte rows, 16-byte chunks XOR-swizzled by
                // the row), so a 128x64 wave tile leaves in two passes of 16 LDS writes, one wait, 8 reads and 8 whole-row
                // stores (per-16-row slabs cost 5500-9000 cycles per tile in wait round trips: in-kernel stamps).
                xl_lds_barrier();
                unsigned char* slab = xl_smem + (step & 1) * STG + wave * (GPP * 2048);
                const int mw = mt0 * BM + m_base;
#pragma unroll
                for (int i0 = 0; i0 < RM; i0 += GPP) {
#pragma unroll
                    for (int ii = 0; ii < GPP; ++ii) {
                        const int i = i0 + ii;
                        const bool rok = mw + i * 16 + lr < g.M;
#pragma unroll
                        for (int j = 0; j < RN; ++j) {
                            const unsigned p0 = pk_bf16(acc[i][j][0], acc[i][j][1]), p1 = pk_bf16(acc[i][j][2], acc[i][j][3]);
                            *reinterpret_cast<uint2*>(slab + ii * 2048 + lr * 128 + (((j * 2 + (lg >> 1)) ^ (lr & 7)) << 4) + (lg & 1) * 8) = make_uint2(p0, p1);
                            if (g.stats && rok) {
                                const float v0 = __uint_as_float(p0 << 16), v1 = __uint_as_float(p0 & 0xffff0000u);
                                const float v2 = __uint_as_float(p1 << 16), v3 = __uint_as_float(p1 & 0xffff0000u);
                                s0[j][0] += v0; s0[j][1] += v1; s0[j][2] += v2; s0[j][3] += v3;
                                s1[j][0] = fmaf(v0, v0, s1[j][0]); s1[j][1] = fmaf(v1, v1, s1[j][1]);
                                s1[j][2] = fmaf(v2, v2, s1[j][2]); s1[j][3] = fmaf(v3, v3, s1[j][3]);
                            }
                            acc[i][j] = xl_f32x4_t{0.f, 0.f, 0.f, 0.f};
                        }
                    }
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");       // the wave's own slab writes have landed
#pragma unroll
                    for (int h = 0; h < 2 * GPP; ++h) {
                        const int row = rin + 8 * (h & 1), ii = h >> 1;
                        const uint4 v = *reinterpret_cast<const uint4*>(slab + ii * 2048 + row * 128 + ((cch ^ (row & 7)) << 4));
                        const int m = mw + (i0 + ii) * 16 + row, n = n0 + n_base + cch * 8;
                        if (m < g.M && n < g.N) *reinterpret_cast<uint4*>(Cg + (i64)m * g.ldc + n) = v;
                    }
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");       // ... and its reads, before the next pass overwrites them
                }
            }
            xl_wait_vm0();                 // this wave's share of step+1 has landed
            xl_lds_barrier();              // everybody's has, and everybody is done reading this step's stage
            ks0 = ks1; mt0 = mt1; XL_ADV(ks1, mt1);
        }
        if (g.stats) {
            // fold the 16 row lanes, one LDS add per wave and column, then one fp64 atomic per column and workgroup
#pragma unroll
            for (int j = 0; j < RN; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    float a = s0[j][r], b = s1[j][r];
#pragma unroll
                    for (int o = 1; o < 16; o <<= 1) { a += __shfl_xor(a, o); b += __shfl_xor(b, o); }
                    if (lr == 0) {
                        atomicAdd(&lred[n_base + j * 16 + 4 * lg + r], a);
                        atomicAdd(&lred[BN + n_base + j * 16 + 4 * lg + r], b);
                    }
                }
            __syncthreads();
            if (tid < 2 * BN) {
                const int which = tid / BN, c = n0 + tid % BN;
                if (c < g.N) stat_add(g.stats, (int)(blockIdx.x % DWN_NREP), g.stat_nchan, which, grp * g.N + c, lred[tid]);
            }
            __syncthreads();
        }
    }
}

#undef XL_ADV

// shapes this kernel takes over (dwn_gemm_nn_args.variant: DWN_NN_XL128 / DWN_NN_XL256 force it where the arguments allow it)
bool gemm_nn_xl_eligible(const GemmNN& g, int dtype) {
#ifdef DWN_DETERMINISTIC
    return false;                                         // the deterministic build keeps to the kernels with ordered reductions
#endif
    if (g.variant == DWN_NN_TILE128) return false;
    if (dtype != DWN_BF16 || g.a_kind != LD_PLAIN || g.epi != EPI_STORE || g.b_sample_stride || g.a2) return false;
    if (g.K % 8 || g.N % 8 || g.a.ld % 8 || g.ldb % 8 || g.ldc % 8 || g.M < 1) return false;
    if (((size_t)g.a.p | (size_t)g.b | (size_t)g.c) & 15) return false;
    if (g.variant == DWN_NN_XL128 || g.variant == DWN_NN_XL256) return true;
    // measured (tools/xl_check.py): wins on the big-M, K = 256 expand convs of the 256-channel blocks (321 -> 222 us at
    // 147456 x 1792 x 256, 80 -> 66 us at 40960 rows); the M = 1024 cortex / readout-gradient shapes stay with the 128x128 kernel
    // (16-75 us there against 20-86 us here: too few tiles to amortise the 256-row pipeline's fill)
    return g.K >= 256 && g.N >= 512 && g.M >= 8192;
}

int launch_gemm_nn_xl(const GemmNN& g, hipStream_t s) {
    XlArgs a;
    a.A = reinterpret_cast<const bf16_t*>(g.a.p); a.lda = g.a.ld; a.B = reinterpret_cast<const bf16_t*>(g.b); a.ldb = g.ldb;
    a.C = reinterpret_cast<bf16_t*>(g.c); a.ldc = g.ldc; a.M = g.M; a.N = g.N; a.K = g.K; a.groups = g.groups;
    a.stats = g.stats; a.stat_nchan = g.stat_nchan;
    const i64 ntm = (g.M + 255) / 256;
    const i64 tiles256 = ntm * ((g.N + 255) / 256) * g.groups;
    const bool wide = g.variant != DWN_NN_AUTO ? g.variant == DWN_NN_XL256 : tiles256 >= 256;      // enough 256-column tiles to fill the chip
    const int grid = 256;                                                 // one workgroup per CU (LDS), a multiple of 8
    if (wide) {
        constexpr size_t lds = 2 * (256 * 128 + 256 * 128) + 2 * 256 * sizeof(float);
        auto kern = gemm_nn_xl_kernel<256>;
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
            (void)hipGetLastError();
        hipLaunchKernelGGL(kern, dim3(grid), dim3(512), lds, s, a);
    } else {
        constexpr size_t lds = 2 * (256 * 128 + 128 * 128) + 2 * 128 * sizeof(float);
        auto kern = gemm_nn_xl_kernel<128>;
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
            (void)hipGetLastError();
        hipLaunchKernelGGL(kern, dim3(grid), dim3(512), lds, s, a);
    }
    DWN_CHECK_LAUNCH();
    return 0;
}
