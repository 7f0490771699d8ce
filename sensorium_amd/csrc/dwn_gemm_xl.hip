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
//   * the epilogue stages each wave's 16x64 sub-tile through a wave-private LDS slab (no workgroup barrier, the next
//     tile's loads stay in flight) and leaves as whole 128-byte row segments.
// Measured (tools/xl_check.py, profiles): 147456 x 1792 x 256: 321 -> 222 us (610 TFLOP/s), 40960 rows: 80 -> 66 us; in the step
// pw_fwd 1.82 -> 1.73 ms.  Ablation builds (no MFMAs / no stores / no A loads / no B loads: 200 / 187 / 181 / 194 of 243 us) say no
// single resource bounds it: what is left is the per-k-step barrier pipeline itself (~3 us per step at two waves per SIMD).
// MFMA operand roles as in dwn_gemm.hip (weights = A operand): acc[i][j][r] = C[m = i*16 + lr][n = j*16 + 4*lg + r].
#include "dwn_internal.h"
#include <stdlib.h>

typedef __attribute__((ext_vector_type(8))) short xl_bf16x8_t;
typedef __attribute__((ext_vector_type(4))) float xl_f32x4_t;
typedef __attribute__((ext_vector_type(4))) unsigned xl_u32x4_t;      // register staging (HIP's uint4 struct arrays went to scratch here)

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
    constexpr unsigned SLAB = 16 * 144;                 // wave-private epilogue slab: 16 rows x (128 + 16) bytes
    constexpr unsigned OFF_SLAB = 2 * STG, OFF_RED = OFF_SLAB + 8 * SLAB;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lr = lane & 15, lg = lane >> 4;
    const int wm = wave / WN, wn = wave % WN;
    const int m_base = wm * (BM / WM), n_base = wn * 64;
    const unsigned lds0 = (unsigned)(size_t)xl_smem;
    unsigned char* slab = xl_smem + OFF_SLAB + wave * SLAB;
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

        // B (weights: L2 hits) by LDS-DMA one k-step ahead; A (activations: first touch comes from HBM) through registers TWO
        // k-steps ahead — an LDS stage holds 64 KB, so with both operands on the ring only 32 KB of A per CU were in flight
        // (measured: 2.6 TB/s, ~3 us per k-step); the register set doubles that without another LDS stage
        auto issue_b = [&](const int step) {
            const int ks = step % nk;
            const unsigned sb = lds0 + (unsigned)(step & 1) * STG;
            const int k = ks * BK + ((cch ^ rin) << 3);                      // source-side swizzle: LDS chunk cch holds k-chunk cch ^ (row & 7)
            const bool kok = k < g.K;
#pragma unroll
            for (int r = 0; r < BN / 64; ++r) {
                const int rowblk = r * 8 + wave;
                int n = n0 + rowblk * 8 + rin;
                n = n < g.N ? n : g.N - 1;
                const void* src = kok ? (const void*)(Bg + (i64)n * g.ldb + k) : (const void*)xl_zero_src;
                xl_glds16(src, (unsigned)__builtin_amdgcn_readfirstlane((int)(sb + A_BYTES + (unsigned)rowblk * 1024u)));
            }
        };
        constexpr int A_CH = BM * 8 / 512;                                   // 16-byte chunks of the A tile per thread
        const int a_row = tid >> 3, a_kc = tid & 7;
#define XL_LOAD_A(STEP, RA) do { \
            const int mt_ = mt_beg + (STEP) / nk, k_ = ((STEP) % nk) * BK + a_kc * 8; \
            const bool kok_ = k_ < g.K; \
            _Pragma("unroll") for (int i_ = 0; i_ < A_CH; ++i_) { \
                int m_ = mt_ * BM + a_row + 64 * i_; \
                m_ = m_ < g.M ? m_ : g.M - 1; \
                RA[i_] = *reinterpret_cast<const xl_u32x4_t*>(kok_ ? (const void*)(Ag + (i64)m_ * g.lda + k_) : (const void*)xl_zero_src); \
            } } while (0)
#define XL_STORE_A(STEP, RA) do { \
            unsigned char* sA_ = xl_smem + ((STEP) & 1) * STG; \
            _Pragma("unroll") for (int i_ = 0; i_ < A_CH; ++i_) { \
                const int row_ = a_row + 64 * i_; \
                *reinterpret_cast<xl_u32x4_t*>(sA_ + row_ * 128 + ((a_kc ^ (row_ & 7)) << 4)) = RA[i_]; \
            } } while (0)

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

        xl_u32x4_t ra_cur[A_CH], ra_nxt[A_CH];                  // A of step+1 (loaded a step ago), A of step+2 (loading)
        issue_b(0);
        XL_LOAD_A(0, ra_cur);
        XL_STORE_A(0, ra_cur);
        if (nsteps > 1) XL_LOAD_A(1, ra_cur);
        xl_wait_vm0();
        xl_lds_barrier();
        for (int step = 0; step < nsteps; ++step) {
            if (step + 1 < nsteps) issue_b(step + 1);               // in flight under this step's MFMAs
            if (step + 2 < nsteps) XL_LOAD_A(step + 2, ra_nxt);     // ... and under the next step's too
            const unsigned char* tA = xl_smem + (step & 1) * STG;
            const unsigned char* tB = tA + A_BYTES;
#pragma unroll
            for (int kb = 0; kb < 2; ++kb) {
                const int chunk = kb * 4 + lg;
                uint4 af[RM], bfr[RN];
#pragma unroll
                for (int j = 0; j < RN; ++j) {
                    const int row = n_base + j * 16 + lr;
                    bfr[j] = *reinterpret_cast<const uint4*>(tB + row * 128 + ((chunk ^ (row & 7)) << 4));
                }
#pragma unroll
                for (int i = 0; i < RM; ++i) {
                    const int row = m_base + i * 16 + lr;
                    af[i] = *reinterpret_cast<const uint4*>(tA + row * 128 + ((chunk ^ (row & 7)) << 4));
                }
#pragma unroll
                for (int i = 0; i < RM; ++i)
#pragma unroll
                    for (int j = 0; j < RN; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(xl_bf16x8_t, bfr[j]),
                                                                            __builtin_bit_cast(xl_bf16x8_t, af[i]), acc[i][j], 0, 0, 0);
            }
            if (step % nk == nk - 1) {
                // ---------------- epilogue of tile mt: 16 rows at a time through this wave's slab
                const int mt = mt_beg + step / nk;
                const int mw = mt * BM + m_base;
#pragma unroll
                for (int i = 0; i < RM; ++i) {
                    const bool rok = mw + i * 16 + lr < g.M;
#pragma unroll
                    for (int j = 0; j < RN; ++j) {
                        const unsigned p0 = pk_bf16(acc[i][j][0], acc[i][j][1]), p1 = pk_bf16(acc[i][j][2], acc[i][j][3]);
                        *reinterpret_cast<uint2*>(slab + lr * 144 + j * 32 + lg * 8) = make_uint2(p0, p1);
                        if (g.stats && rok) {
                            const float v0 = __uint_as_float(p0 << 16), v1 = __uint_as_float(p0 & 0xffff0000u);
                            const float v2 = __uint_as_float(p1 << 16), v3 = __uint_as_float(p1 & 0xffff0000u);
                            s0[j][0] += v0; s0[j][1] += v1; s0[j][2] += v2; s0[j][3] += v3;
                            s1[j][0] = fmaf(v0, v0, s1[j][0]); s1[j][1] = fmaf(v1, v1, s1[j][1]);
                            s1[j][2] = fmaf(v2, v2, s1[j][2]); s1[j][3] = fmaf(v3, v3, s1[j][3]);
                        }
                        acc[i][j] = xl_f32x4_t{0.f, 0.f, 0.f, 0.f};
                    }
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");       // the wave's own slab writes have landed
#pragma unroll
                    for (int h = 0; h < 2; ++h) {
                        const int row = rin + 8 * h;
                        const uint4 v = *reinterpret_cast<const uint4*>(slab + row * 144 + cch * 16);
                        const int m = mw + i * 16 + row, n = n0 + n_base + cch * 8;
                        if (m < g.M && n < g.N) *reinterpret_cast<uint4*>(Cg + (i64)m * g.ldc + n) = v;
                    }
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");       // ... and its reads, before the next 16 rows overwrite them
                }
            }
            if (step + 1 < nsteps) XL_STORE_A(step + 1, ra_cur);    // that stage was last read a step ago (barrier since)
#pragma unroll
            for (int i = 0; i < A_CH; ++i) ra_cur[i] = ra_nxt[i];
            // this wave's B blocks of step+1 have landed; the A loads of step+2 (issued after them) stay in flight except behind
            // an epilogue's stores
            if (step % nk == nk - 1 || step + 2 >= nsteps) xl_wait_vm0();
            else asm volatile("s_waitcnt vmcnt(%0)" :: "n"(A_CH) : "memory");
            xl_lds_barrier();              // everybody's has, and everybody is done reading this step's stage
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

#undef XL_LOAD_A
#undef XL_STORE_A

// shapes this kernel takes over (DWN_NN_XL=0 never, =1 whenever the arguments allow it)
bool gemm_nn_xl_eligible(const GemmNN& g, int dtype) {
    const char* e = getenv("DWN_NN_XL");                  // read per call: A/B inside one process
    if (e && e[0] == '0') return false;
    if (dtype != DWN_BF16 || g.a_kind != LD_PLAIN || g.epi != EPI_STORE || g.b_sample_stride || g.a2) return false;
    if (g.K % 8 || g.N % 8 || g.a.ld % 8 || g.ldb % 8 || g.ldc % 8 || g.M < 1) return false;
    if (((size_t)g.a.p | (size_t)g.b | (size_t)g.c) & 15) return false;
    if (e && e[0] == '1') return true;
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
    const char* fbn = getenv("DWN_NN_XL_BN");
    const bool wide = fbn ? atoi(fbn) == 256 : tiles256 >= 256;          // enough 256-column tiles to fill the chip
    const int grid = 256;                                                 // one workgroup per CU (LDS), a multiple of 8
    if (wide) {
        constexpr size_t lds = 2 * (256 * 128 + 256 * 128) + 8 * 16 * 144 + 2 * 256 * sizeof(float);
        auto kern = gemm_nn_xl_kernel<256>;
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
            (void)hipGetLastError();
        hipLaunchKernelGGL(kern, dim3(grid), dim3(512), lds, s, a);
    } else {
        constexpr size_t lds = 2 * (256 * 128 + 128 * 128) + 8 * 16 * 144 + 2 * 128 * sizeof(float);
        auto kern = gemm_nn_xl_kernel<128>;
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
            (void)hipGetLastError();
        hipLaunchKernelGGL(kern, dim3(grid), dim3(512), lds, s, a);
    }
    DWN_CHECK_LAUNCH();
    return 0;
}
