// gemm_kd: C[M][N] = load(A)[M][K] . B[N][K]^T for the DEEP-K, NARROW-N point-wise shapes (bf16): the project conv conv_pwl
// (K = 7 C, N = C: src/models/dwiseneuro.py:117-120) with the SE gate applied to the activation, and conv_pw's data gradient
// (K = 7 C + C K-concatenated, N = C: dwiseneuro.py:90-91 backward) of the 128- and 256-channel blocks.
//
// Why another kernel.  gemm_nn_kernel moves BOTH operands through LDS (registers -> ds_write, or the LDS-DMA ring).  At these
// shapes the launch is bound by what a CU can pull through its load path (~23 GB/s per CU streamed from HBM, ~70 from L2:
// MI355X_MICROARCH.md "Indexed rows"), and with 128-column tiles the 256-column outputs read every A row twice.  Here
//   * A fragments go global -> VGPR in MFMA operand layout (lane (lr, lg) = row lr, k = 8 lg .. 8 lg + 7 of a 32-deep k-step: one
//     16-byte load) four k-steps ahead in a register ring, no LDS write and no LDS read for the big operand; the SE gate (fp32 row
//     of the tile's sample, staged once in LDS) multiplies the fragment in registers, rounded exactly as DWN_LD_GATE rounds;
//   * only the weights pass through LDS: BN rows x 64 bytes per k-step, registers -> ds_write two k-steps ahead, two stages, one
//     LDS-only barrier (no vector-memory drain) per k-step; chunk XOR so that fragment reads and staging writes are conflict-free;
//   * a workgroup (4 waves) owns 128 rows x BN columns, a wave 32 rows x BN columns (every weight fragment read from LDS feeds two
//     MFMAs).  BN = 256 (one pass over A, two workgroups per CU) where that still gives >= 512 tiles, else BN = 128 (three per CU).
// What was tried and made no difference (so the limit is the load path, not the schedule): A loads of two k-steps issued back to
// back (L1 line reuse); an eight-k-step ring loaded four steps at a time (DRAM page locality; costs a wave per SIMD: slower);
// weight fragments read 4 or 8 at a time instead of one (the dependent LDS round trips of a k-step); the bank-conflict-free layout.
// MFMA operand roles are swapped as in gemm_nn_kernel (weights = A operand): a lane's four accumulator registers are four
// consecutive output channels of one row.  The epilogue stages 64 rows at a time through LDS, leaves as 16-byte row segments and
// accumulates the next BatchNorm's sums from the rounded values (one flush per workgroup: LDS atomics, then fp64 atomics into
// the DWN_NREP replicas).  Results equal gemm_nn_kernel's to the bit (same k order, same roundings): tests/test_gpu_gemm.py.
#include "dwn_internal.h"
#include <type_traits>

#ifndef KD_FRG
#define KD_FRG 2
#endif
#ifndef KD_FRG256
#define KD_FRG256 1           // (256 columns: 128 accumulator registers leave no room: 2 spills 6-24 registers)
#endif
#ifndef KD_WIDE
#define KD_WIDE 1
#endif
#ifndef KD_MINW128
#define KD_MINW128 3
#endif
typedef __attribute__((ext_vector_type(8))) short kd_bf16x8_t;
typedef __attribute__((ext_vector_type(4))) float kd_f32x4_t;

struct KdArgs {
    const bf16_t* A; i64 lda;
    const bf16_t* A2; i64 a2_ld; int K1;            // columns k >= K1 of the A operand come from A2[m][k - K1] (K1 = K: none)
    const bf16_t* B; i64 ldb;
    bf16_t* C; i64 ldc;
    int M, N, K;
    const float* bias;                               // fp32 [N], added before rounding (K-concat epilogue) or NULL
    const float* gate; int gate_ld, rows_per_sample; // SE gate [B][K] fp32 on the A operand or NULL
    double* stats; int stat_nchan;
    int ntm, ntn;
};

static __device__ __forceinline__ void kd_lds_barrier() {           // LDS hand-off only: global loads stay in flight across it
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}

template <int BN, int RT, bool GATE, bool CAT>
__global__ __launch_bounds__(256, BN * RT >= 512 ? 2 : KD_MINW128) void gemm_kd_kernel(const KdArgs g) {
    constexpr int BM = 64 * RT;                     // RT 16-row tiles per wave, four waves
    constexpr int NJ = BN / 16;                     // 16-column accumulator tiles per wave row tile
    constexpr int NCH = BN * 4 / 256;               // 16-byte weight chunks a thread stages per k-step
    constexpr int STG = BN * 64;                    // bytes of one weight stage (BN rows x 32 k)
    constexpr int CROW = BN * 2 + 16;               // epilogue staging row stride (bytes)
    constexpr int CPR = BN / 8;                     // 16-byte chunks per output row
    constexpr int NIT = 64 * CPR / 256;             // read-back iterations per 64-row pass
    constexpr int FRG = BN == 256 ? KD_FRG256 : KD_FRG;                     // weight fragments read from LDS before their MFMAs are issued
    extern __shared__ __attribute__((aligned(16))) unsigned char kd_smem[];
    unsigned char* const sB = kd_smem;
    unsigned char* const sC = kd_smem + 2 * STG;
    float* const lred = reinterpret_cast<float*>(sC + 64 * CROW);
    float* const sgate = lred + 2 * BN;

    const int tid = threadIdx.x;
    const int wave = tid >> 6, lane = tid & 63, lr = lane & 15, lg = lane >> 4;
    // XCD-aware order: an XCD takes a contiguous run of tiles, the N-tiles of one M-tile next to each other (they re-read the same
    // A rows: the second read hits that XCD's L2)
    const int t = (int)(blockIdx.x >> 3) + (int)(blockIdx.x & 7) * (int)(gridDim.x >> 3);
    const int nt = t % g.ntn, mt = t / g.ntn;
    if (mt >= g.ntm) { DET_EXIT(); return; }
    const int m0 = mt * BM, n0 = nt * BN;
    const int nk = g.K >> 5;                        // 32-deep k-steps (a multiple of 4)

    // ---- A fragments: rows m0 + 32 wave + 16 i + lr (clamped: rows past M are computed and never stored)
    const bf16_t* pa[RT];
    [[maybe_unused]] const bf16_t* pa2[RT];
#pragma unroll
    for (int i = 0; i < RT; ++i) {
        int m = m0 + 16 * RT * wave + 16 * i + lr;
        m = m < g.M ? m : g.M - 1;
        pa[i] = g.A + (i64)m * g.lda + 8 * lg;
        if constexpr (CAT) pa2[i] = g.A2 + (i64)m * g.a2_ld + 8 * lg;
    }
    uint4 ar[4][RT];                                // A ring: four k-steps in flight
    auto load_a = [&](int ks, uint4 (&dst)[RT]) {
        ks = ks < nk ? ks : nk - 1;                 // past the end: a valid address, the data is never used
        const int k = ks << 5;
#pragma unroll
        for (int i = 0; i < RT; ++i) {
            const bf16_t* src = pa[i] + k;
            if constexpr (CAT) src = k >= g.K1 ? pa2[i] + (k - g.K1) : src;       // a pointer select, not a branch around the load
            dst[i] = *reinterpret_cast<const uint4*>(src);
        }
    };
    // ---- weight staging: chunk c = tid + 256 i -> row c >> 2, 16-byte k chunk c & 3 of the k-step
    const bf16_t* pb[NCH];
    unsigned sboff[NCH];
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
        const int c = tid + 256 * i, n = c >> 2, kc = c & 3;
        pb[i] = g.B + (i64)(n0 + n) * g.ldb + 8 * kc;
        sboff[i] = (unsigned)(n * 64 + ((kc ^ ((0 - (n >> 2)) & 3)) * 16));
    }
    uint4 bs[2][NCH];
    auto load_b = [&](int ks, uint4 (&dst)[NCH]) {
        ks = ks < nk ? ks : nk - 1;
#pragma unroll
        for (int i = 0; i < NCH; ++i) dst[i] = *reinterpret_cast<const uint4*>(pb[i] + (ks << 5));
    };
    auto store_b = [&](const int stage, const uint4 (&src)[NCH]) {
#pragma unroll
        for (int i = 0; i < NCH; ++i) *reinterpret_cast<uint4*>(sB + stage * STG + sboff[i]) = make_uint4(src[i].x, src[i].y, src[i].z, src[i].w);
    };

    kd_f32x4_t acc[RT][NJ];
#pragma unroll
    for (int i = 0; i < RT; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j) acc[i][j] = kd_f32x4_t{0.f, 0.f, 0.f, 0.f};

    // ---- prologue: four k-steps of A and two of the weights in flight, the sample's gate row in LDS, weight stage 0 written
#pragma unroll
    for (int u = 0; u < 4; ++u) load_a(u, ar[u]);
    load_b(0, bs[0]);
    load_b(1, bs[1]);
    if constexpr (GATE) {
        const float* grow = g.gate + (i64)(m0 / g.rows_per_sample) * g.gate_ld;
        for (int k = tid * 4; k < g.K; k += 1024) *reinterpret_cast<float4*>(sgate + k) = *reinterpret_cast<const float4*>(grow + k);
    }
    store_b(0, bs[0]);
    load_b(2, bs[0]);
    kd_lds_barrier();

    // this lane's fragment address inside a 16-row weight tile.  Rows are 64 bytes, so four rows share a 256-byte bank row and the
    // 16-lane groups of ds_read_b128 ({0-3, 12-15, 20-27}, ...) would meet two-way on every read; the 16-byte chunk of row n is
    // stored at chunk ^ (-(n >> 2) & 3), which gives every group sixteen distinct slots (and keeps the staging writes, 8 lanes = 2
    // whole rows, conflict-free)
    const unsigned char* const fb = sB + lr * 64 + ((lg ^ ((0 - (lr >> 2)) & 3)) * 16);
    // one k-step; u = ks & 3 is a compile-time constant so that every ring index is static (a runtime index sends the rings to scratch)
    auto step = [&](auto uc, const int ks) {
            constexpr int u = decltype(uc)::value;
            // weights of k-step ks + 1 (loaded two steps ago) into the stage step ks - 1 has released; then k-step ks + 3's loads
            {
                constexpr int sb = (u + 1) & 1;
                const int kb = (ks + 3 < nk ? ks + 3 : nk - 1) << 5;
#pragma unroll
                for (int i = 0; i < NCH; ++i) {
                    // (component-wise: a whole-struct copy from the ring becomes a memcpy from a stack object and pins the ring to scratch)
                    *reinterpret_cast<uint4*>(sB + sb * STG + sboff[i]) = make_uint4(bs[sb][i].x, bs[sb][i].y, bs[sb][i].z, bs[sb][i].w);
                    bs[sb][i] = *reinterpret_cast<const uint4*>(pb[i] + kb);
                }
            }
            uint4 af[RT];
#pragma unroll
            for (int i = 0; i < RT; ++i) af[i] = ar[u][i];
            if constexpr (GATE) {
                const float4 g0 = *reinterpret_cast<const float4*>(sgate + (ks << 5) + 8 * lg);
                const float4 g1 = *reinterpret_cast<const float4*>(sgate + (ks << 5) + 8 * lg + 4);
                const float gv[8] = {g0.x, g0.y, g0.z, g0.w, g1.x, g1.y, g1.z, g1.w};
#pragma unroll
                for (int i = 0; i < RT; ++i) {
                    float v[8];
                    unpack16<bf16_t>(af[i], v);
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] *= gv[e];
                    af[i] = pack16<bf16_t>(v);
                }
            }
            const unsigned char* const st = fb + (u & 1) * STG;
            // weight fragments FRG at a time into registers of their own, then their MFMAs (FRG = 1, 2, 4, 8 measured within 1 % of each
            // other; 2 is what fits three waves per SIMD at 128 columns, 1 at 256 columns where 128 accumulators leave no room)
#pragma unroll
            for (int j0 = 0; j0 < NJ; j0 += FRG) {
                uint4 wf[FRG];
#pragma unroll
                for (int j = 0; j < FRG; ++j) wf[j] = *reinterpret_cast<const uint4*>(st + (j0 + j) * 1024);
                if constexpr (FRG > 1) __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int j = 0; j < FRG; ++j)
#pragma unroll
                    for (int i = 0; i < RT; ++i)
                        acc[i][j0 + j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(kd_bf16x8_t, wf[j]), __builtin_bit_cast(kd_bf16x8_t, af[i]),
                                                                                 acc[i][j0 + j], 0, 0, 0);
            }
            // A rows are 128-byte lines of which a k-step uses 64 bytes: the loads of two consecutive k-steps are issued back to back
            // (after the odd step, into the two slots just consumed), so that the second half of a line is requested while the
            // first is still in flight in the L1
            if constexpr (u & 1) { load_a(ks + 3, ar[u - 1]); load_a(ks + 4, ar[u]); }
            kd_lds_barrier();
    };
    for (int ks0 = 0; ks0 < nk; ks0 += 4) {
        step(std::integral_constant<int, 0>{}, ks0);
        step(std::integral_constant<int, 1>{}, ks0 + 1);
        step(std::integral_constant<int, 2>{}, ks0 + 2);
        step(std::integral_constant<int, 3>{}, ks0 + 3);
    }

    // ---- epilogue: 64 rows per pass through LDS (waves 2p, 2p + 1 own the rows of pass p)
    const int ch = tid % CPR;
    float st0[8], st1[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) { st0[e] = 0.f; st1[e] = 0.f; }
#pragma unroll
    for (int p = 0; p < BM / 64; ++p) {
        // rows 64 p .. 64 p + 63 of the tile: waves 2p, 2p + 1 (RT = 2) or wave p (RT = 4)
        if (16 * RT * wave / 64 == p) {
#pragma unroll
            for (int i = 0; i < RT; ++i) {
                const int row = (16 * RT * wave) % 64 + 16 * i + lr;
#pragma unroll
                for (int j = 0; j < NJ; ++j) {
                    const int col = 16 * j + 4 * lg;
                    kd_f32x4_t v = acc[i][j];
                    if (g.bias) {
                        const float4 bv = *reinterpret_cast<const float4*>(g.bias + n0 + col);
                        v[0] += bv.x; v[1] += bv.y; v[2] += bv.z; v[3] += bv.w;
                    }
                    *reinterpret_cast<uint2*>(sC + row * CROW + col * 2) = make_uint2(pk_bf16(v[0], v[1]), pk_bf16(v[2], v[3]));
                }
            }
        }
        kd_lds_barrier();
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int row = tid / CPR + it * (256 / CPR);
            const int m = m0 + 64 * p + row;
            const uint4 raw = *reinterpret_cast<const uint4*>(sC + row * CROW + ch * 16);
            if (m < g.M) {
                *reinterpret_cast<uint4*>(g.C + (i64)m * g.ldc + n0 + ch * 8) = raw;
                float v[8];
                unpack16<bf16_t>(raw, v);
#pragma unroll
                for (int e = 0; e < 8; ++e) { st0[e] += v[e]; st1[e] = fmaf(v[e], v[e], st1[e]); }
            }
        }
        kd_lds_barrier();
    }
    if (g.stats) {
        for (int i = tid; i < 2 * BN; i += 256) lred[i] = 0.f;
        __syncthreads();
        DET_WAVES_BEGIN
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            atomicAdd(&lred[ch * 8 + e], st0[e]);
            atomicAdd(&lred[BN + ch * 8 + e], st1[e]);
        }
        DET_WAVES_END
        __syncthreads();
        DET_ENTER();
        for (int i = tid; i < 2 * BN; i += 256)
            stat_add(g.stats, (int)(blockIdx.x % DWN_NREP), g.stat_nchan, i / BN, n0 + i % BN, lred[i]);
    }
    DET_EXIT();
}

// ------------------------------------------------------------------------------------------------
static bool kd_args_ok(const GemmNN& g, int dtype) {
    if (dtype != DWN_BF16 || g.groups != 1 || g.b_sample_stride || g.M < 1) return false;
    if (g.epi == EPI_STORE) {
        if (g.a_kind != LD_PLAIN && g.a_kind != LD_GATE) return false;
        if (g.a2) return false;
    } else if (g.epi == EPI_STORE_CAT) {
        if (g.a_kind != LD_PLAIN || !g.a2 || !g.bias || g.K1 <= 0 || g.K1 % 32 || g.a2_ld % 8 || ((size_t)g.a2 & 15) || ((size_t)g.bias & 15)) return false;
    } else {
        return false;
    }
    if (g.a_kind == LD_GATE && (!g.a.gate || g.a.rows_per_sample <= 0 || g.a.rows_per_sample % 128 || g.a.gate_ld % 4 || ((size_t)g.a.gate & 15) || g.K > 4096))
        return false;
    if (g.K % 128 || g.N % 128 || g.a.ld % 8 || g.ldb % 8 || g.ldc % 8) return false;
    if (((size_t)g.a.p | (size_t)g.b | (size_t)g.c) & 15) return false;
    return true;
}

bool gemm_nn_kd_eligible(const GemmNN& g, int dtype) {
    if (g.variant == DWN_NN_TILE128 || g.variant == DWN_NN_XL128 || g.variant == DWN_NN_XL256) return false;
    if (!kd_args_ok(g, dtype)) return false;
    if (g.variant == DWN_NN_KD) return true;
    // by shape, measured (tools/kd_time.py, us, 128-row kernels -> this one): conv_pwl forward 171 -> 147 (147456 x 256 x 896),
    // 118 -> 82 (40960 x 256 x 1792), 115 -> 109 (147456 x 128 x 896); conv_pw data gradient 253 -> 237 (147456 x 256 x 2048),
    // 90 -> 75 (40960 rows) — but 320 -> 361 and 91 -> 101 at N = 128: there the weights (re-read from L2 once per 128 rows)
    // are as many bytes as the A rows and the kernel with both operands in LDS moves them at the same ~26 GB/s per CU
    return g.K >= 512 && g.M >= 8192 && (g.N == 256 || (g.N == 128 && g.a_kind == LD_GATE));
}

template <int BN, int RT, bool GATE, bool CAT>
static int launch_kd_t(const KdArgs& a, hipStream_t s) {
    const size_t smem = 2 * (size_t)BN * 64 + 64 * ((size_t)BN * 2 + 16) + 2 * (size_t)BN * 4 + (GATE ? (size_t)a.K * 4 : 0);
    // per launch: the attribute belongs to the (function, device) pair and the API takes a device argument (a process-wide flag left
    // a second device without it); the call is a host-side table write
    if (smem > 48 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_kd_kernel<BN, RT, GATE, CAT>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
        if (e != hipSuccess) return dwn_set_error((int)e, hipGetErrorString(e));
    }
    const unsigned tiles = (unsigned)a.ntm * (unsigned)a.ntn;
    const unsigned grid = (tiles + 7u) & ~7u;
    hipLaunchKernelGGL((gemm_kd_kernel<BN, RT, GATE, CAT>), dim3(grid), dim3(256), smem, s, a);
    DWN_CHECK_LAUNCH();
    return 0;
}

int launch_gemm_nn_kd(const GemmNN& g, hipStream_t s) {
    KdArgs a;
    a.A = reinterpret_cast<const bf16_t*>(g.a.p); a.lda = g.a.ld;
    const bool cat = g.epi == EPI_STORE_CAT;
    a.A2 = cat ? reinterpret_cast<const bf16_t*>(g.a2) : nullptr; a.a2_ld = cat ? g.a2_ld : 0; a.K1 = cat ? g.K1 : g.K;
    a.B = reinterpret_cast<const bf16_t*>(g.b); a.ldb = g.ldb;
    a.C = reinterpret_cast<bf16_t*>(g.c); a.ldc = g.ldc;
    a.M = g.M; a.N = g.N; a.K = g.K;
    a.bias = cat ? g.bias : nullptr;
    const bool gate = g.a_kind == LD_GATE;
    a.gate = gate ? g.a.gate : nullptr; a.gate_ld = gate ? g.a.gate_ld : 0; a.rows_per_sample = gate ? g.a.rows_per_sample : 1;
    a.stats = g.stats; a.stat_nchan = g.stat_nchan;
    // 256-column tiles read A once; they need two resident workgroups per CU's worth of tiles to fill the chip.  (256-ROW tiles at
    // 128 columns — RT = 4, the weights re-read from L2 half as often — measured the same time as 128-row tiles, 360 us at
    // 589824 x 128 x 1024, against 310 us for the kernel with both operands in LDS: not built.)
    const int ntm128 = (g.M + 127) / 128;
    const bool wide = KD_WIDE && g.N % 256 == 0 && (i64)ntm128 * (g.N / 256) >= 512;
    a.ntm = ntm128;
    a.ntn = wide ? g.N / 256 : g.N / 128;
    if (wide) {
        if (gate) return launch_kd_t<256, 2, true, false>(a, s);
        return cat ? launch_kd_t<256, 2, false, true>(a, s) : launch_kd_t<256, 2, false, false>(a, s);
    }
    if (gate) return launch_kd_t<128, 2, true, false>(a, s);
    return cat ? launch_kd_t<128, 2, false, true>(a, s) : launch_kd_t<128, 2, false, false>(a, s);
}
