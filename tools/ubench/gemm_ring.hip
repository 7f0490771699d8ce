// Prototype: bf16 C[M][N] = A[M][K] . B[N][K]^T with an LDS-DMA (global_load_lds_dwordx4) ring of S stages, raw
// s_barrier and counted s_waitcnt vmcnt — measures what a glds pipeline buys over the register-staged k-loop GEMM.
// Build: hipcc -O3 --offload-arch=gfx950 -o gemm_ring gemm_ring.hip ; run: ./gemm_ring M N K
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <stdint.h>
#include <string.h>
#include <math.h>
#include <vector>
typedef uint16_t bf16_t;
typedef __attribute__((ext_vector_type(8))) short bf16x8_t;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;
static __device__ __forceinline__ bf16_t f2bf(float f) { __bf16 b = (__bf16)f; return __builtin_bit_cast(bf16_t, b); }
static inline float bf2f_h(bf16_t v) { uint32_t u = ((uint32_t)v) << 16; float f; memcpy(&f, &u, 4); return f; }
static inline bf16_t f2bf_h(float f) { uint32_t u; memcpy(&u, &f, 4); u += 0x7fff + ((u >> 16) & 1); return (bf16_t)(u >> 16); }

template <int N> static __device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
static __device__ __forceinline__ void wait_lgkm0() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }

constexpr int BM = 128, BN = 128, BK = 64, S = 3, STAGE = (BM + BN) * 128, CROW = BN * 2 + 16;

__global__ __launch_bounds__(256, 1) void gemm_ring(const bf16_t* A, const bf16_t* B, bf16_t* C, int M, int N, int K, int nranges) {
    extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
    unsigned char* sC = smem + S * STAGE;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int wm = wave >> 1, wn = wave & 1, lr = lane & 15, lg = lane >> 4;
    const int ntn = N / BN, ntm = M / BM, nk = K / BK;
    const int bid = blockIdx.x;
    int nt, mr;
    if ((nranges & 7) == 0) { const int jj = bid >> 3; nt = jj % ntn; mr = (jj / ntn) * 8 + (bid & 7); }
    else { const int lid = (bid & 7) * (int)(gridDim.x >> 3) + (bid >> 3); nt = lid % ntn; mr = lid / ntn; }
    const int mt_beg = (int)((long long)mr * ntm / nranges), mt_end = (int)((long long)(mr + 1) * ntm / nranges);
    if (mt_beg >= mt_end) return;
    const int n0 = nt * BN;
    const int total = (mt_end - mt_beg) * nk;
    // issue the 8 wave-instructions (4 for A, 4 for B) of flat step s into stage s % S
    auto glds16 = [&](const void* gsrc, unsigned lds_dst) {
        unsigned keep;
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
    };
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)smem;
    auto issue = [&](int s) {
        const int mt = mt_beg + s / nk, k0 = (s % nk) * BK;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int blk8 = wave * 4 + j;                       // 8-row group of the A tile
            const int row = blk8 * 8 + (lane >> 3), slot = lane & 7;
            const bf16_t* src = A + (size_t)(mt * BM + row) * K + k0 + ((slot ^ (row & 7)) << 3);
            glds16(src, __builtin_amdgcn_readfirstlane(lds0 + (unsigned)((s % S) * STAGE + blk8 * 1024)));
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int blk8 = wave * 4 + j;
            const int row = blk8 * 8 + (lane >> 3), slot = lane & 7;
            const bf16_t* src = B + (size_t)(n0 + row) * K + k0 + ((slot ^ (row & 7)) << 3);
            glds16(src, __builtin_amdgcn_readfirstlane(lds0 + (unsigned)((s % S) * STAGE + BM * 128 + blk8 * 1024)));
        }
    };
    f32x4_t acc[4][4];
    auto zero_acc = [&]() {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    };
    zero_acc();
    issue(0);
    if (total > 1) issue(1);
    bool after_epi = false;
    for (int s = 0; s < total; ++s) {
        const int rem = total - 1 - s;                           // groups issued after group s that may be outstanding
        if (rem >= 1) { if (after_epi) wait_vm<16>(); else wait_vm<8>(); }
        else { if (after_epi) wait_vm<8>(); else wait_vm<0>(); }
        __builtin_amdgcn_s_barrier();
        if (s + 2 < total) issue(s + 2);
        after_epi = false;
        const unsigned char* sA = smem + (s % S) * STAGE;
        const unsigned char* sB = sA + BM * 128;
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
            uint4 af[4], bfr[4];
            const int chunk = kb * 4 + lg;
#pragma unroll
            for (int i = 0; i < 4; ++i) { int row = wm * 64 + i * 16 + lr; af[i] = *reinterpret_cast<const uint4*>(sA + row * 128 + ((chunk ^ (row & 7)) << 4)); }
#pragma unroll
            for (int j = 0; j < 4; ++j) { int row = wn * 64 + j * 16 + lr; bfr[j] = *reinterpret_cast<const uint4*>(sB + row * 128 + ((chunk ^ (row & 7)) << 4)); }
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, bfr[j]), __builtin_bit_cast(bf16x8_t, af[i]), acc[i][j], 0, 0, 0);
        }
        if (s % nk == nk - 1) {
            const int m0 = (mt_beg + s / nk) * BM;
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int trow = wm * 64 + i * 16 + lr, col = wn * 64 + j * 16 + lg * 4;
                    uint2 v;
                    v.x = (uint32_t)f2bf(acc[i][j][0]) | ((uint32_t)f2bf(acc[i][j][1]) << 16);
                    v.y = (uint32_t)f2bf(acc[i][j][2]) | ((uint32_t)f2bf(acc[i][j][3]) << 16);
                    *reinterpret_cast<uint2*>(sC + trow * CROW + col * 2) = v;
                }
            wait_lgkm0();
            __builtin_amdgcn_s_barrier();
            const int ch = tid & 15;
#pragma unroll
            for (int it = 0; it < 8; ++it) {
                const int row = (tid >> 4) + it * 16;
                const uint4 raw = *reinterpret_cast<const uint4*>(sC + row * CROW + ch * 16);
                *reinterpret_cast<uint4*>(C + (size_t)(m0 + row) * N + n0 + ch * 8) = raw;
            }
            zero_acc();
            after_epi = true;
        }
    }
}

int main(int argc, char** argv) {
    int M = argc > 1 ? atoi(argv[1]) : 589824, N = argc > 2 ? atoi(argv[2]) : 128, K = argc > 3 ? atoi(argv[3]) : 896;
    size_t na = (size_t)M * K, nb = (size_t)N * K, nc = (size_t)M * N;
    std::vector<bf16_t> ha(na), hb(nb), hc(nc);
    srand(1);
    for (size_t i = 0; i < na; ++i) ha[i] = f2bf_h((float)((rand() % 7) - 3));
    for (size_t i = 0; i < nb; ++i) hb[i] = f2bf_h((float)((rand() % 5) - 2) * 0.25f);
    bf16_t *A, *B, *C;
    hipMalloc(&A, na * 2); hipMalloc(&B, nb * 2); hipMalloc(&C, nc * 2);
    hipMemcpy(A, ha.data(), na * 2, hipMemcpyHostToDevice); hipMemcpy(B, hb.data(), nb * 2, hipMemcpyHostToDevice);
    const int lds = S * STAGE + BM * CROW;
    hipFuncSetAttribute((const void*)gemm_ring, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    const int ntn = N / BN, ntm = M / BM;
    int nranges = 256 / ntn; if (nranges > ntm) nranges = ntm; if (nranges < 1) nranges = 1;
    while (nranges > 1 && (nranges * ntn) % 8 != 0) --nranges;
    if ((nranges * ntn) % 8 != 0) nranges = 8;
    dim3 grid(nranges * ntn);
    hipLaunchKernelGGL(gemm_ring, grid, dim3(256), lds, 0, A, B, C, M, N, K, nranges);
    hipError_t e = hipDeviceSynchronize();
    if (e != hipSuccess) { printf("error %s\n", hipGetErrorString(e)); return 1; }
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    for (int r = 0; r < 10; ++r) hipLaunchKernelGGL(gemm_ring, grid, dim3(256), lds, 0, A, B, C, M, N, K, nranges);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 10;
    hipMemcpy(hc.data(), C, nc * 2, hipMemcpyDeviceToHost);
    // spot check 2000 random entries (exact: small integers / quarter steps)
    int bad = 0;
    for (int t = 0; t < 2000; ++t) {
        size_t m = (size_t)rand() % M, n = (size_t)rand() % N;
        float ref = 0; for (int k = 0; k < K; ++k) ref += bf2f_h(ha[m * K + k]) * bf2f_h(hb[n * K + k]);
        float got = bf2f_h(hc[m * N + n]);
        if (fabsf(got - bf2f_h(f2bf_h(ref))) > 1e-6f * fabsf(ref) + 1e-6f) { if (bad < 5) printf("mismatch m=%zu n=%zu got %f ref %f\n", m, n, got, ref); ++bad; }
    }
    double bytes = (double)na * 2 + (double)nc * 2 + (double)nb * 2;
    printf("gemm_ring M=%d N=%d K=%d grid=%d lds=%d: %.1f us  %.1f GB/s  mismatches %d\n", M, N, K, grid.x, lds, ms * 1e3, bytes / (ms * 1e-3) / 1e9, bad);
    return bad != 0;
}
