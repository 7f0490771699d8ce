// Prototype: conv_pw backward of a 64-channel block (E = 448, Cin = 64) reading dh1 / y1 ONCE:
//   dy1 = A1*dh1 + A2*y1 + A3   (BatchNorm-backward affine, per channel of E)        built per 64-column chunk in LDS
//   da0[M][64]  = dy1 . W1          (data gradient:   A fragments read row-major)
//   dW1[448][64] += dy1^T . a0      (weight gradient: fragments read transposed, ds_read_tr16_b64)
// against the two kernels of the library (gemm_nn K-concat + gemm_tn affine2) that each stream dh1.
// 512 threads, one workgroup per CU, persistent over 128-row tiles; W1^T resident in LDS.
// Build: hipcc -O3 --offload-arch=gfx950 -o pw_bwd_fused pw_bwd_fused.hip ; run: ./pw_bwd_fused [M]
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <stdint.h>
#include <string.h>
#include <math.h>
#include <vector>
#include <type_traits>
typedef uint16_t bf16_t;
typedef __attribute__((ext_vector_type(8))) short bf16x8_t;
typedef __attribute__((ext_vector_type(4))) short s16x4_t;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;
static __device__ __forceinline__ unsigned pack2(float a, float b) {
    __bf16 x = (__bf16)a, y = (__bf16)b;
    return (unsigned)__builtin_bit_cast(bf16_t, x) | ((unsigned)__builtin_bit_cast(bf16_t, y) << 16);
}
static inline float bf2f_h(bf16_t v) { uint32_t u = ((uint32_t)v) << 16; float f; memcpy(&f, &u, 4); return f; }
static inline bf16_t f2bf_h(float f) { uint32_t u; memcpy(&u, &f, 4); u += 0x7fff + ((u >> 16) & 1); return (bf16_t)(u >> 16); }

constexpr int E = 448, CIN = 64, BM = 128, NKC = E / 64;
constexpr int RS = 160;                       // LDS row stride of the [128][64] bf16 tiles (128 B + 32 B shift)
constexpr int WRS = E * 2 + 16;               // row stride of the resident W1^T [64][448]
constexpr int SW_BYTES = CIN * WRS, SD_BYTES = BM * RS, SX_BYTES = BM * RS;
constexpr int SABC_BYTES = 3 * E * 4;          // the affine coefficients live in LDS: read per chunk, not hoisted into 168 registers
constexpr int LDS_BYTES = SW_BYTES + 2 * SD_BYTES + 2 * SX_BYTES + SABC_BYTES;
// RC variant (y1 recomputed from the a0 tile instead of read): one a0 buffer + one y1-chunk tile
constexpr int LDS_BYTES_RC = SW_BYTES + 2 * SD_BYTES + SX_BYTES + SD_BYTES + SABC_BYTES;

template <bool RC>
__global__ __launch_bounds__(512, 2) void pw_bwd_fused(const bf16_t* __restrict__ dh1, const bf16_t* __restrict__ y1,
                                                       const bf16_t* __restrict__ a0, const bf16_t* __restrict__ w1t,
                                                       const float* __restrict__ abc, bf16_t* __restrict__ da0,
                                                       float* __restrict__ dW, int M) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* sW = smem;
    unsigned char* sD = smem + SW_BYTES;
    unsigned char* sX = sD + 2 * SD_BYTES;
    unsigned char* sY = sX + SX_BYTES;                        // RC only (takes the place of the second a0 buffer)
    float* sABC = reinterpret_cast<float*>(sX + 2 * SX_BYTES);
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int lr = lane & 15, lg = lane >> 4;
    const int wm = wave & 3, wn = wave >> 2;
    for (int c = tid; c < 3 * E; c += 512) sABC[c] = abc[c];
    // resident W1^T: [n][k], 16-byte chunks
    for (int c = tid; c < CIN * (E / 8); c += 512) {
        const int n = c / (E / 8), kc8 = c % (E / 8);
        *reinterpret_cast<uint4*>(sW + n * WRS + kc8 * 16) = *reinterpret_cast<const uint4*>(w1t + (size_t)n * E + kc8 * 8);
    }
    __syncthreads();
    f32x4_t acc_dw[NKC][2];
#pragma unroll
    for (int k = 0; k < NKC; ++k) { acc_dw[k][0] = f32x4_t{0, 0, 0, 0}; acc_dw[k][1] = f32x4_t{0, 0, 0, 0}; }
    const int ntiles = M / BM;
    const int ch = tid & 7;                   // this thread's 16-byte column chunk inside a 64-column chunk (fixed)
    const int row_a = tid >> 3;               // rows row_a and row_a + 64
    uint4 rd[2], ry[2], rx[2];
    // RC: one register slot per chunk index, loaded PD chunk-steps ahead of its use (also across the tile boundary): with a
    // single chunk in flight the kernel is latency-bound (one HBM round trip per chunk), whatever the byte count
    constexpr int PD = 3;
    [[maybe_unused]] uint4 rq[RC ? NKC : 1][2];
    auto issue_chunk = [&](int t, auto jc) {
        constexpr int j = decltype(jc)::value;
#pragma unroll
        for (int u = 0; u < 2; ++u)
            rq[RC ? j : 0][u] = *reinterpret_cast<const uint4*>(dh1 + ((size_t)t * BM + row_a + 64 * u) * E + j * 64 + ch * 8);
    };
    if constexpr (RC) {
        if ((int)blockIdx.x < ntiles) {
            issue_chunk(blockIdx.x, std::integral_constant<int, 0>{});
            issue_chunk(blockIdx.x, std::integral_constant<int, 1>{});
            issue_chunk(blockIdx.x, std::integral_constant<int, 2>{});
#pragma unroll
            for (int u = 0; u < 2; ++u) rx[u] = *reinterpret_cast<const uint4*>(a0 + ((size_t)blockIdx.x * BM + row_a + 64 * u) * CIN + ch * 8);
        }
    }
    int tpar = 0, step = 0;                  // step: running chunk counter (LDS buffer parity continues across tiles)
    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x, tpar ^= 1) {
        const size_t m0 = (size_t)tile * BM;
        unsigned char* sXt = sX + (RC ? 0 : tpar) * SX_BYTES;
        // a0 tile + first chunk of dh1 / y1 (RC: already in flight / in registers)
        if constexpr (!RC) {
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const size_t m = m0 + row_a + 64 * u;
            rx[u] = *reinterpret_cast<const uint4*>(a0 + m * CIN + ch * 8);
            rd[u] = *reinterpret_cast<const uint4*>(dh1 + m * E + ch * 8);
            ry[u] = *reinterpret_cast<const uint4*>(y1 + m * E + ch * 8);
        }
        }
#pragma unroll
        for (int u = 0; u < 2; ++u) *reinterpret_cast<uint4*>(sXt + (row_a + 64 * u) * RS + ch * 16) = rx[u];
        f32x4_t acc_da[2][2];
#pragma unroll
        for (int i = 0; i < 2; ++i) { acc_da[i][0] = f32x4_t{0, 0, 0, 0}; acc_da[i][1] = f32x4_t{0, 0, 0, 0}; }
        auto chunk_step = [&](auto kc_c) {
            constexpr int kc = decltype(kc_c)::value;
            unsigned char* sDk = sD + ((step + kc) & 1) * SD_BYTES;
            if constexpr (RC) {
                // y1 chunk = round_bf16(a0_tile . W1[chunk]^T), the forward's values: a0 fragments row-major from sX, W1
                // fragments (k = c) transposed out of the resident W1^T [c][e]; lanes own 4 consecutive e of a row
                if (kc == 0) __syncthreads();                 // the a0 tile (and, first tile, W1^T) is in LDS
                f32x4_t acc_y[2][2];
#pragma unroll
                for (int i = 0; i < 2; ++i) { acc_y[i][0] = f32x4_t{0, 0, 0, 0}; acc_y[i][1] = f32x4_t{0, 0, 0, 0}; }
                const int q = lr >> 2, p = lr & 3;
#pragma unroll
                for (int kb = 0; kb < 2; ++kb) {
                    bf16x8_t xf[2], wf[2];
#pragma unroll
                    for (int i = 0; i < 2; ++i)
                        xf[i] = *reinterpret_cast<const bf16x8_t*>(sXt + (wm * 32 + i * 16 + lr) * RS + (kb * 4 + lg) * 16);
                    const int rb = kb * 32 + 8 * lg + q;
#pragma unroll
                    for (int j = 0; j < 2; ++j) {
                        const int colb = (kc * 64 + wn * 32 + j * 16 + 4 * p) * 2;
                        auto lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_t*)(sW + rb * WRS + colb));
                        auto hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_t*)(sW + (rb + 4) * WRS + colb));
                        wf[j] = bf16x8_t{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
                    }
#pragma unroll
                    for (int i = 0; i < 2; ++i)
#pragma unroll
                        for (int j = 0; j < 2; ++j)
                            acc_y[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[j], xf[i], acc_y[i][j], 0, 0, 0);
                }
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j)
                        *reinterpret_cast<uint2*>(sY + (wm * 32 + i * 16 + lr) * RS + (wn * 32 + j * 16 + 4 * lg) * 2) =
                            make_uint2(pack2(acc_y[i][j][0], acc_y[i][j][1]), pack2(acc_y[i][j][2], acc_y[i][j][3]));
                __syncthreads();
#pragma unroll
                for (int u = 0; u < 2; ++u) { ry[u] = *reinterpret_cast<const uint4*>(sY + (row_a + 64 * u) * RS + ch * 16); rd[u] = rq[kc][u]; }
            }
            // BatchNorm-backward affine of this chunk -> LDS (bf16, as the library's loader rounds it)
            {
                float A1[8], A2[8], A3[8];
                const int e0 = kc * 64 + ch * 8;
#pragma unroll
                for (int q = 0; q < 8; ++q) { A1[q] = sABC[e0 + q]; A2[q] = sABC[E + e0 + q]; A3[q] = sABC[2 * E + e0 + q]; }
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    const unsigned d4[4] = {rd[u].x, rd[u].y, rd[u].z, rd[u].w}, y4[4] = {ry[u].x, ry[u].y, ry[u].z, ry[u].w};
                    unsigned o[4];
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const float dl = __uint_as_float(d4[q] << 16), dhi = __uint_as_float(d4[q] & 0xffff0000u);
                        const float yl = __uint_as_float(y4[q] << 16), yh = __uint_as_float(y4[q] & 0xffff0000u);
                        o[q] = pack2(fmaf(A1[2 * q], dl, fmaf(A2[2 * q], yl, A3[2 * q])),
                                     fmaf(A1[2 * q + 1], dhi, fmaf(A2[2 * q + 1], yh, A3[2 * q + 1])));
                    }
                    *reinterpret_cast<uint4*>(sDk + (row_a + 64 * u) * RS + ch * 16) = make_uint4(o[0], o[1], o[2], o[3]);
                }
            }
            __syncthreads();
            if constexpr (RC) {
                constexpr int jn = (kc + PD) % NKC;
                const int tn = (kc + PD < NKC) ? tile : tile + (int)gridDim.x;
                if (tn < ntiles) issue_chunk(tn, std::integral_constant<int, jn>{});
                if (kc == 3 && tile + (int)gridDim.x < ntiles) {
#pragma unroll
                    for (int u = 0; u < 2; ++u)
                        rx[u] = *reinterpret_cast<const uint4*>(a0 + ((size_t)(tile + gridDim.x) * BM + row_a + 64 * u) * CIN + ch * 8);
                }
            } else if (kc + 1 < NKC) {               // next chunk in flight under the MFMAs
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    const size_t m = m0 + row_a + 64 * u;
                    rd[u] = *reinterpret_cast<const uint4*>(dh1 + m * E + (kc + 1) * 64 + ch * 8);
                    ry[u] = *reinterpret_cast<const uint4*>(y1 + m * E + (kc + 1) * 64 + ch * 8);
                }
            }
            // ---- data gradient: acc_da[m][n] += sum_k dy1[m][k] W1t[n][k]   (swapped roles: lanes own 4 consecutive n)
#pragma unroll
            for (int kb = 0; kb < 2; ++kb) {
                bf16x8_t af[2], wf[2];
#pragma unroll
                for (int i = 0; i < 2; ++i)
                    af[i] = *reinterpret_cast<const bf16x8_t*>(sDk + (wm * 32 + i * 16 + lr) * RS + (kb * 4 + lg) * 16);
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    wf[j] = *reinterpret_cast<const bf16x8_t*>(sW + (wn * 32 + j * 16 + lr) * WRS + (kc * 64 + kb * 32 + lg * 8) * 2);
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j)
                        acc_da[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[j], af[i], acc_da[i][j], 0, 0, 0);
            }
            // ---- weight gradient: acc_dw[kc][e][c] += sum_rows dy1[row][e] a0[row][c]   (transposed fragments)
            {
                const int q = lr >> 2, p = lr & 3;
#pragma unroll
                for (int kb = 0; kb < BM / 32; ++kb) {
                    const int rb = kb * 32 + 8 * lg + q;
                    bf16x8_t ef, cf[2];
                    {
                        const int colb = (wm * 16 + 4 * p) * 2;
                        auto lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_t*)(sDk + rb * RS + colb));
                        auto hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_t*)(sDk + (rb + 4) * RS + colb));
                        ef = bf16x8_t{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
                    }
#pragma unroll
                    for (int j = 0; j < 2; ++j) {
                        const int colb = (wn * 32 + j * 16 + 4 * p) * 2;
                        auto lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_t*)(sXt + rb * RS + colb));
                        auto hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_t*)(sXt + (rb + 4) * RS + colb));
                        cf[j] = bf16x8_t{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
                    }
#pragma unroll
                    for (int j = 0; j < 2; ++j)
                        acc_dw[kc][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ef, cf[j], acc_dw[kc][j], 0, 0, 0);
                }
            }
        };
        chunk_step(std::integral_constant<int, 0>{}); chunk_step(std::integral_constant<int, 1>{});
        chunk_step(std::integral_constant<int, 2>{}); chunk_step(std::integral_constant<int, 3>{});
        chunk_step(std::integral_constant<int, 4>{}); chunk_step(std::integral_constant<int, 5>{});
        chunk_step(std::integral_constant<int, 6>{});
        static_assert(NKC == 7, "chunk_step calls are written out");
        step += NKC;
        if constexpr (RC) __syncthreads();                    // single a0 buffer: every wave is done with this tile's sX
        // da0 tile: acc_da[i][j][r] = da0[m = wm*32 + i*16 + lr][n = wn*32 + j*16 + 4*lg + r]
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const size_t m = m0 + wm * 32 + i * 16 + lr;
                const int n = wn * 32 + j * 16 + 4 * lg;
                uint2 v = make_uint2(pack2(acc_da[i][j][0], acc_da[i][j][1]), pack2(acc_da[i][j][2], acc_da[i][j][3]));
                *reinterpret_cast<uint2*>(da0 + m * CIN + n) = v;
            }
    }
    // acc_dw[kc][j][r] = dW[e = kc*64 + wm*16 + 4*lg + r][c = wn*32 + j*16 + lr]
#pragma unroll
    for (int kc = 0; kc < NKC; ++kc)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r)
                atomicAdd(dW + (size_t)(kc * 64 + wm * 16 + 4 * lg + r) * CIN + wn * 32 + j * 16 + lr, acc_dw[kc][j][r]);
}

int main(int argc, char** argv) {
    const int M = argc > 1 ? atoi(argv[1]) : 589824;
    const size_t ne = (size_t)M * E, nc = (size_t)M * CIN;
    std::vector<bf16_t> hd(ne), hy(ne), hx(nc), hw((size_t)CIN * E), hda(nc);
    std::vector<float> habc(3 * E), hdw((size_t)E * CIN);
    srand(1);
    for (size_t i = 0; i < ne; ++i) { hd[i] = f2bf_h((float)((rand() % 5) - 2)); hy[i] = f2bf_h((float)((rand() % 3) - 1)); }
    for (size_t i = 0; i < nc; ++i) hx[i] = f2bf_h((float)((rand() % 3) - 1));
    for (size_t i = 0; i < hw.size(); ++i) hw[i] = f2bf_h((float)((rand() % 5) - 2) * 0.25f);
    const bool rc = argc > 2 && atoi(argv[2]) != 0;
    if (rc) {       // the recompute variant needs y1 to BE the forward's output: y1 = round_bf16(a0 . W1^T)  (hw is W1^T [c][e])
        for (size_t m = 0; m < (size_t)M; ++m)
            for (int e = 0; e < E; ++e) {
                float acc = 0;
                for (int c = 0; c < CIN; ++c) acc += bf2f_h(hx[m * CIN + c]) * bf2f_h(hw[(size_t)c * E + e]);
                hy[m * E + e] = f2bf_h(acc);
            }
    }
    for (int e = 0; e < E; ++e) { habc[e] = (float)(1 + e % 2); habc[E + e] = (float)((e % 3) - 1); habc[2 * E + e] = (float)((e % 2)); }
    bf16_t *D, *Y, *X, *W, *DA; float *ABC, *DW;
    hipMalloc(&D, ne * 2); hipMalloc(&Y, ne * 2); hipMalloc(&X, nc * 2); hipMalloc(&W, hw.size() * 2); hipMalloc(&DA, nc * 2);
    hipMalloc(&ABC, habc.size() * 4); hipMalloc(&DW, hdw.size() * 4);
    hipMemcpy(D, hd.data(), ne * 2, hipMemcpyHostToDevice); hipMemcpy(Y, hy.data(), ne * 2, hipMemcpyHostToDevice);
    hipMemcpy(X, hx.data(), nc * 2, hipMemcpyHostToDevice); hipMemcpy(W, hw.data(), hw.size() * 2, hipMemcpyHostToDevice);
    hipMemcpy(ABC, habc.data(), habc.size() * 4, hipMemcpyHostToDevice);
    hipMemset(DW, 0, hdw.size() * 4);
    hipFuncSetAttribute((const void*)pw_bwd_fused<false>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
    hipFuncSetAttribute((const void*)pw_bwd_fused<true>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES_RC);
    int grid = 256; if (grid > M / BM) grid = M / BM;
    if (rc) hipLaunchKernelGGL(pw_bwd_fused<true>, dim3(grid), dim3(512), LDS_BYTES_RC, 0, D, Y, X, W, ABC, DA, DW, M);
    else hipLaunchKernelGGL(pw_bwd_fused<false>, dim3(grid), dim3(512), LDS_BYTES, 0, D, Y, X, W, ABC, DA, DW, M);
    hipError_t e = hipDeviceSynchronize();
    if (e != hipSuccess) { printf("error %s\n", hipGetErrorString(e)); return 1; }
    hipMemcpy(hda.data(), DA, nc * 2, hipMemcpyDeviceToHost); hipMemcpy(hdw.data(), DW, hdw.size() * 4, hipMemcpyDeviceToHost);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    for (int r = 0; r < 10; ++r) { if (rc) hipLaunchKernelGGL(pw_bwd_fused<true>, dim3(grid), dim3(512), LDS_BYTES_RC, 0, D, Y, X, W, ABC, DA, DW, M);
    else hipLaunchKernelGGL(pw_bwd_fused<false>, dim3(grid), dim3(512), LDS_BYTES, 0, D, Y, X, W, ABC, DA, DW, M); }
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 10;
    // checks (all values are small integers / quarters: exact in fp32 up to the bf16 rounding of dy1 and of the output)
    auto dy = [&](size_t m, int k) {
        return bf2f_h(f2bf_h(fmaf(habc[k], bf2f_h(hd[m * E + k]), fmaf(habc[E + k], bf2f_h(hy[m * E + k]), habc[2 * E + k]))));
    };
    int bad = 0;
    for (int t = 0; t < 2000; ++t) {
        size_t m = (size_t)rand() % M; int n = rand() % CIN;
        float ref = 0; for (int k = 0; k < E; ++k) ref += dy(m, k) * bf2f_h(hw[(size_t)n * E + k]);
        float got = bf2f_h(hda[m * CIN + n]);
        if (fabsf(got - bf2f_h(f2bf_h(ref))) > 1e-6f * fabsf(ref) + 1e-6f) { if (bad < 5) printf("da0 mismatch m=%zu n=%d got %f ref %f\n", m, n, got, ref); ++bad; }
    }
    if (M <= 65536) {
        for (int t = 0; t < 200; ++t) {
            int ee = rand() % E, c = rand() % CIN;
            double ref = 0; for (size_t m = 0; m < (size_t)M; ++m) ref += (double)dy(m, ee) * bf2f_h(hx[m * CIN + c]);
            if (fabs(hdw[(size_t)ee * CIN + c] - ref) > 1e-3 * fabs(ref) + 1e-2) { if (bad < 10) printf("dW mismatch e=%d c=%d got %f ref %f\n", ee, c, hdw[(size_t)ee * CIN + c], ref); ++bad; }
        }
    }
    double bytes = (rc ? 1.0 : 2.0) * ne * 2 + 2.0 * nc * 2;
    printf("pw_bwd_fused%s M=%d grid=%d lds=%d: %.1f us  %.1f GB/s  mismatches %d\n", rc ? " (y1 recomputed)" : "", M, grid, LDS_BYTES, ms * 1e3, bytes / (ms * 1e-3) / 1e9, bad);
    return bad != 0;
}
