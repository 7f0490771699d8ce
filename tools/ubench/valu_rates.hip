// VALU issue-rate microbenchmark for gfx950: cycles per wave-instruction of the ops the stencil kernels are made of.
// Build: hipcc -O3 --offload-arch=gfx950 -o valu_rates valu_rates.hip ; run: ./valu_rates
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f2_t __attribute__((ext_vector_type(2)));
typedef __attribute__((ext_vector_type(2))) __bf16 bf2;
#define ITERS 4096
#define REP 16
template <int OP>
__global__ __launch_bounds__(256) void k(float* out, long long* cyc, float seed) {
    float a[REP]; f2_t p[REP]; unsigned u[REP];
    for (int i = 0; i < REP; ++i) { a[i] = seed + i + threadIdx.x * 1e-3f; p[i] = f2_t{a[i], a[i] * 0.5f}; u[i] = 0x3f803f80u + i; }
    const float b = seed * 0.999f, c = 1e-6f;
    const f2_t pb = f2_t{b, b}, pc = f2_t{c, c};
    long long t0 = clock64();
    for (int it = 0; it < ITERS; ++it) {
#pragma unroll
        for (int i = 0; i < REP; ++i) {
            if (OP == 0) a[i] = fmaf(a[i], b, c);
            else if (OP == 1) p[i] = p[i] * pb + pc;
            else if (OP == 2) a[i] = __builtin_amdgcn_exp2f(a[i]);
            else if (OP == 3) a[i] = __builtin_amdgcn_rcpf(a[i]);
            else if (OP == 4) a[i] = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf2, u[i]), __builtin_bit_cast(bf2, u[(i + 1) % REP]), a[i], false);
            else if (OP == 5) u[i] = (u[i] << 16) ^ u[(i + 3) % REP];
            else if (OP == 6) { __bf16 h = (__bf16)a[i]; a[i] = (float)h + c; }
            else if (OP == 7) u[i] = __builtin_amdgcn_alignbit(u[i], u[(i + 1) % REP], 16);
            else if (OP == 8) a[i] = a[i] * b;
            else if (OP == 9) p[i] = p[i] + pb;
        }
    }
    long long t1 = clock64();
    float s = 0; for (int i = 0; i < REP; ++i) s += a[i] + p[i].x + p[i].y + (float)u[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
template <int OP> void run(const char* name, int waves_per_simd) {
    float* out; long long* cyc;
    int blocks = 256 * waves_per_simd;        // 256 threads = 4 waves = one per SIMD; waves_per_simd blocks per CU
    hipMalloc(&out, blocks * 256 * sizeof(float)); hipMalloc(&cyc, blocks * sizeof(long long));
    hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, out, cyc, 1.0001f);
    hipDeviceSynchronize();
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, out, cyc, 1.0001f);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    long long h[4]; hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
    double per = (double)h[0] / ((double)ITERS * REP);
    // wave-instructions per second per SIMD = waves_per_simd * ITERS*REP / time
    double rate = (double)waves_per_simd * ITERS * REP / (ms * 1e-3);
    printf("%-22s waves/SIMD=%d  clock64 ticks per instr (one wave) %.2f   wave-instr/s/SIMD %.3e  (time %.3f ms)\n", name, waves_per_simd, per, rate, ms);
    hipFree(out); hipFree(cyc);
}
int main() {
    for (int w : {1, 4}) {
        if (w == 1) {
            run<0>("v_fma_f32", 1); run<1>("v_pk_fma_f32", 1); run<8>("v_mul_f32", 1); run<9>("v_pk_add_f32", 1); run<2>("v_exp_f32", 1); run<3>("v_rcp_f32", 1);
            run<4>("v_dot2c_f32_bf16", 1); run<5>("lshl+xor (2 int ops)", 1); run<6>("cvt bf16 + unpack + add", 1); run<7>("v_alignbit_b32", 1);
        } else {
            run<0>("v_fma_f32", 4); run<1>("v_pk_fma_f32", 4); run<8>("v_mul_f32", 4); run<9>("v_pk_add_f32", 4); run<2>("v_exp_f32", 4); run<3>("v_rcp_f32", 4);
            run<4>("v_dot2c_f32_bf16", 4); run<5>("lshl+xor (2 int ops)", 4); run<6>("cvt bf16 + unpack + add", 4); run<7>("v_alignbit_b32", 4);
        }
    }
    return 0;
}
