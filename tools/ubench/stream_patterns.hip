// HBM efficiency of the access patterns the streaming kernels use (round 3): a [M][E] bf16 matrix copied
//   mode 0: flat, 16 bytes per lane, grid-stride (the "device copy" ceiling)
//   mode 1: channel-sliced like the library's streaming kernels: blockIdx.y = 64-channel slice (128 bytes of a row),
//           8 lanes per row, 32 rows per workgroup iteration, persistent grid
//   mode 2: same, SW adjacent slices per workgroup (SW*128 contiguous bytes per row): SW = 2, 4
//   mode 3: whole rows per wave (E*2 contiguous bytes)
// Build: hipcc -O3 --offload-arch=gfx950 stream_patterns.hip -o stream_patterns ; run: ./stream_patterns
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

__global__ void k_flat(const uint4* __restrict__ in, uint4* __restrict__ out, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) out[i] = in[i];
}
template <int SW, int U = 4>
__global__ __launch_bounds__(256) void k_sliced(const unsigned short* __restrict__ in, unsigned short* __restrict__ out, long long M, int E) {
    constexpr int LPR = 8 * SW;                       // lanes per row
    const int tid = threadIdx.x, cv = tid % LPR, pl = tid / LPR;
    const int c = blockIdx.y * 64 * SW + cv * 8;
    if (c >= E) return;
    constexpr int RPB = 256 / LPR;
    for (long long r0 = (long long)blockIdx.x * RPB + pl; r0 < M; r0 += (long long)U * gridDim.x * RPB) {
        uint4 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) { long long r = r0 + (long long)u * gridDim.x * RPB; v[u] = *reinterpret_cast<const uint4*>(in + (r < M ? r : r0) * E + c); }
#pragma unroll
        for (int u = 0; u < U; ++u) { long long r = r0 + (long long)u * gridDim.x * RPB; if (r < M) *reinterpret_cast<uint4*>(out + r * E + c) = v[u]; }
    }
}
__global__ __launch_bounds__(256) void k_rows(const unsigned short* __restrict__ in, unsigned short* __restrict__ out, long long M, int E) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int cpr = E / 8;                            // 16-byte chunks per row
    for (long long r0 = ((long long)blockIdx.x * 4 + wave) * 4; r0 < M; r0 += (long long)gridDim.x * 16) {
        uint4 v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) if (lane < cpr) v[u] = *reinterpret_cast<const uint4*>(in + (r0 + u < M ? r0 + u : r0) * E + lane * 8);
#pragma unroll
        for (int u = 0; u < 4; ++u) if (lane < cpr && r0 + u < M) *reinterpret_cast<uint4*>(out + (r0 + u) * E + lane * 8) = v[u];
    }
}

int main() {
    const long long M = 589824; const int Es[3] = {448, 896, 1792};
    for (int ei = 0; ei < 3; ++ei) {
        const int E = Es[ei];
        const long long Mx = ei == 0 ? M : (ei == 1 ? M / 4 * 2 : M / 4);      // keep ~0.5 GB per tensor
        size_t bytes = (size_t)Mx * E * 2;
        unsigned short *in, *out;
        hipMalloc(&in, bytes); hipMalloc(&out, bytes);
        hipMemset(in, 1, bytes);
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        auto time = [&](auto launch, const char* name) {
            for (int i = 0; i < 3; ++i) launch();
            hipEventRecord(e0);
            for (int i = 0; i < 10; ++i) launch();
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 10;
            printf("E=%4d rows=%lld %-28s %8.1f us  %6.2f TB/s (read+write)\n", E, Mx, name, ms * 1e3, 2.0 * bytes / ms / 1e9);
        };
        time([&] { hipLaunchKernelGGL(k_flat, dim3(256 * 8), dim3(256), 0, 0, (const uint4*)in, (uint4*)out, bytes / 16); }, "flat 16B/lane");
        const int s1 = E / 64;
        time([&] { hipLaunchKernelGGL(k_sliced<1>, dim3(2048 / s1, s1), dim3(256), 0, 0, in, out, Mx, E); }, "sliced 128 B x 8 blocks/CU");
        time([&] { hipLaunchKernelGGL(k_sliced<1>, dim3(1024 / s1, s1), dim3(256), 0, 0, in, out, Mx, E); }, "sliced 128 B x 4 blocks/CU");
        for (int bpc = 1; bpc <= 8; ++bpc) {
            char nm[64];
            snprintf(nm, sizeof nm, "sliced U=2 %d blocks/CU", bpc);
            time([&] { hipLaunchKernelGGL((k_sliced<1, 2>), dim3(256 * bpc / s1, s1), dim3(256), 0, 0, in, out, Mx, E); }, nm);
            snprintf(nm, sizeof nm, "sliced U=4 %d blocks/CU", bpc);
            time([&] { hipLaunchKernelGGL((k_sliced<1, 4>), dim3(256 * bpc / s1, s1), dim3(256), 0, 0, in, out, Mx, E); }, nm);
            snprintf(nm, sizeof nm, "sliced U=8 %d blocks/CU", bpc);
            time([&] { hipLaunchKernelGGL((k_sliced<1, 8>), dim3(256 * bpc / s1, s1), dim3(256), 0, 0, in, out, Mx, E); }, nm);
        }
        time([&] { hipLaunchKernelGGL(k_sliced<2>, dim3(2048 / ((s1 + 1) / 2), (s1 + 1) / 2), dim3(256), 0, 0, in, out, Mx, E); }, "sliced 256 B");
        time([&] { hipLaunchKernelGGL(k_sliced<4>, dim3(2048 / ((s1 + 3) / 4), (s1 + 3) / 4), dim3(256), 0, 0, in, out, Mx, E); }, "sliced 512 B");
        if (E <= 448) time([&] { hipLaunchKernelGGL(k_rows, dim3(2048), dim3(256), 0, 0, in, out, Mx, E); }, "whole rows per wave");
        hipFree(in); hipFree(out);
    }
    return 0;
}
