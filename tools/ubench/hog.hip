// A resident "communication-like" kernel for interference experiments: `wgs` workgroups of `threads` threads that
// hold their CU slots for `ms` milliseconds (wall clock, s_sleep polling — no memory traffic, next to no issue slots), launched on a
// caller-supplied stream.  Emulates what an RCCL all-reduce kernel resident beside the training kernels does to the persistent grids
// (sized to fill the chip): the displaced workgroups run as a second partial wave.
//   hipcc -O2 --offload-arch=gfx950 -shared -fPIC tools/ubench/hog.hip -o tools/ubench/libhog.so
#include <hip/hip_runtime.h>

__global__ void hog_kernel(long long ticks, int regs_pad) {
    // keep some VGPRs live so that the occupancy footprint resembles a real kernel's (~64 VGPRs)
    float keep[48];
#pragma unroll
    for (int i = 0; i < 48; ++i) keep[i] = (float)(threadIdx.x + i * regs_pad);
    const long long t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(64);
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 48; ++i) s += keep[i];
    if (s == -1.f) __builtin_trap();
}

extern "C" int hog_launch(int wgs, int threads, double ms, void* stream) {
    int rate_khz = 100000;                                   // wall_clock64 ticks at 100 MHz on gfx9
    (void)hipDeviceGetAttribute(&rate_khz, hipDeviceAttributeWallClockRate, 0);
    const long long ticks = (long long)(ms * (double)rate_khz);
    hipLaunchKernelGGL(hog_kernel, dim3(wgs), dim3(threads), 0, (hipStream_t)stream, ticks, 1);
    return (int)hipGetLastError();
}
