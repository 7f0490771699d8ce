// Shared device-side helpers for the DwiseNeuro gfx950 kernels.
//
// Data model: every activation is a channels-last matrix [rows][C] with rows = (b, t, h, w) and the
// channel dimension contiguous ("NDHWC").  Storage type T is float (parity mode) or bf16 (raw
// uint16_t bits, fast mode); all statistics, per-channel coefficients and accumulators are fp32
// (fp64 for the cross-workgroup batch-norm sums).  One "vector" is 16 bytes of T: KC = 4 floats or
// 8 bf16 — the unit of every global load/store (coalesced along C) and of every MFMA fragment read.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/dwn.h"

typedef uint16_t bf16_t;
typedef long long i64;


template <typename T> struct TT;
template <> struct TT<float>  { static constexpr int KC = 4; static constexpr int IS_BF16 = 0; };
template <> struct TT<bf16_t> { static constexpr int KC = 8; static constexpr int IS_BF16 = 1; };

__device__ __forceinline__ float bf2f(bf16_t v) { return __uint_as_float(((uint32_t)v) << 16); }
__device__ __forceinline__ bf16_t f2bf(float f) {
    // plain cast: hipcc emits v_cvt_pk_bf16_f32 (RNE, NaN-preserving) on gfx950
    __bf16 b = (__bf16)f;
    return __builtin_bit_cast(bf16_t, b);
}
template <typename T> __device__ __forceinline__ float to_f(T v);
template <> __device__ __forceinline__ float to_f<float>(float v) { return v; }
template <> __device__ __forceinline__ float to_f<bf16_t>(bf16_t v) { return bf2f(v); }
template <typename T> __device__ __forceinline__ T from_f(float v);
template <> __device__ __forceinline__ float from_f<float>(float v) { return v; }
template <> __device__ __forceinline__ bf16_t from_f<bf16_t>(float v) { return f2bf(v); }
// value as it will read back from storage
template <typename T> __device__ __forceinline__ float round_t(float v) { return to_f<T>(from_f<T>(v)); }

// ---- 16-byte vector <-> float[KC]
template <typename T> __device__ __forceinline__ void unpack16(const uint4& raw, float* o);
template <> __device__ __forceinline__ void unpack16<float>(const uint4& raw, float* o) {
    o[0] = __uint_as_float(raw.x); o[1] = __uint_as_float(raw.y);
    o[2] = __uint_as_float(raw.z); o[3] = __uint_as_float(raw.w);
}
template <> __device__ __forceinline__ void unpack16<bf16_t>(const uint4& raw, float* o) {
    o[0] = __uint_as_float(raw.x << 16); o[1] = __uint_as_float(raw.x & 0xffff0000u);
    o[2] = __uint_as_float(raw.y << 16); o[3] = __uint_as_float(raw.y & 0xffff0000u);
    o[4] = __uint_as_float(raw.z << 16); o[5] = __uint_as_float(raw.z & 0xffff0000u);
    o[6] = __uint_as_float(raw.w << 16); o[7] = __uint_as_float(raw.w & 0xffff0000u);
}
template <typename T> __device__ __forceinline__ uint4 pack16(const float* v);
template <> __device__ __forceinline__ uint4 pack16<float>(const float* v) {
    return make_uint4(__float_as_uint(v[0]), __float_as_uint(v[1]), __float_as_uint(v[2]), __float_as_uint(v[3]));
}
template <> __device__ __forceinline__ uint4 pack16<bf16_t>(const float* v) {
    uint4 r;
    r.x = (uint32_t)f2bf(v[0]) | ((uint32_t)f2bf(v[1]) << 16);
    r.y = (uint32_t)f2bf(v[2]) | ((uint32_t)f2bf(v[3]) << 16);
    r.z = (uint32_t)f2bf(v[4]) | ((uint32_t)f2bf(v[5]) << 16);
    r.w = (uint32_t)f2bf(v[6]) | ((uint32_t)f2bf(v[7]) << 16);
    return r;
}
template <typename T> __device__ __forceinline__ void ld_vec(const T* p, float* o) {
    uint4 raw = *reinterpret_cast<const uint4*>(p);
    unpack16<T>(raw, o);
}
template <typename T> __device__ __forceinline__ void st_vec(T* p, const float* v) {
    *reinterpret_cast<uint4*>(p) = pack16<T>(v);
}
// KC consecutive fp32 per-channel coefficients
template <int KC> __device__ __forceinline__ void ld_coef(const float* p, float* o) {
#pragma unroll
    for (int i = 0; i < KC; i += 4) {
        float4 v = *reinterpret_cast<const float4*>(p + i);
        o[i] = v.x; o[i + 1] = v.y; o[i + 2] = v.z; o[i + 3] = v.w;
    }
}

// ---- activation math
__device__ __forceinline__ float sigmoidf_(float h) { return 1.0f / (1.0f + __expf(-h)); }
__device__ __forceinline__ float siluf_(float h) { return h * sigmoidf_(h); }
// d silu(h) / dh
__device__ __forceinline__ float silu_gradf_(float h) {
    float s = sigmoidf_(h);
    return s * (1.0f + h * (1.0f - s));
}

// ---- the operand "loader": how a kernel reads one 16-byte channel vector of an operand.
// Fusing the producer's batch-norm / activation / gate / positional-encoding / BN-backward affine
// into the consumer's load is what removes the elementwise HBM round trips.
enum { LD_PLAIN = 0, LD_PE = 1, LD_BNACT = 2, LD_AFFINE2 = 3, LD_DY3 = 4 };

typedef dwn_load_desc LoadDesc;   // field meanings: include/dwn.h and the loader below

template <int KIND, typename T>
__device__ __forceinline__ void load_op(const LoadDesc& d, i64 row, int col, float* o) {
    constexpr int KC = TT<T>::KC;
    const T* p = reinterpret_cast<const T*>(d.p) + row * d.ld + col;
    ld_vec<T>(p, o);
    if constexpr (KIND == LD_PE) {
        int w = (int)(row % d.pW);
        i64 r2 = row / d.pW;
        int h = (int)(r2 % d.pH);
        int t = (int)((r2 / d.pH) % d.pT);
        float a[KC], b[KC], c[KC];
        ld_coef<KC>(d.pe_t + (i64)t * d.pe_ld + col, a);
        ld_coef<KC>(d.pe_h + (i64)h * d.pe_ld + col, b);
        ld_coef<KC>(d.pe_w + (i64)w * d.pe_ld + col, c);
#pragma unroll
        for (int i = 0; i < KC; ++i) o[i] = o[i] + ((a[i] + b[i]) + c[i]);
    } else if constexpr (KIND == LD_BNACT) {
        float s[KC], t[KC];
        ld_coef<KC>(d.v1 + col, s);
        ld_coef<KC>(d.v2 + col, t);
#pragma unroll
        for (int i = 0; i < KC; ++i) {
            float h = fmaf(o[i], s[i], t[i]);
            o[i] = d.act ? siluf_(h) : h;
        }
        if (d.gate) {
            int b = (int)(row / d.rows_per_sample);
            float g[KC];
            ld_coef<KC>(d.gate + (i64)b * d.gate_ld + col, g);
#pragma unroll
            for (int i = 0; i < KC; ++i) o[i] *= g[i];
        }
    } else if constexpr (KIND == LD_AFFINE2) {
        float y[KC], a1[KC], a2[KC], a3[KC];
        ld_vec<T>(reinterpret_cast<const T*>(d.q) + row * d.ld + col, y);
        ld_coef<KC>(d.v1 + col, a1);
        ld_coef<KC>(d.v2 + col, a2);
        ld_coef<KC>(d.v3 + col, a3);
#pragma unroll
        for (int i = 0; i < KC; ++i) o[i] = fmaf(a1[i], o[i], fmaf(a2[i], y[i], a3[i]));
    } else if constexpr (KIND == LD_DY3) {
        // o = du (grad wrt gated SE output).  dz3 = du*gate + dpS; dh3 = dz3*silu'(h3); dy3 = A1 dh3 + A2 y3 + A3
        float y[KC], a1[KC], a2[KC], a3[KC], s[KC], t[KC], g[KC], g2[KC];
        ld_vec<T>(reinterpret_cast<const T*>(d.q) + row * d.ld + col, y);
        ld_coef<KC>(d.v1 + col, a1);
        ld_coef<KC>(d.v2 + col, a2);
        ld_coef<KC>(d.v3 + col, a3);
        ld_coef<KC>(d.v4 + col, s);
        ld_coef<KC>(d.v5 + col, t);
        int b = (int)(row / d.rows_per_sample);
        ld_coef<KC>(d.gate + (i64)b * d.gate_ld + col, g);
        ld_coef<KC>(d.gate2 + (i64)b * d.gate_ld + col, g2);
#pragma unroll
        for (int i = 0; i < KC; ++i) {
            float h = fmaf(y[i], s[i], t[i]);
            float dh = fmaf(o[i], g[i], g2[i]) * silu_gradf_(h);
            o[i] = fmaf(a1[i], dh, fmaf(a2[i], y[i], a3[i]));
        }
    }
}

// ---- cross-workgroup statistics: double atomics into one of DWN_NREP replicas
__device__ __forceinline__ void stat_add(double* base, int rep, int nchan, int which, int c, float v) {
    atomicAdd(base + ((i64)rep * 2 + which) * nchan + c, (double)v);
}

#define DWN_CHECK_LAUNCH() do { hipError_t e__ = hipGetLastError(); if (e__ != hipSuccess) return dwn_set_error((int)e__, hipGetErrorString(e__)); } while (0)

int dwn_set_error(int code, const char* msg);
