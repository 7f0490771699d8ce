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
// two floats -> one dword of two bf16 (lo in bits 0-15): ONE v_cvt_pk_bf16_f32 (RNE, NaN-preserving).  Packing two
// separately converted halves with shift/or costs four VALU instructions instead.
typedef float dwn_f32x2_t __attribute__((ext_vector_type(2)));
typedef __bf16 dwn_bf16x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t pk_bf16(float lo, float hi) {
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(dwn_f32x2_t{lo, hi}, dwn_bf16x2_t));
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
    r.x = pk_bf16(v[0], v[1]);
    r.y = pk_bf16(v[2], v[3]);
    r.z = pk_bf16(v[4], v[5]);
    r.w = pk_bf16(v[6], v[7]);
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

// ---- small exact integer division by a workgroup-uniform divisor (flat pixel index -> row, column).
// floor(f / d) == trunc(f * (1/d) + 0.5/d) for 0 <= f < 2^22: (f + 0.5)/d is at least 0.5/d away from every integer
// and the fp32 error of the fma is below (f/d) * 2^-23.  3 full-rate VALU ops instead of the ~12-op udiv expansion.
struct FastDiv {
    int d;
    float inv, hinv;
    __device__ __forceinline__ explicit FastDiv(int d_) : d(d_), inv(1.0f / (float)d_), hinv(0.5f / (float)d_) {}
    __device__ __forceinline__ int div(int f) const { return (int)fmaf((float)f, inv, hinv); }
    // remainder with the 24-bit multiplier (full rate; operands are pixel counts < 2^24)
    __device__ __forceinline__ int rem(int f, int q) const { return f - __mul24(q, d); }
};

// ---- exact unsigned division by a kernel-uniform divisor for 0 <= n < 2^31 (row indices): one v_mul_hi_u32 and a
// shift instead of the ~80-instruction 64-bit division the row -> (frame, y, x) decodes used to cost per row.
// With s = ceil(log2 d) and m = ceil(2^(31+s) / d) (< 2^32), floor(n / d) = (n * m) >> (31 + s) for every n < 2^31.
struct UDiv32 {
    unsigned d, m;
    int sh;
    __device__ __forceinline__ explicit UDiv32(unsigned d_) : d(d_) {
        const int s = d_ <= 1 ? 1 : 32 - __clz((int)(d_ - 1));
        sh = s - 1;
        m = d_ <= 1 ? 0u : (unsigned)((((unsigned long long)1 << (31 + s)) + d_ - 1) / d_);
    }
    __device__ __forceinline__ unsigned div(unsigned n) const { return d <= 1 ? n : (__umulhi(n, m) >> sh); }
};
// row of a [frames][H][W] raster -> (frame, y, x)
struct RasterIdx {
    UDiv32 dw, dh;
    unsigned W, H;
    __device__ __forceinline__ RasterIdx(int H_, int W_) : dw((unsigned)W_), dh((unsigned)H_), W((unsigned)W_), H((unsigned)H_) {}
    __device__ __forceinline__ void decode(unsigned row, unsigned& f, int& y, int& x) const {
        const unsigned r2 = dw.div(row);
        x = (int)(row - r2 * W);
        f = dh.div(r2);
        y = (int)(r2 - f * H);
    }
};

// ---- activation math
// v_exp_f32 + v_rcp_f32 (1 ulp each): an IEEE fp32 division would cost ~10 VALU instructions per element
__device__ __forceinline__ float sigmoidf_(float h) { return __builtin_amdgcn_rcpf(1.0f + __expf(-h)); }
__device__ __forceinline__ float siluf_(float h) { return h * sigmoidf_(h); }
// two sigmoids with the multiply and the add packed (v_pk_mul_f32 / v_pk_add_f32 are the scalar instructions' arithmetic, two lanes
// at a time; left to itself hipcc emits them per element: 2 of the 4.5 instructions of a SiLU).  Bit-identical to sigmoidf_:
// __expf(-h) = exp2(-h * log2(e)) = exp2(h * -log2(e)).
typedef float dwn_f2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ dwn_f2_t sigmoid2f_(dwn_f2_t h) {
    const dwn_f2_t t = h * dwn_f2_t{-0x1.715476p+0f, -0x1.715476p+0f};
    const dwn_f2_t d = dwn_f2_t{__builtin_amdgcn_exp2f(t.x), __builtin_amdgcn_exp2f(t.y)} + dwn_f2_t{1.0f, 1.0f};
    return dwn_f2_t{__builtin_amdgcn_rcpf(d.x), __builtin_amdgcn_rcpf(d.y)};
}
// N sigmoids (N even), pairwise
template <int N> __device__ __forceinline__ void sigmoid_n(const float* h, float* s) {
    static_assert(N % 2 == 0, "pairs");
#pragma unroll
    for (int i = 0; i < N; i += 2) {
        const dwn_f2_t r = sigmoid2f_(dwn_f2_t{h[i], h[i + 1]});
        s[i] = r.x; s[i + 1] = r.y;
    }
}
// SqueezeExcite pooling sums are accumulated as 64-bit fixed point in units of 2^-32: integer addition is associative, so the
// sums do not depend on the order in which lanes, waves and workgroups add (fp32 atomics made the gate — and through bf16
// roundings every later activation — differ from run to run).  A float of magnitude >= 2^-8 converts exactly; smaller ones
// lose bits below 2^-32, the same bits in every run.  |sum| < 2^31.
__device__ __forceinline__ long long pool_fix(float v) { return (long long)(v * 4294967296.0f); }
__device__ __forceinline__ float pool_unfix(long long s) { return (float)((double)s * (1.0 / 4294967296.0)); }
// d silu(h) / dh
__device__ __forceinline__ float silu_gradf_(float h) {
    float s = sigmoidf_(h);
    return s * (1.0f + h * (1.0f - s));
}
// v[i] = silu(v[i]) / g[i] = silu'(h[i]) for N (even) values: the same arithmetic as siluf_ / silu_gradf_, sigmoids in pairs
template <int N> __device__ __forceinline__ void silu_n(float* v) {
    float s[N];
    sigmoid_n<N>(v, s);
#pragma unroll
    for (int i = 0; i < N; ++i) v[i] = v[i] * s[i];
}
template <int N> __device__ __forceinline__ void silu_grad_n(const float* h, float* g) {
    float s[N];
    sigmoid_n<N>(h, s);
#pragma unroll
    for (int i = 0; i < N; ++i) g[i] = s[i] * (1.0f + h[i] * (1.0f - s[i]));
}

// ---- 4-channel vectors (16 B fp32 / 8 B bf16): the unit of the streaming (stencil / elementwise) kernels.
// Half the registers per thread of the 16-byte bf16 vector -> twice the waves in flight to hide HBM latency.
template <typename T> struct V4 { };
template <> struct V4<float> {
    typedef uint4 raw_t;
    static constexpr int NCV = 8;       // vectors per 128-byte channel slice
    static __device__ __forceinline__ void unpack(const raw_t& r, float* o) {
        o[0] = __uint_as_float(r.x); o[1] = __uint_as_float(r.y); o[2] = __uint_as_float(r.z); o[3] = __uint_as_float(r.w);
    }
    static __device__ __forceinline__ raw_t pack(const float* v) {
        return make_uint4(__float_as_uint(v[0]), __float_as_uint(v[1]), __float_as_uint(v[2]), __float_as_uint(v[3]));
    }
    static __device__ __forceinline__ raw_t zero() { return make_uint4(0, 0, 0, 0); }
};
template <> struct V4<bf16_t> {
    typedef uint2 raw_t;
    static constexpr int NCV = 16;
    static __device__ __forceinline__ void unpack(const raw_t& r, float* o) {
        o[0] = __uint_as_float(r.x << 16); o[1] = __uint_as_float(r.x & 0xffff0000u);
        o[2] = __uint_as_float(r.y << 16); o[3] = __uint_as_float(r.y & 0xffff0000u);
    }
    static __device__ __forceinline__ raw_t pack(const float* v) {
        uint2 r;
        r.x = pk_bf16(v[0], v[1]);
        r.y = pk_bf16(v[2], v[3]);
        return r;
    }
    static __device__ __forceinline__ raw_t zero() { return make_uint2(0, 0); }
};
template <typename T> __device__ __forceinline__ typename V4<T>::raw_t ld4_raw(const T* p) {
    return *reinterpret_cast<const typename V4<T>::raw_t*>(p);
}
template <typename T> __device__ __forceinline__ void ld4(const T* p, float* o) { V4<T>::unpack(ld4_raw<T>(p), o); }
template <typename T> __device__ __forceinline__ void st4(T* p, const float* v) {
    *reinterpret_cast<typename V4<T>::raw_t*>(p) = V4<T>::pack(v);
}
__device__ __forceinline__ void ldc4(const float* p, float* o) {
    float4 v = *reinterpret_cast<const float4*>(p);
    o[0] = v.x; o[1] = v.y; o[2] = v.z; o[3] = v.w;
}

// ---- the operand "loader": how a kernel reads one 16-byte channel vector of an operand.
// Fusing the producer's batch-norm / activation / gate / positional-encoding / BN-backward affine
// into the consumer's load is what removes the elementwise HBM round trips.
enum { LD_PLAIN = 0, LD_PE = 1, LD_BNACT = 2, LD_AFFINE2 = 3, LD_DY3 = 4, LD_GATE = 5, LD_CAT1 = 6 };

typedef dwn_load_desc LoadDesc;   // field meanings: include/dwn.h and the loader below

template <int KIND, typename T>
__device__ __forceinline__ void load_op(const LoadDesc& d, i64 row, int col, float* o) {
    constexpr int KC = TT<T>::KC;
    const T* p = reinterpret_cast<const T*>(d.p) + row * d.ld + col;
    ld_vec<T>(p, o);
    // row indices fit 32 bits (rows = B*T*H*W < 2^31): 32-bit unsigned div/mod is ~10x cheaper than the i64 form
    const unsigned r32 = (unsigned)row;
    if constexpr (KIND == LD_PE) {
        const unsigned r2 = r32 / (unsigned)d.pW;
        const int w = (int)(r32 - r2 * (unsigned)d.pW);
        const unsigned r3 = r2 / (unsigned)d.pH;
        const int h = (int)(r2 - r3 * (unsigned)d.pH);
        const int t = (int)(r3 % (unsigned)d.pT);
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
        for (int i = 0; i < KC; ++i) o[i] = fmaf(o[i], s[i], t[i]);
        if (d.act) silu_n<KC>(o);
        if (d.gate) {
            int b = (int)(r32 / (unsigned)d.rows_per_sample);
            float g[KC];
            ld_coef<KC>(d.gate + (i64)b * d.gate_ld + col, g);
#pragma unroll
            for (int i = 0; i < KC; ++i) o[i] *= g[i];
        }
    } else if constexpr (KIND == LD_GATE) {
        int b = (int)(r32 / (unsigned)d.rows_per_sample);
        float g[KC];
        ld_coef<KC>(d.gate + (i64)b * d.gate_ld + col, g);
#pragma unroll
        for (int i = 0; i < KC; ++i) o[i] *= g[i];
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
        int b = (int)(r32 / (unsigned)d.rows_per_sample);
        ld_coef<KC>(d.gate + (i64)b * d.gate_ld + col, g);
        ld_coef<KC>(d.gate2 + (i64)b * d.gate_ld + col, g2);
        float h[KC], sp[KC];
#pragma unroll
        for (int i = 0; i < KC; ++i) h[i] = fmaf(y[i], s[i], t[i]);
        silu_grad_n<KC>(h, sp);
#pragma unroll
        for (int i = 0; i < KC; ++i) {
            float dh = fmaf(o[i], g[i], g2[i]) * sp[i];
            o[i] = fmaf(a1[i], dh, fmaf(a2[i], y[i], a3[i]));
        }
    }
}

// ---- hoisted form for kernels where a thread's channel column is fixed while it walks rows (GEMM staging):
// the per-channel coefficient vectors are loaded once (ColCoef::load) instead of once per 16-byte data chunk —
// otherwise every data load drags 4-6 coefficient loads through the L1/TA path, which then bounds the kernel.
template <int KIND, typename T> struct ColCoef {
    static constexpr int KC = TT<T>::KC;
    float c1[KC], c2[KC], c3[KC], c4[KC], c5[KC], g[KC], g2[KC];
    int gb;                                    // sample whose gate vectors are cached (-1: none)
    int col;
    __device__ __forceinline__ void load(const LoadDesc& d, int column) {
        col = column; gb = -1;
        if constexpr (KIND == LD_BNACT) { ld_coef<KC>(d.v1 + col, c1); ld_coef<KC>(d.v2 + col, c2); }
        if constexpr (KIND == LD_AFFINE2 || KIND == LD_DY3) {
            ld_coef<KC>(d.v1 + col, c1); ld_coef<KC>(d.v2 + col, c2); ld_coef<KC>(d.v3 + col, c3);
        }
        if constexpr (KIND == LD_DY3) { ld_coef<KC>(d.v4 + col, c4); ld_coef<KC>(d.v5 + col, c5); }
    }
    __device__ __forceinline__ void gate_for(const LoadDesc& d, unsigned row) {
        int b = (int)(row / (unsigned)d.rows_per_sample);
        if (b != gb) {
            gb = b;
            ld_coef<KC>(d.gate + (i64)b * d.gate_ld + col, g);
            if constexpr (KIND == LD_DY3) ld_coef<KC>(d.gate2 + (i64)b * d.gate_ld + col, g2);
        }
    }
    // p, q: raw 16-byte vectors already loaded (q unused for single-tensor kinds)
    __device__ __forceinline__ uint4 apply(const LoadDesc& d, unsigned row, const uint4& p, const uint4& q) {
        if constexpr (KIND == LD_PLAIN) { return p; }
        float o[KC];
        unpack16<T>(p, o);
        if constexpr (KIND == LD_GATE) {
            gate_for(d, row);
#pragma unroll
            for (int i = 0; i < KC; ++i) o[i] *= g[i];
        } else if constexpr (KIND == LD_BNACT) {
#pragma unroll
            for (int i = 0; i < KC; ++i) o[i] = fmaf(o[i], c1[i], c2[i]);
            if (d.act) silu_n<KC>(o);
            if (d.gate) {
                gate_for(d, row);
#pragma unroll
                for (int i = 0; i < KC; ++i) o[i] *= g[i];
            }
        } else if constexpr (KIND == LD_AFFINE2) {
            float y[KC];
            unpack16<T>(q, y);
#pragma unroll
            for (int i = 0; i < KC; ++i) o[i] = fmaf(c1[i], o[i], fmaf(c2[i], y[i], c3[i]));
        } else if constexpr (KIND == LD_DY3) {
            float y[KC];
            unpack16<T>(q, y);
            gate_for(d, row);
            float h[KC], sp[KC];
#pragma unroll
            for (int i = 0; i < KC; ++i) h[i] = fmaf(y[i], c4[i], c5[i]);
            silu_grad_n<KC>(h, sp);
#pragma unroll
            for (int i = 0; i < KC; ++i) {
                float dh = fmaf(o[i], g[i], g2[i]) * sp[i];
                o[i] = fmaf(c1[i], dh, fmaf(c2[i], y[i], c3[i]));
            }
        }
        return pack16<T>(o);
    }
    static constexpr bool two_tensors = (KIND == LD_AFFINE2 || KIND == LD_DY3);
};

// ---- deterministic build (-DDWN_DETERMINISTIC -> libdwiseneuro_hip_det.so, selected with DWN_DETERMINISTIC=1; SURVEY.md §5
// "deterministic re-run equality tests").  The normal build leaves two things to arrival order: which WAVE of a workgroup adds
// first to a shared LDS word, and which WORKGROUP adds first to a global word.  Here
//   * every region of LDS float atomics is executed one wave after the other (DET_WAVES_BEGIN / _END: wave 0, barrier,
//     wave 1, ...; lanes of one wave instruction that hit the same word are serialised by the LDS in lane order), and
//   * a workgroup performs its global float atomics only while it holds the launch's ticket (DET_ENTER / DET_EXIT), handed
//     on in linear workgroup-id order — workgroups are dispatched in that order, so the holder of an earlier ticket is always
//     resident or finished.  A workgroup keeps the ticket from its first global atomic to its exit; kernels that flush
//     inside their main loop therefore run one workgroup after the other.  Slow by design (a debugging build).
// Every workgroup of a kernel that uses the ticket must pass it, whatever path it takes to its exit.
#ifdef DWN_DETERMINISTIC
static __device__ unsigned dwn_det_ticket_ = 0;          // one per translation unit; kernels of a stream run one at a time
__device__ __forceinline__ unsigned det_wg_id_() { return blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z); }
__device__ __forceinline__ void det_enter_() {
    if (threadIdx.x == 0) {
        const unsigned me = det_wg_id_();
        while (__hip_atomic_load(&dwn_det_ticket_, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) != me) __builtin_amdgcn_s_sleep(16);
    }
    __syncthreads();
}
__device__ __forceinline__ void det_exit_() {             // waits for the turn first: also the pass of a workgroup with nothing to add
    det_enter_();
    if (threadIdx.x == 0) {
        __threadfence();
        unsigned next = det_wg_id_() + 1;
        if (next == gridDim.x * gridDim.y * gridDim.z) next = 0;           // leave the ticket ready for the next launch
        __hip_atomic_store(&dwn_det_ticket_, next, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    }
}
#define DET_WAVES_BEGIN for (unsigned dw__ = 0; dw__ < (blockDim.x >> 6); ++dw__) { if ((threadIdx.x >> 6) == dw__) {
#define DET_WAVES_END } __syncthreads(); }
#define DET_ENTER() det_enter_()
#define DET_EXIT() det_exit_()
#define DET_ONLY(...) __VA_ARGS__
#else
#define DET_WAVES_BEGIN {
#define DET_WAVES_END }
#define DET_ENTER() ((void)0)
#define DET_EXIT() ((void)0)
#define DET_ONLY(...)
#endif

// ---- cross-workgroup statistics: double atomics into one of DWN_NREP replicas
__device__ __forceinline__ void stat_add(double* base, int rep, int nchan, int which, int c, float v) {
    atomicAdd(base + ((i64)rep * 2 + which) * nchan + c, (double)v);
}

#define DWN_CHECK_LAUNCH() do { hipError_t e__ = hipGetLastError(); if (e__ != hipSuccess) return dwn_set_error((int)e__, hipGetErrorString(e__)); } while (0)

int dwn_set_error(int code, const char* msg);
