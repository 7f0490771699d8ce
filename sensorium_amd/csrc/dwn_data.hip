// Batch assembly on the device: the per-sample work the reference does in 8 DataLoader workers on the CPU —
// StackInputsProcessor (src/inputs.py:15-36), responses_to_tensor (src/responses.py:25-29), CutMix
// (src/mixers.py:52-67), ConcatMiceVideoDataset.construct_mice_sample (src/datasets.py:172-187) and the default
// collate — as two launches over trials kept resident in HBM in their on-disk layout (video [H0][W0][L],
// behavior / pupil_center [2][L], responses [N][L]; src/datasets.py:37-51).
//
// Pure byte/float moves: bit-exact against the reference (the only arithmetic is the cut-mix target blend, done
// with separately rounded mul, mul, add as torch does).  HBM-bound; the work per step is tiny next to the model
// (47 MB of inputs written, ~1 MB of video read for the metric batch) — the point is removing the host pipeline and
// the 200 MB/step host-to-device copy, not the kernel time.
#include "dwn_internal.h"
#include "dwn_kernels.h"

// hipcc contracts a*b + c into an FMA by default and HIP's __fmul_rn / __fadd_rn are plain operators (a
// `#pragma clang fp contract(off)` did not stop it either): the blends below must round the two products and the sum
// separately, as torch's three element-wise kernels do, so every product passes through an opaque register barrier.
__device__ __forceinline__ float rounded(float x) {
    asm volatile("" : "+v"(x));
    return x;
}
__device__ __forceinline__ float blend3(float a, float x, float b, float y) { return rounded(a * x) + rounded(b * y); }

namespace {

__device__ __forceinline__ float video_at(const dwn_clip_src& s, long long off) {
    return s.video_dtype == DWN_VID_U8 ? (float)((const unsigned char*)s.video)[off] : ((const float*)s.video)[off];
}

// One workgroup per (sample b, output row y).  Channel 0: the needed [W0][T] slab of the video row is read with the
// frame axis fastest (it is the contiguous axis on disk), transposed through LDS and written as T contiguous rows.
// Channels 1-4: per-frame scalars broadcast over the row.  Inside the cut-mix box every channel comes from `mix`;
// mix_mode == DWN_MIX_BLEND (Mixup, src/mixers.py:22-33): every element of all five channels is
// one_minus_lam*src + lam*mix (three separately rounded fp32 operations, pad pixels included, like torch).
__global__ __launch_bounds__(256) void assemble_inputs_kernel(const dwn_clip_desc* __restrict__ descs, int T, int H0,
                                                              int W0, int H, int W, float pad, float* __restrict__ x) {
    extern __shared__ float sm[];
    const int b = blockIdx.x / H, y = blockIdx.x % H;
    const dwn_clip_desc d = descs[b];
    const int hs = (H - H0) / 2, ws = (W - W0) / 2;
    const int yy = y - hs;
    const bool row_in_video = yy >= 0 && yy < H0;
    const bool blend = d.mix.valid && d.mix_mode == DWN_MIX_BLEND;
    const bool row_in_box = d.mix.valid && !blend && y >= d.bbx1 && y < d.bbx2;
    float* tile = sm;                          // [T][W0 + 1]
    float* scal = sm + (size_t)T * (W0 + 1);   // [2 sources][4 channels][T]
    float* tile2 = scal + 8 * (size_t)T;       // [T][W0 + 1] of `mix` (blend mode only)
    const int tw = W0 + 1;
    if (row_in_video) {
        for (int i = threadIdx.x; i < W0 * T; i += 256) {
            int t = i % T, xx = i / T;
            int xo = xx + ws;
            const bool from_mix = row_in_box && xo >= d.bby1 && xo < d.bby2;
            const dwn_clip_src& s = from_mix ? d.mix : d.src;
            long long f = s.frame_start + (long long)t * s.frame_step;
            tile[t * tw + xx] = video_at(s, ((long long)yy * W0 + xx) * s.length + f);
            if (blend) {
                long long f2 = d.mix.frame_start + (long long)t * d.mix.frame_step;
                tile2[t * tw + xx] = video_at(d.mix, ((long long)yy * W0 + xx) * d.mix.length + f2);
            }
        }
    }
    for (int i = threadIdx.x; i < 8 * T; i += 256) {
        int t = i % T, c = (i / T) & 3, which = i / (4 * T);
        const dwn_clip_src& s = which ? d.mix : d.src;
        float v = 0.f;
        if (which == 0 || d.mix.valid) {
            long long f = s.frame_start + (long long)t * s.frame_step;
            const float* base = c < 2 ? s.behavior : s.pupil_center;
            v = base[(long long)(c & 1) * s.length + f];
        }
        scal[i] = v;
    }
    __syncthreads();
    const long long plane = (long long)H * W;
    float* xb = x + (long long)b * 5 * T * plane + (long long)y * W;
    for (int i = threadIdx.x; i < T * W; i += 256) {
        int xo = i % W, t = i / W;
        int xx = xo - ws;
        const bool in_video = row_in_video && xx >= 0 && xx < W0;
        float v = in_video ? tile[t * tw + xx] : pad;
        if (blend) v = blend3(d.one_minus_lam, v, d.lam, in_video ? tile2[t * tw + xx] : pad);
        xb[(long long)t * plane + xo] = v;
    }
    for (int i = threadIdx.x; i < 4 * T * W; i += 256) {
        int xo = i % W, t = (i / W) % T, c = i / (W * T);
        const bool from_mix = row_in_box && xo >= d.bby1 && xo < d.bby2;
        float v = scal[((from_mix ? 4 : 0) + c) * T + t];
        if (blend) v = blend3(d.one_minus_lam, v, d.lam, scal[(4 + c) * T + t]);
        xb[((long long)(c + 1) * T + t) * plane + xo] = v;
    }
}

// grid (chunks of neurons, n_mice, B): the owning mouse's rows get relu(resp) (cut-mix: (1-lam)*relu(r1) + lam*relu(r2),
// three separately rounded operations like torch); every other mouse's rows of this sample are zero-filled, and the
// one-hot mice_weights row is written by the first chunk.
__global__ __launch_bounds__(256) void assemble_targets_kernel(const dwn_clip_desc* __restrict__ descs, int T,
                                                               float* const* __restrict__ targets,
                                                               const int* __restrict__ n_neurons, int n_mice,
                                                               float* __restrict__ mice_weights) {
    const int b = blockIdx.z, m = blockIdx.y;
    const dwn_clip_desc d = descs[b];
    const int N = n_neurons[m];
    if (blockIdx.x == 0 && m == 0)
        for (int i = threadIdx.x; i < n_mice; i += 256) mice_weights[(long long)b * n_mice + i] = (i == d.mouse) ? 1.f : 0.f;
    float* out = targets[m] + (long long)b * N * T;
    const long long total = (long long)N * T;
    const long long per_block = ((total + gridDim.x - 1) / gridDim.x + 255) / 256 * 256;
    const long long beg = (long long)blockIdx.x * per_block;
    long long end = beg + per_block;
    if (end > total) end = total;
    if (m != d.mouse) {
        for (long long i = beg + threadIdx.x; i < end; i += 256) out[i] = 0.f;
        return;
    }
    const bool mixed = d.mix.valid != 0;
    for (long long i = beg + threadIdx.x; i < end; i += 256) {
        int t = (int)(i % T);
        long long n = i / T;
        float v = fmaxf(d.src.responses[n * d.src.length + d.src.frame_start + (long long)t * d.src.frame_step], 0.f);
        if (mixed) {
            float v2 = fmaxf(d.mix.responses[n * d.mix.length + d.mix.frame_start + (long long)t * d.mix.frame_step], 0.f);
            v = blend3(d.one_minus_lam, v, d.lam, v2);
        }
        out[i] = v;
    }
}

}  // namespace

int k_assemble_inputs(const dwn_clip_desc* descs, int B, int T, int H0, int W0, int H, int W, float pad, float* x,
                      hipStream_t s) {
    size_t lds = (2 * (size_t)T * (W0 + 1) + 8 * (size_t)T) * sizeof(float);
    if (lds > 64 * 1024) return dwn_set_error(-3, "assemble_inputs: T*(W0+1) tile exceeds 64 KB of LDS");
    hipLaunchKernelGGL(assemble_inputs_kernel, dim3((unsigned)(B * H)), dim3(256), lds, s, descs, T, H0, W0, H, W, pad, x);
    DWN_CHECK_LAUNCH();
    return 0;
}

int k_assemble_targets(const dwn_clip_desc* descs, int B, int T, float* const* targets, const int* n_neurons,
                       int n_mice, int max_neurons, float* mice_weights, hipStream_t s) {
    long long total = (long long)max_neurons * T;
    int chunks = (int)((total + 256 * 16 - 1) / (256 * 16));
    if (chunks < 1) chunks = 1;
    if (chunks > 64) chunks = 64;
    hipLaunchKernelGGL(assemble_targets_kernel, dim3((unsigned)chunks, (unsigned)n_mice, (unsigned)B), dim3(256), 0, s,
                       descs, T, targets, n_neurons, n_mice, mice_weights);
    DWN_CHECK_LAUNCH();
    return 0;
}
