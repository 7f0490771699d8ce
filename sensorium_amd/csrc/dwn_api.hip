// C-ABI layer of libdwiseneuro_hip.so (declarations and contracts: include/dwn.h).
// Composites chain the fused kernels of one reference module; they enqueue on the caller's stream only,
// never allocate, never synchronise (hipGraph-capturable), and carve all scratch from the caller's workspace.
#include "dwn_internal.h"
#include "dwn_kernels.h"
#include <string.h>
#include <math.h>

static thread_local char g_err[512] = "";
#ifndef DWN_EVAL_CHAIN
#define DWN_EVAL_CHAIN 1      // eval forward: the chained stencil's rebuilt-input form where it is built (0: dwn_dwrc.hip everywhere; A/B builds)
#endif

// y1-recomputing spatial forward (dwn_dwrc.hip): used by the eval-mode block forward, where neither the BatchNorm-1
// statistics nor a saved y1 are needed, so conv_pw never runs as a pass of its own
extern "C" size_t dwn_dw_spatial_rc_blob_bytes(int E, int Cin);
extern "C" int dwn_dw_spatial_rc_supported(int dtype, int Cin, int E, int ks, int stride, int Hin, int Win);
extern "C" int dwn_dw_spatial_rc_prep(const float* w_pw, const float* w_dws, const float* bn1_coef, int E, int Cin, void* blob,
                                      int device, void* stream);
extern "C" int dwn_dw_spatial_fwd_rc(const dwn_dw_spatial_rc_fwd_args* a, int device, void* stream);
bool dw_spatial_bwd_rc_supported(const DwSpatialBwd& a, int dtype);          // dwn_dwbwd.hip
bool dw_spatial_fwd_rc_walk_supported(const DwSpatialFwd& a, int dtype);     // dwn_dwfwd.hip

int dwn_set_error(int code, const char* msg) {
    snprintf(g_err, sizeof(g_err), "dwn error %d: %s", code, msg ? msg : "");
    return code;
}

#define TRY(x) do { int rc__ = (x); if (rc__ != 0) return rc__; } while (0)

// ---- opt-in kernel-family timer (HIP events on the launch stream).  Off by default: zero cost and no state.
// bench.py turns it on for the kernel family it reports a roofline for (dwn_profile_enable / _collect).
#include <vector>
#include <dlfcn.h>
namespace {
struct ProfFam { std::vector<hipEvent_t> beg, end; size_t used = 0; };
static unsigned long long g_prof_mask = 0;
static ProfFam g_prof[DWN_FAM_COUNT];
constexpr size_t PROF_POOL = 8192;
// roctx ranges per kernel family: resolved at run time so that the library has no link-time dependency
static const char* const FAM_NAMES[DWN_FAM_COUNT] = {
    "dwn:pw_fwd", "dwn:dws_fwd", "dwn:dwt_fwd", "dwn:se_pool", "dwn:pwl_fwd", "dwn:resid_fwd", "dwn:resid_bwd", "dwn:pwl_dgrad",
    "dwn:pwl_wgrad", "dwn:bn3_reduce", "dwn:dwt_bwd", "dwn:dws_bwd", "dwn:pw_dgrad", "dwn:pw_wgrad", "dwn:cortex_fwd",
    "dwn:cortex_bwd", "dwn:readout_fwd", "dwn:readout_bwd"};
typedef int (*roctx_push_t)(const char*);
typedef int (*roctx_pop_t)(void);
static roctx_push_t g_roctx_push = nullptr;
static roctx_pop_t g_roctx_pop = nullptr;
static bool roctx_resolve() {
    static int state = 0;                         // 0 untried, 1 ok, -1 absent
    if (state == 0) {
        state = -1;
        for (const char* lib : {"librocprofiler-sdk-roctx.so", "libroctx64.so"}) {
            void* h = dlopen(lib, RTLD_NOW | RTLD_GLOBAL);
            if (!h) continue;
            g_roctx_push = (roctx_push_t)dlsym(h, "roctxRangePushA");
            g_roctx_pop = (roctx_pop_t)dlsym(h, "roctxRangePop");
            if (g_roctx_push && g_roctx_pop) { state = 1; break; }
        }
    }
    return state == 1;
}
struct ProfScope {
    int fam; hipStream_t s; bool on; bool range;
    ProfScope(int f, hipStream_t st) : fam(f), s(st), on(false), range(false) {
        if (g_prof_mask == 0) return;
        if ((g_prof_mask & DWN_PROF_ROCTX) && g_roctx_push) { g_roctx_push(FAM_NAMES[f]); range = true; }
        if (!((g_prof_mask >> f) & 1ull)) return;
        ProfFam& p = g_prof[f];
        if (p.used >= p.beg.size()) return;
        on = true;
        (void)hipEventRecord(p.beg[p.used], s);
    }
    ~ProfScope() {
        if (on) {
            ProfFam& p = g_prof[fam];
            (void)hipEventRecord(p.end[p.used], s);
            p.used++;
        }
        if (range) g_roctx_pop();
    }
};
}  // namespace
#define PROF(fam, call) do { ProfScope ps__((fam), s); TRY(call); } while (0)
#define HIP_TRY(x) do { hipError_t e__ = (x); if (e__ != hipSuccess) return dwn_set_error((int)e__, hipGetErrorString(e__)); } while (0)
#define ENTER(device) do { g_err[0] = 0; HIP_TRY(hipSetDevice(device)); } while (0)

namespace {

struct Carver {
    char* base; size_t off; size_t cap;
    explicit Carver(void* b, size_t c) : base((char*)b), off(0), cap(c) {}
    template <class U> U* take(size_t n) {
        off = (off + 255) & ~(size_t)255;
        U* p = base ? (U*)(base + off) : (U*)nullptr;
        off += n * sizeof(U);
        return p;
    }
    bool ok() const { return base == nullptr || off <= cap; }
};

inline size_t tsize(int dtype) { return dtype == DWN_BF16 ? 2 : 4; }
inline size_t nstat(int C) { return (size_t)DWN_NREP * 2 * C; }

LoadDesc ld_plain(const void* p, i64 ld) {
    LoadDesc d; memset(&d, 0, sizeof(d));
    d.p = p; d.ld = ld;
    return d;
}
LoadDesc ld_bnact(const void* p, i64 ld, const float* coef, int C, int act, const float* gate, int gate_ld,
                  int rows_per_sample) {
    LoadDesc d = ld_plain(p, ld);
    d.v1 = coef; d.v2 = coef + C; d.act = act; d.gate = gate; d.gate_ld = gate_ld;
    d.rows_per_sample = rows_per_sample > 0 ? rows_per_sample : 1;
    return d;
}
LoadDesc ld_affine2(const void* p, const void* q, i64 ld, const float* abc, int C) {
    LoadDesc d = ld_plain(p, ld);
    d.q = q; d.v1 = abc; d.v2 = abc + C; d.v3 = abc + 2 * C;
    return d;
}
LoadDesc ld_pe(const void* p, i64 ld, const float* pe_t, const float* pe_h, const float* pe_w, int T, int H, int W) {
    LoadDesc d = ld_plain(p, ld);
    d.pe_t = pe_t; d.pe_h = pe_h; d.pe_w = pe_w; d.pT = T; d.pH = H; d.pW = W; d.pe_ld = (int)ld;
    return d;
}
// p = raw y; v1..v4 = scale, shift, mean, invstd (used by the dw backward kernels)
LoadDesc ld_ycoef(const void* y, i64 ld, const float* coef, int C) {
    LoadDesc d = ld_plain(y, ld);
    d.v1 = coef; d.v2 = coef + C; d.v3 = coef + 2 * C; d.v4 = coef + 3 * C;
    return d;
}

int bn_finalize(const double* stats, int stat_c, double count, const dwn_bn& bn, int C, int training, float momentum,
                float eps, hipStream_t s) {
    if (training)
        return k_bn_finalize_train(stats, stat_c, count, bn.gamma, bn.beta, bn.running_mean, bn.running_var,
                                   bn.num_batches_tracked, momentum, eps, bn.coef, C, s);
    return k_bn_finalize_eval(bn.gamma, bn.beta, bn.running_mean, bn.running_var, eps, bn.coef, C, s);
}

BnFinJob fin_job(const double* stats, int stat_c, double count, const dwn_bn& bn, int C) {
    BnFinJob j; memset(&j, 0, sizeof(j));
    j.stats = stats; j.stat_c = stat_c; j.count = count; j.gamma = bn.gamma; j.beta = bn.beta;
    j.running_mean = bn.running_mean; j.running_var = bn.running_var; j.nbt = bn.num_batches_tracked; j.coef = bn.coef; j.C = C;
    return j;
}
BnBwdJob bwd_job(const double* stats, double count, const dwn_bn& bn, float* abc, int C) {
    BnBwdJob j; memset(&j, 0, sizeof(j));
    j.stats = stats; j.count = count; j.coef = bn.coef; j.dgamma = bn.dgamma; j.dbeta = bn.dbeta; j.abc = abc; j.C = C;
    return j;
}

GemmNN nn_base(const LoadDesc& a, int a_kind, const void* b, i64 ldb, void* c, i64 ldc, int M, int N, int K, int groups) {
    GemmNN g; memset(&g, 0, sizeof(g));
    g.a = a; g.a_kind = a_kind; g.b = b; g.ldb = ldb; g.c = c; g.ldc = ldc; g.M = M; g.N = N; g.K = K;
    g.groups = groups; g.epi = EPI_STORE;
    return g;
}
GemmTN tn_base(const LoadDesc& p, int pk, const LoadDesc& q, int qk, int M, int R, int Cc, float* dw, i64 lddw, int groups) {
    GemmTN g; memset(&g, 0, sizeof(g));
    g.p = p; g.p_kind = pk; g.q = q; g.q_kind = qk; g.M = M; g.R = R; g.Cc = Cc; g.dw = dw; g.lddw = lddw;
    g.groups = groups; g.nsplit = 0;
    return g;
}

// ---------------------------------------------------------------- block workspace layout
struct BlockWs {
    void *wpw, *wpwl;            // forward: W1 [Cmid][Cin], W2 [Cout][Cmid] in T; backward: W1^T [Cin][Cmid], W2^T [Cmid][Cout]
    float *wdws, *wdwt;          // tap-major depth-wise weights
    double *st1, *st2, *st3, *st4, *stsc;
    long long* pooled;           // forward: SE pooled sums (64-bit fixed point, pool_fix); backward: the same slot holds dg (float)
    float *abc1, *abc2, *abc3, *abc4, *abcsc, *ident3;
    float *dgp, *dhp, *dps;
    void* bp; float* r3; float* gacc;                      // conv_pw data-gradient folding (see dwn_elementwise.hip)
    float* tacc;                                           // conv_pw weight gradient: raw products T1 / Ga / s (k_pw_wgrad_fold)
    void* wgated;                                          // [B][Cout][Cmid] W2 . diag(gate_b) (forward only)
    void* rcblob;                                          // forward: slice images of the y1-recomputing stencil
    double* gram;                                          // y1-free training forward: [(Cin + 8)][Cin] raw products a0^T a0, 1^T a0 (fp64 atomics)
    float* pb;                                             // [B][Cout][Cmid] per-sample dy4^T z3 (backward, see pwl_bwd_per_sample)
    char* zero_beg; char* zero_end;
    size_t bytes;
};
// conv_pwl forward with the SE gate folded into per-sample weights (plain A loader): needs whole tiles per sample and
// the k-loop GEMM variant
// ... and enough rows per sample to pay for writing B weight copies: with 18432 rows (blocks 0-3) the copies are 1.8 MB and the
// plain-loader GEMM on its LDS-DMA ring wins; at 4608 / 1280 rows (blocks 4-8) the copies are 7-29 MB per block — written, then
// re-read at 1.35-1.53x the activations' bytes — and the gate in the A loader is faster for the step although the GEMM itself is
// slower (round 5, same box: pwl_fwd 1.085 -> 1.289 ms, step 24.27 -> 24.12 ms)
#ifndef PWL_GATED_MIN_ROWS
#define PWL_GATED_MIN_ROWS 8192
#endif
static bool pwl_gated_weights(const dwn_block_args& a) {
    const int bk = a.dtype == DWN_BF16 ? 64 : 32;
    // (bf16 only: fp32 keeps the per-sample weights everywhere — its eval-mode split products were validated in that form)
    return ((a.T * a.Hout * a.Wout) % 128) == 0 && a.Cmid > bk && (a.dtype != DWN_BF16 || a.T * a.Hout * a.Wout >= PWL_GATED_MIN_ROWS);
}
// conv_pwl backward through per-sample products (k_pwl_bwd_reduce) + the recompute-du GEMM epilogue (EPI_DH3): saves
// three passes over a [Mout][Cmid] tensor at the price of zeroing / accumulating / reading B [Cout][Cmid] fp32 matrices,
// so it is used when those are small next to one such pass.  dwn_block_args.pwl_bwd = 1 | 2 forces a path (the parity
// tests run both).
static bool pwl_bwd_per_sample(const dwn_block_args& a) {
#ifdef DWN_DETERMINISTIC
    return true;                 // the other path's GEMM epilogue adds to dg from several waves of a workgroup, inside its tile loop
#endif
    if (a.pwl_bwd == 2) return false;
    if (a.pwl_bwd == 1) return true;
    const double pb = 4.0 * a.B * a.Cout * a.Cmid * sizeof(float);
    const double pass = (double)a.B * a.T * a.Hout * a.Wout * a.Cmid * tsize(a.dtype);
    return pb <= pass;
}
// Training WITHOUT a materialised y1 (round 5; dwn_block_args.y1_mode).  y1 = a0 . W1^T is the widest tensor of a block and a
// Cin-deep product of one seven times narrower: the forward stencil rebuilds its rows on the matrix cores (dwn_dwfwd.hip, CIN > 0) with
// BatchNorm-1's batch statistics taken from the Gram matrix of a0 (k_bn1_gram_finalize), the backward stencil rebuilds the rows it
// needs the same way (dwn_dwbwd.hip, CIN > 0), and conv_pw's backward has not read y1 since round 4 — so y1 is neither written nor
// read: three E-wide passes over M_in rows less per block.  Both directions decide with this one predicate.
static bool block_y1_free(const dwn_block_args& a) {
    if (!a.training || a.dtype != DWN_BF16 || a.y1_mode == 1) return false;
    DwSpatialFwd f; memset(&f, 0, sizeof(f));
    f.planes = a.B * a.T; f.Hin = a.Hin; f.Win = a.Win; f.Hout = a.Hout; f.Wout = a.Wout; f.C = a.Cmid; f.stride = a.stride; f.ks = a.ks;
    f.in.ld = a.Cmid; f.a0_ld = a.Cin; f.Cin = a.Cin;
    if (!dw_spatial_fwd_rc_walk_supported(f, a.dtype)) return false;
    DwSpatialBwd d; memset(&d, 0, sizeof(d));
    d.planes = a.B * a.T; d.Hin = a.Hin; d.Win = a.Win; d.Hout = a.Hout; d.Wout = a.Wout; d.C = a.Cmid; d.stride = a.stride; d.ks = a.ks;
    d.dy.ld = a.Cmid; d.y1.ld = a.Cmid; d.a0_ld = a.Cin; d.Cin = a.Cin;
    if (!dw_spatial_bwd_rc_supported(d, a.dtype)) return false;
    // y1_mode 0: where it is also the faster path — 64 input channels (blocks 0-3: -0.4 ms per step); with 128 (blocks 4-6: four k-steps
    // per MFMA tile, 256 B of a0 per pixel against 128 B of a y1 slice) the rebuilding stencils lose more than conv_pw costs
    // (stand-alone 508 vs 295 us forward, 614 vs 549 us backward on block 4's shape); y1_mode 2 takes every block that is built
    return a.y1_mode == 2 || a.Cin == 64;
}
// eval-mode forward without conv_pw as its own pass (BatchNorm-1 is known: the tile-resident stencil of dwn_dwrc.hip)
static bool block_fwd_rc(const dwn_block_args& a) {
    return !a.training && dwn_dw_spatial_rc_supported(a.dtype, a.Cin, a.Cmid, a.ks, a.stride, a.Hin, a.Win) != 0;
}
// floats of the conv_pw data-gradient folding scratch: G accumulator [Cin][Cin] and r3 [Cin]
static size_t pw_fold_floats(int Cin) { return (size_t)Cin * Cin + Cin; }
BlockWs carve_block(const dwn_block_args& a, int backward, void* base, size_t cap) {
    BlockWs w; memset(&w, 0, sizeof(w));
    Carver c(base, cap);
    const size_t ts = tsize(a.dtype);
    w.wpw = c.take<char>((size_t)a.Cmid * a.Cin * ts);
    w.wpwl = c.take<char>((size_t)a.Cout * a.Cmid * ts);
    w.wdws = c.take<float>((size_t)a.ks * a.ks * a.Cmid);
    w.wdwt = c.take<float>((size_t)a.kt * a.Cmid);
    if (!backward && pwl_gated_weights(a)) w.wgated = c.take<char>((size_t)a.B * a.Cout * a.Cmid * ts);
    if (!backward && block_fwd_rc(a)) w.rcblob = c.take<char>(dwn_dw_spatial_rc_blob_bytes(a.Cmid, a.Cin));      // eval only
    c.take<char>(0);
    size_t z0 = (c.off + 255) & ~(size_t)255;
    w.st1 = c.take<double>(nstat(a.Cmid));
    w.st2 = c.take<double>(nstat(a.Cmid));
    w.st3 = c.take<double>(nstat(a.Cmid));
    w.st4 = c.take<double>(nstat(a.Cout));
    w.stsc = c.take<double>(nstat(backward ? a.Cout : a.Cin));
    w.pooled = c.take<long long>((size_t)a.B * a.Cmid);
    if (!backward && block_y1_free(a)) w.gram = c.take<double>((size_t)(a.Cin + 8) * a.Cin);
    size_t z1 = c.off;
    if (backward) {
        w.abc1 = c.take<float>(3 * (size_t)a.Cmid);
        w.abc2 = c.take<float>(3 * (size_t)a.Cmid);
        w.abc3 = c.take<float>(3 * (size_t)a.Cmid);
        w.abc4 = c.take<float>(3 * (size_t)a.Cout);
        w.abcsc = c.take<float>(3 * (size_t)a.Cout);
        w.ident3 = c.take<float>(3 * (size_t)a.Cmid);
        w.dgp = c.take<float>((size_t)a.B * a.Cmid);
        w.dhp = c.take<float>((size_t)a.B * a.se_r);
        w.dps = c.take<float>((size_t)a.B * a.Cmid);
        w.bp = c.take<char>((size_t)a.Cin * (a.Cmid + a.Cin) * ts);
        w.gacc = c.take<float>(pw_fold_floats(a.Cin));                  // one zeroed range: G accumulator, r3
        w.r3 = w.gacc + (size_t)a.Cin * a.Cin;
        w.tacc = c.take<float>(pw_wgrad_tacc_floats(a.Cmid, a.Cin));
        if (pwl_bwd_per_sample(a)) w.pb = c.take<float>((size_t)a.B * a.Cout * a.Cmid);
    }
    w.bytes = c.off + 256;
    if (base) { w.zero_beg = (char*)base + z0; w.zero_end = (char*)base + z1; }
    return w;
}

// dwn.h DWN_F32_*: does this fp32 GEMM run as three bf16 products?
static inline int f32_split_of(int policy, bool eval_forward) {
    return policy == DWN_F32_SPLIT3 ? 1 : policy == DWN_F32_NATIVE ? 0 : (eval_forward ? 1 : 0);
}

ResGeom geom_of(const dwn_block_args& a) {
    ResGeom gm;
    gm.BT = a.B * a.T; gm.T = a.T; gm.Hin = a.Hin; gm.Win = a.Win; gm.Hout = a.Hout; gm.Wout = a.Wout;
    gm.Cin = a.Cin; gm.Cout = a.Cout; gm.hsrc = a.hsrc; gm.wsrc = a.wsrc; gm.hinv = a.hinv; gm.winv = a.winv;
    return gm;
}

int check_block(const dwn_block_args& a) {
    if (a.Cin % 8 || a.Cmid % 8 || a.Cout % 8) return dwn_set_error(-2, "block: channel counts must be multiples of 8");
    if (a.ks != 3) return dwn_set_error(-4, "block: spatial_kernel must be 3");
    if (a.kt != 3 && a.kt != 5) return dwn_set_error(-4, "block: temporal_kernel must be 3 or 5");
    if (a.Hout != (a.Hin - 1) / a.stride + 1 || a.Wout != (a.Win - 1) / a.stride + 1)
        return dwn_set_error(-2, "block: Hout/Wout inconsistent with stride");
    return 0;
}

}  // namespace

extern "C" {

int dwn_abi_version(void) { return DWN_ABI_VERSION; }
int dwn_sizeof(const char* name) {
#define SZ(T) if (strcmp(name, #T) == 0) return (int)sizeof(T)
    SZ(dwn_load_desc); SZ(dwn_gemm_nn_args); SZ(dwn_gemm_tn_args); SZ(dwn_dw_spatial_fwd_args);
    SZ(dwn_dw_spatial_bwd_args); SZ(dwn_dw_temporal_fwd_args); SZ(dwn_dw_temporal_bwd_args); SZ(dwn_bn);
    SZ(dwn_stem_args); SZ(dwn_block_args); SZ(dwn_pool_args); SZ(dwn_cortex_args); SZ(dwn_readout_args);
    SZ(dwn_tensor_entry); SZ(dwn_clip_src); SZ(dwn_clip_desc); SZ(dwn_pw_bwd_args); SZ(dwn_dw_spatial_rc_fwd_args);
#undef SZ
    return -1;
}
const char* dwn_last_error(void) { return g_err; }

int dwn_profile_enable(unsigned long long family_mask, int device) {
    ENTER(device);
    for (int f = 0; f < DWN_FAM_COUNT; ++f) {
        ProfFam& p = g_prof[f];
        p.used = 0;
        if (((family_mask >> f) & 1ull) && p.beg.empty()) {
            p.beg.resize(PROF_POOL); p.end.resize(PROF_POOL);
            for (size_t i = 0; i < PROF_POOL; ++i) { HIP_TRY(hipEventCreate(&p.beg[i])); HIP_TRY(hipEventCreate(&p.end[i])); }
        }
    }
    if (family_mask & DWN_PROF_ROCTX) (void)roctx_resolve();      // absent library: ranges are skipped, events still work
    g_prof_mask = family_mask;
    return 0;
}
int dwn_profile_collect(int family, double* total_ms, long long* launches) {
    if (family < 0 || family >= DWN_FAM_COUNT) return dwn_set_error(-2, "profile_collect: bad family");
    ProfFam& p = g_prof[family];
    double tot = 0;
    for (size_t i = 0; i < p.used; ++i) {
        HIP_TRY(hipEventSynchronize(p.end[i]));
        float ms = 0;
        HIP_TRY(hipEventElapsedTime(&ms, p.beg[i], p.end[i]));
        tot += ms;
    }
    *total_ms = tot; *launches = (long long)p.used;
    p.used = 0;
    return 0;
}

int dwn_gemm_nn(const dwn_gemm_nn_args* a, int dtype, int device, void* stream) {
    ENTER(device);
    return launch_gemm_nn(*a, dtype, (hipStream_t)stream);
}
int dwn_gemm_tn(const dwn_gemm_tn_args* a, int dtype, int device, void* stream) {
    ENTER(device);
    return launch_gemm_tn(*a, dtype, (hipStream_t)stream);
}
int dwn_dw_spatial_fwd(const dwn_dw_spatial_fwd_args* a, int dtype, int device, void* stream) {
    ENTER(device);
    return launch_dw_spatial_fwd(*a, dtype, (hipStream_t)stream);
}
int dwn_dw_spatial_bwd(const dwn_dw_spatial_bwd_args* a, int dtype, int device, void* stream) {
    ENTER(device);
    return launch_dw_spatial_bwd(*a, dtype, (hipStream_t)stream);
}
int dwn_dw_spatial_bwd_rc_supported(const dwn_dw_spatial_bwd_args* a, int dtype) { return dw_spatial_bwd_rc_supported(*a, dtype) ? 1 : 0; }
int dwn_dw_spatial_fwd_rc_supported(const dwn_dw_spatial_fwd_args* a, int dtype) { return dw_spatial_fwd_rc_walk_supported(*a, dtype) ? 1 : 0; }
int dwn_dw_temporal_fwd(const dwn_dw_temporal_fwd_args* a, int dtype, int device, void* stream) {
    ENTER(device);
    return launch_dw_temporal_fwd(*a, dtype, (hipStream_t)stream);
}
int dwn_dw_temporal_bwd(const dwn_dw_temporal_bwd_args* a, int dtype, int device, void* stream) {
    ENTER(device);
    return launch_dw_temporal_bwd(*a, dtype, (hipStream_t)stream);
}
int dwn_bn_finalize(const double* stats, int stat_c, double count, const dwn_bn* bn, int C, int training,
                    float momentum, float eps, int device, void* stream) {
    ENTER(device);
    return bn_finalize(stats, stat_c, count, *bn, C, training, momentum, eps, (hipStream_t)stream);
}
int dwn_bn_bwd_finalize(const double* stats, double count, const dwn_bn* bn, float* abc, int C, int device,
                        void* stream) {
    ENTER(device);
    return k_bn_bwd_finalize(stats, count, bn->coef, bn->dgamma, bn->dbeta, abc, C, (hipStream_t)stream);
}
// BatchNorm-1 of conv_pw WITHOUT conv_pw's output (dwn.h): raw products [a0 | 1]^T a0 by one gemm_tn pass, then k_bn1_gram_finalize
size_t dwn_conv_pw_bn_stats_workspace_bytes(int Cin) { return (size_t)(Cin + 8) * Cin * sizeof(double) + 256; }
int dwn_conv_pw_bn_stats(const void* a0, long long a0_ld, long long M, const float* w_pw, int E, int Cin, const dwn_bn* bn,
                         float momentum, float eps, double* sc_stats, void* ws, size_t ws_bytes, int dtype, int device, void* stream) {
    ENTER(device);
    hipStream_t s = (hipStream_t)stream;
    if (!a0 || !w_pw || !bn || !bn->coef || !ws) return dwn_set_error(-1, "conv_pw_bn_stats: null pointer");
    if (Cin % 8 || E <= 0 || M <= 0 || M >= (1ll << 31)) return dwn_set_error(-2, "conv_pw_bn_stats: Cin % 8 == 0, 0 < M < 2^31");
    if (ws_bytes < dwn_conv_pw_bn_stats_workspace_bytes(Cin)) return dwn_set_error(-6, "conv_pw_bn_stats: workspace too small");
    double* gram = reinterpret_cast<double*>(((size_t)ws + 255) & ~(size_t)255);
    TRY(k_zero(gram, (size_t)(Cin + 8) * Cin * sizeof(double), s));
    LoadDesc cat = ld_plain(a0, a0_ld);
    cat.cat_c1 = Cin; cat.cat_c2 = 0;
    GemmTN g = tn_base(cat, LD_CAT1, ld_plain(a0, a0_ld), LD_PLAIN, (int)M, Cin + 8, Cin, reinterpret_cast<float*>(gram), Cin, 1);
    g.dw_f64 = 1;
    TRY(launch_gemm_tn(g, dtype, s));
    return k_bn1_gram_finalize(gram, w_pw, E, Cin, (double)M, bn->gamma, bn->beta, bn->running_mean, bn->running_var,
                               bn->num_batches_tracked, momentum, eps, bn->coef, sc_stats, dtype, s);
}
int dwn_pack_weight(const float* src, void* dst, int groups, int R, int C, int transpose, int Rd, int Cd, int dtype,
                    int device, void* stream) {
    ENTER(device);
    return k_pack_weight(src, dst, groups, R, C, transpose, Rd, Cd, dtype, (hipStream_t)stream);
}

// ------------------------------------------------------------------------------------------------ stem
size_t dwn_stem_workspace_bytes(const dwn_stem_args* a) {
    return ((size_t)DWN_NREP * stem_moment_count() * sizeof(double) + 256 +
            (size_t)DWN_NREP * a->C0 * stem_acc_stride() * sizeof(double) + 1024);
}
// Conv3d(Cin -> C0, 1x1x1) + BatchNorm3d (dwiseneuro.py:306-309).  y0 = W0 x is linear in the Cin <= 8 input channels, so the
// batch statistics come from the input moments (xmom) and y0 is never written: a->y0 is ignored (kept in the struct for
// layout compatibility), a->xmom receives [8] sums + [8][8] second moments (doubles) that the backward reads.
int dwn_stem_forward(const dwn_stem_args* a, int device, void* stream) {
    ENTER(device);
    hipStream_t s = (hipStream_t)stream;
    if (a->C0 % 8) return dwn_set_error(-2, "stem: C0 must be a multiple of 8");
    if (a->pe_t && (i64)a->T * a->H * a->W != a->S) return dwn_set_error(-2, "stem: T*H*W != S");
    Carver c(a->ws, a->ws_bytes);
    double* mom = c.take<double>((size_t)DWN_NREP * stem_moment_count());
    if (!c.ok()) return dwn_set_error(-6, "stem: workspace too small");
    const i64 M = (i64)a->B * a->S;
    if (a->training) {
        if (!a->xmom) return dwn_set_error(-1, "stem_forward: xmom buffer required in training mode");
        TRY(k_zero(mom, (size_t)DWN_NREP * stem_moment_count() * sizeof(double), s));
        TRY(k_stem_xmom(a->x, a->B, a->Cin, a->S, mom, s));
        TRY(k_stem_bn_finalize(mom, (double)M, a->w, a->bn.gamma, a->bn.beta, a->bn.running_mean, a->bn.running_var,
                               a->bn.num_batches_tracked, a->momentum, a->eps, a->bn.coef, a->xmom, a->C0, a->Cin, s));
    } else {
        TRY(bn_finalize(nullptr, a->C0, (double)M, a->bn, a->C0, 0, a->momentum, a->eps, s));
    }
    return k_stem_out(a->x, a->w, a->bn.coef, a->pe_t, a->pe_h, a->pe_w, a->T, a->H, a->W, a->B, a->Cin, a->S, a->C0, a->out,
                      a->dtype, s);
}
int dwn_stem_backward(const dwn_stem_args* a, int device, void* stream) {
    ENTER(device);
    hipStream_t s = (hipStream_t)stream;
    if (!a->xmom) return dwn_set_error(-1, "stem_backward: xmom (saved by the forward) required");
    Carver c(a->ws, a->ws_bytes);
    (void)c.take<double>((size_t)DWN_NREP * stem_moment_count());
    double* acc = c.take<double>((size_t)DWN_NREP * a->C0 * stem_acc_stride());
    if (!c.ok()) return dwn_set_error(-6, "stem: workspace too small");
    const i64 M = (i64)a->B * a->S;
    TRY(k_zero(acc, (size_t)DWN_NREP * a->C0 * stem_acc_stride() * sizeof(double), s));
    TRY(k_stem_bwd_acc(a->dout, a->x, a->xmom, (double)M, a->B, a->Cin, a->S, a->C0, acc, a->dtype, s));
    return k_stem_bwd_finalize(acc, a->xmom, a->w, a->bn.coef, (double)M, a->bn.dgamma, a->bn.dbeta, a->dw, a->C0, a->Cin, s);
}

// ------------------------------------------------------------------------------------------------ block
size_t dwn_block_workspace_bytes(const dwn_block_args* a, int backward) {
    return carve_block(*a, backward, nullptr, 0).bytes;
}

int dwn_block_forward(const dwn_block_args* ap, int device, void* stream) {
    ENTER(device);
    const dwn_block_args& a = *ap;
    hipStream_t s = (hipStream_t)stream;
    TRY(check_block(a));
    BlockWs w = carve_block(a, 0, a.ws, a.ws_bytes);
    if (w.bytes > a.ws_bytes) return dwn_set_error(-6, "block_forward: workspace too small");
    const int dt = a.dtype, tr = a.training;
    const i64 Min = (i64)a.B * a.T * a.Hin * a.Win, Mout = (i64)a.B * a.T * a.Hout * a.Wout;
    const int S_out = a.T * a.Hout * a.Wout;
    {   // one launch: zero the statistics arena, pack the four weights
        PrepArgs pa;
        bool ok = pa.zero(w.zero_beg, ((size_t)(w.zero_end - w.zero_beg) + 15) & ~(size_t)15);
        ok = ok && pa.packw(a.w_pw, w.wpw, 1, a.Cmid, a.Cin, 0, a.Cmid, a.Cin);
        ok = ok && pa.packw(a.w_pwl, w.wpwl, 1, a.Cout, a.Cmid, 0, a.Cout, a.Cmid);
        ok = ok && pa.packdw(a.w_dws, w.wdws, a.Cmid, a.ks * a.ks);
        ok = ok && pa.packdw(a.w_dwt, w.wdwt, a.Cmid, a.kt);
        if (!tr) {      // eval: the five BatchNorm coefficient sets come from running statistics — same launch
            ok = ok && pa.bneval(a.bn1, a.Cmid, a.eps) && pa.bneval(a.bn2, a.Cmid, a.eps) && pa.bneval(a.bn3, a.Cmid, a.eps);
            ok = ok && pa.bneval(a.bn4, a.Cout, a.eps) && pa.bneval(a.bnsc, a.Cout, a.eps);
        }
        if (!ok) return dwn_set_error(-2, "block_forward: workspace arena must be 16-byte aligned");
        TRY(k_prep(pa, dt, s));
    }

    // PositionalEncoding3d (dwiseneuro.py:184-192): normally already folded into x by the producer (stem /
    // previous block's residual kernel); stand-alone callers get it materialised here
    const void* a0 = a.x;
    if (!a.x_has_pe) {
        if (!a.a0) return dwn_set_error(-2, "block_forward: a0 buffer required when x_has_pe == 0");
        TRY(k_ew_apply(ld_pe(a.x, a.Cin, a.pe_t, a.pe_h, a.pe_w, a.T, a.Hin, a.Win), LD_PE, a.a0, a.Cin, Min, a.Cin, dt, s));
        a0 = a.a0;
    }
    // conv_pw (dwiseneuro.py:90-93): y1 = a0 @ W1^T, Σ/Σ² for bn1
    LoadDesc xin = ld_plain(a0, a.Cin);
    const bool identity_sc = a.stride == 1 && a.Hin == a.Hout && a.Win == a.Wout;
    const bool y1_free = tr && block_y1_free(a);
    // eval, 64 or 128 input channels, row-walk plane widths: the chained stencil's rebuilt-input form is also the faster way to skip
    // conv_pw (623 vs 768 us on block 0's shape, 276 vs 308 us on blocks 1-3'; faster than the tile-resident kernel at 128 channels
    // too: profiles/r5_predict_bf16_kernel_stats.csv); the tile-resident kernel of dwn_dwrc.hip keeps the other geometries
    bool eval_chain = false;
    if (DWN_EVAL_CHAIN && !tr && dt == DWN_BF16) {
        DwSpatialFwd f; memset(&f, 0, sizeof(f));
        f.planes = a.B * a.T; f.Hin = a.Hin; f.Win = a.Win; f.Hout = a.Hout; f.Wout = a.Wout; f.C = a.Cmid; f.stride = a.stride; f.ks = a.ks;
        f.in.ld = a.Cmid; f.a0_ld = a.Cin; f.Cin = a.Cin;
        eval_chain = dw_spatial_fwd_rc_walk_supported(f, dt);
    }
    if (eval_chain) {
        DwSpatialFwd d; memset(&d, 0, sizeof(d));
        d.in = ld_bnact(nullptr, a.Cmid, a.bn1.coef, a.Cmid, 1, nullptr, 0, 1);
        d.a0 = a0; d.a0_ld = a.Cin; d.w1 = w.wpw; d.Cin = a.Cin;
        d.w = w.wdws; d.out = a.y2; d.planes = a.B * a.T; d.Hin = a.Hin; d.Win = a.Win; d.Hout = a.Hout;
        d.Wout = a.Wout; d.C = a.Cmid; d.stride = a.stride; d.ks = a.ks; d.stats = nullptr;
        PROF(DWN_FAM_DWS_FWD, launch_dw_spatial_fwd(d, dt, s));
    } else if (y1_free) {
        // y1-free training: BatchNorm-1's batch statistics from the Gram matrix of a0 (one C-wide pass: gemm_tn [a0 | 1]^T a0), then the
        // chained stencil rebuilds the y1 rows it needs from a0 on the matrix cores: conv_pw is no pass, a.y1 is not written
        LoadDesc cat = ld_plain(a0, a.Cin);
        cat.cat_c1 = a.Cin; cat.cat_c2 = 0;
        GemmTN g = tn_base(cat, LD_CAT1, ld_plain(a0, a.Cin), LD_PLAIN, (int)Min, a.Cin + 8, a.Cin, reinterpret_cast<float*>(w.gram), a.Cin, 1);
        g.dw_f64 = 1;                                 // fp64 atomics: the variance is a difference of these sums
        PROF(DWN_FAM_PW_FWD, launch_gemm_tn(g, dt, s));
        PROF(DWN_FAM_PW_FWD, k_bn1_gram_finalize(w.gram, a.w_pw, a.Cmid, a.Cin, (double)Min, a.bn1.gamma, a.bn1.beta, a.bn1.running_mean,
                                                 a.bn1.running_var, a.bn1.num_batches_tracked, a.momentum, a.eps, a.bn1.coef,
                                                 identity_sc ? w.stsc : nullptr, dt, s));
        DwSpatialFwd d; memset(&d, 0, sizeof(d));
        d.in = ld_bnact(nullptr, a.Cmid, a.bn1.coef, a.Cmid, 1, nullptr, 0, 1);
        d.a0 = a0; d.a0_ld = a.Cin; d.w1 = w.wpw; d.Cin = a.Cin;
        d.w = w.wdws; d.out = a.y2; d.planes = a.B * a.T; d.Hin = a.Hin; d.Win = a.Win; d.Hout = a.Hout;
        d.Wout = a.Wout; d.C = a.Cmid; d.stride = a.stride; d.ks = a.ks; d.stats = w.st2;
        PROF(DWN_FAM_DWS_FWD, launch_dw_spatial_fwd(d, dt, s));
    } else if (block_fwd_rc(a)) {
        // eval: BatchNorm-1 needs no batch statistics and nobody reads y1 again, so the stencil kernel rebuilds its y1
        // tiles from a0 (MFMA) and conv_pw disappears as a pass (a.y1 is not written)
        TRY(dwn_dw_spatial_rc_prep(a.w_pw, w.wdws, a.bn1.coef, a.Cmid, a.Cin, w.rcblob, device, stream));
        dwn_dw_spatial_rc_fwd_args r; memset(&r, 0, sizeof(r));
        r.a0 = a0; r.a0_ld = a.Cin; r.blob = w.rcblob; r.out = a.y2; r.planes = a.B * a.T; r.Hin = a.Hin; r.Win = a.Win;
        r.Hout = a.Hout; r.Wout = a.Wout; r.Cin = a.Cin; r.E = a.Cmid; r.stride = a.stride; r.stats = nullptr;
        r.rows_band = 0; r.round_y1 = 1;
        PROF(DWN_FAM_DWS_FWD, dwn_dw_spatial_fwd_rc(&r, device, stream));
    } else {
    {
        GemmNN g = nn_base(xin, LD_PLAIN, w.wpw, a.Cin, a.y1, a.Cmid, (int)Min, a.Cmid, a.Cin, 1);
        g.f32_split = f32_split_of(a.f32_products, !tr);      // eval-mode fp32: bf16 hi/lo products (dwn_gemm.hip NN_F32_X3) unless NATIVE
        g.stats = tr ? w.st1 : nullptr; g.stat_nchan = a.Cmid;
        PROF(DWN_FAM_PW_FWD, launch_gemm_nn(g, dt, s));
    }
    if (tr) TRY(bn_finalize(w.st1, a.Cmid, (double)Min, a.bn1, a.Cmid, tr, a.momentum, a.eps, s));
    // spat_covn_dw (:96-102)
    {
        DwSpatialFwd d; memset(&d, 0, sizeof(d));
        d.in = ld_bnact(a.y1, a.Cmid, a.bn1.coef, a.Cmid, 1, nullptr, 0, 1);
        d.w = w.wdws; d.out = a.y2; d.planes = a.B * a.T; d.Hin = a.Hin; d.Win = a.Win; d.Hout = a.Hout;
        d.Wout = a.Wout; d.C = a.Cmid; d.stride = a.stride; d.ks = a.ks; d.stats = tr ? w.st2 : nullptr;
        PROF(DWN_FAM_DWS_FWD, launch_dw_spatial_fwd(d, dt, s));
    }
    }
    if (tr) TRY(bn_finalize(w.st2, a.Cmid, (double)Mout, a.bn2, a.Cmid, tr, a.momentum, a.eps, s));
    // temp_covn_dw (:105-111)
    const bool eval_z3 = !tr;       // eval: z3 and the SE pooling sums come straight from the temporal pass (integer sums: any order)
    {
        DwTemporalFwd d; memset(&d, 0, sizeof(d));
        d.in = ld_bnact(a.y2, a.Cmid, a.bn2.coef, a.Cmid, 1, nullptr, 0, 1);
        d.w = w.wdwt; d.out = a.y3; d.B = a.B; d.T = a.T; d.HW = a.Hout * a.Wout; d.C = a.Cmid; d.kt = a.kt;
        d.stats = tr ? w.st3 : nullptr;
        if (eval_z3) {       // eval: BatchNorm-3 is known -> z3 and the SE pooling sums straight from this pass, y3 never stored
            d.out = a.z3; d.z_scale = a.bn3.coef; d.z_shift = a.bn3.coef + a.Cmid; d.pooled = w.pooled;
        }
        PROF(DWN_FAM_DWT_FWD, launch_dw_temporal_fwd(d, dt, s));
    }
    if (tr) TRY(bn_finalize(w.st3, a.Cmid, (double)Mout, a.bn3, a.Cmid, tr, a.momentum, a.eps, s));
    // se (:38-43)
    if (!eval_z3) {
        LoadDesc z3 = ld_bnact(a.y3, a.Cmid, a.bn3.coef, a.Cmid, 1, nullptr, 0, S_out);
        PROF(DWN_FAM_SE_POOL, k_se_pool(z3, a.B, a.Cmid, S_out, w.pooled, a.z3, dt, s));
    }
    // (z3 * gate_b) @ W2^T == z3 @ (W2 . diag(gate_b))^T: where conv_pwl runs on per-sample weights the SE kernel writes them too
    int gate_folded = 0;
    TRY(k_se_mlp_fwd(w.pooled, 1.0f / (float)S_out, a.se_wr, a.se_br, a.se_we, a.se_be, a.B, a.Cmid, a.se_r,
                     a.se_pmean, a.se_hidpre, a.se_gate, pwl_gated_weights(a) ? a.w_pwl : nullptr, w.wgated, a.Cout, dt, &gate_folded, s));
    // conv_pwl (:117-120): y4 = (silu(bn3(y3)) * gate) @ W2^T
    if (pwl_gated_weights(a)) {
        if (!gate_folded) TRY(k_gate_weights(a.w_pwl, a.se_gate, w.wgated, a.B, a.Cout, a.Cmid, dt, s));
        GemmNN g = nn_base(ld_plain(a.z3, a.Cmid), LD_PLAIN, w.wgated, a.Cmid, a.y4, a.Cout, (int)Mout, a.Cout, a.Cmid, 1);
        g.b_sample_stride = (i64)a.Cout * a.Cmid; g.b_rows_per_sample = S_out;
        g.stats = tr ? w.st4 : nullptr; g.stat_nchan = a.Cout; g.f32_split = f32_split_of(a.f32_products, !tr);
        PROF(DWN_FAM_PWL_FWD, launch_gemm_nn(g, dt, s));
    } else {
        LoadDesc u = ld_plain(a.z3, a.Cmid);
        u.gate = a.se_gate; u.gate_ld = a.Cmid; u.rows_per_sample = S_out;
        GemmNN g = nn_base(u, LD_GATE, w.wpwl, a.Cmid, a.y4, a.Cout, (int)Mout, a.Cout, a.Cmid, 1);
        g.stats = tr ? w.st4 : nullptr; g.stat_nchan = a.Cout; g.f32_split = f32_split_of(a.f32_products, !tr);
        PROF(DWN_FAM_PWL_FWD, launch_gemm_nn(g, dt, s));
    }
    // shortcut (:125-134) + residual (:143); the two linear BatchNorms (conv_pwl.1.bn, bn_sc.bn) finalise in one launch
    ResGeom gm = geom_of(a);
    if (tr) {
        // (y1-free on an identity-map block: the shortcut's sums came out of the Gram pass, k_bn1_gram_finalize)
        if (!(y1_free && identity_sc)) PROF(DWN_FAM_RESID_FWD, k_shortcut_stats(xin, gm, w.stsc, dt, s));
        TRY(k_bn_finalize_train2(fin_job(w.st4, a.Cout, (double)Mout, a.bn4, a.Cout),
                                 fin_job(w.stsc, a.Cin, (double)Mout, a.bnsc, a.Cout), a.momentum, a.eps, s));
    }
    PROF(DWN_FAM_RESID_FWD, k_residual_fwd(xin, a.y4, a.bn4.coef, a.bnsc.coef, a.drop_scale, gm, a.out_pe_t, a.out_pe_h,
                                           a.out_pe_w, a.out, dt, s));
    return 0;
}

// conv_pw backward from (dh1, a0, W1, BatchNorm-1 backward coefficients): da0 and dW1, y1 not read (see the comment at its call
// in dwn_block_backward).  gacc (pw_fold_floats) and tacc (pw_wgrad_tacc_floats) must be zero on entry.
// res / res_abc / res_C (optional, dwn.h dwn_pw_bwd_args): the stride-1 shortcut branch's gradient folded in — the A2 * a0 and A3
// terms into G / r3, the A1 * dout term as the data-gradient GEMM's residual epilogue — so that da0 is the block's input gradient.
// rg (optional, with res): the shortcut's nearest map when it is not the identity (stride-2 block) — hinv / winv and the two
// plane sizes; the x terms then stay in the kernel's epilogue (only sampled rows have them).
static int pw_backward(int dt, const void* dh1, const void* a0, const float* w_pw, const float* abc, int E, int Cin, i64 M,
                       void* bp, float* gacc, float* r3, float* tacc, void* da0, float* dw, const void* res,
                       const float* res_abc, int res_C, const ResGeom* rg, hipStream_t s) {
    const int res_n = res ? res_C / Cin : 0;
    const bool gathered = res && rg && rg->hinv;
    if (res && (!res_abc || res_C % Cin || res_n < 1 || res_n > 2))
        return dwn_set_error(-2, "pw_backward: the shortcut term needs res_abc and res_C in {Cin, 2 Cin}");
    if (res && !pw_bwd_fused_supported(dt, M, E, Cin))
        return dwn_set_error(-3, "pw_backward: the shortcut term is built into the one-pass kernel only (dwn_pw_bwd_fused_supported)");
    TRY(k_pw_bwd_prep(w_pw, abc, E, Cin, bp, gacc, r3, dt, (res && !gathered) ? res_abc : nullptr, res_C, s));
    if (pw_bwd_fused_supported(dt, M, E, Cin)) {
        // 64-channel blocks: both products from ONE pass over dh1
        PROF(DWN_FAM_PW_DGRAD, launch_pw_bwd_fused(dh1, a0, bp, r3, da0, tacc, M, E, Cin, dt, res, res_abc, res_n,
                                                   gathered ? rg->hinv : nullptr, gathered ? rg->winv : nullptr, gathered ? rg->Hin : 0,
                                                   gathered ? rg->Win : 0, gathered ? rg->Hout : 0, gathered ? rg->Wout : 0, s));
    } else {
        {
            GemmNN g = nn_base(ld_plain(dh1, E), LD_PLAIN, bp, (i64)E + Cin, da0, Cin, (int)M, Cin, E + Cin, 1);
            g.epi = EPI_STORE_CAT; g.a2 = a0; g.a2_ld = Cin; g.K1 = E; g.bias = r3;
            PROF(DWN_FAM_PW_DGRAD, launch_gemm_nn(g, dt, s));
        }
        {   // raw products [dh1 | a0 | 1]^T a0 -> tacc
            LoadDesc cat = ld_plain(dh1, E);
            cat.q = a0; cat.ld2 = Cin; cat.cat_c1 = E; cat.cat_c2 = Cin;
            GemmTN g = tn_base(cat, LD_CAT1, ld_plain(a0, Cin), LD_PLAIN, (int)M, E + Cin + 8, Cin, tacc, Cin, 1);
            PROF(DWN_FAM_PW_WGRAD, launch_gemm_tn(g, dt, s));
        }
    }
    return k_pw_wgrad_fold(tacc, abc, w_pw, E, Cin, dw, dt, s);
}

int dwn_block_backward(const dwn_block_args* ap, int device, void* stream) {
    ENTER(device);
    const dwn_block_args& a = *ap;
    hipStream_t s = (hipStream_t)stream;
    TRY(check_block(a));
    if (!a.training) return dwn_set_error(-7, "block_backward: only training-mode (batch-statistics) backward is built");
    BlockWs w = carve_block(a, 1, a.ws, a.ws_bytes);
    if (w.bytes > a.ws_bytes) return dwn_set_error(-6, "block_backward: workspace too small");
    const int dt = a.dtype;
    const i64 Min = (i64)a.B * a.T * a.Hin * a.Win, Mout = (i64)a.B * a.T * a.Hout * a.Wout;
    const int S_out = a.T * a.Hout * a.Wout;
    float* dg = reinterpret_cast<float*>(w.pooled);
    const bool y1_free = block_y1_free(a);           // the forward of these arguments wrote no y1 (dwn_block_forward_writes)
    {   // one launch: zero the statistics arena, W2^T [Cmid][Cout], tap-major depth-wise weights, identity affine
        PrepArgs pa;
        bool ok = pa.zero(w.zero_beg, ((size_t)(w.zero_end - w.zero_beg) + 15) & ~(size_t)15);
        ok = ok && pa.packw(a.w_pwl, w.wpwl, 1, a.Cout, a.Cmid, 1, a.Cmid, a.Cout);
        if (y1_free) ok = ok && pa.packw(a.w_pw, w.wpw, 1, a.Cmid, a.Cin, 0, a.Cmid, a.Cin);      // W1 as rounded: the stencil rebuilds y1 with it
        ok = ok && pa.packdw(a.w_dws, w.wdws, a.Cmid, a.ks * a.ks);
        ok = ok && pa.packdw(a.w_dwt, w.wdwt, a.Cmid, a.kt);
        ok = ok && pa.fill(w.ident3, 1.0f, a.Cmid);
        ok = ok && pa.fill(w.ident3 + a.Cmid, 0.0f, 2 * a.Cmid);
        // the atomically accumulated weight gradients are cleared here (callers pass uninitialised buffers); dw_pw is written
        // whole by k_pw_wgrad_fold from the raw products accumulated in tacc
        ok = ok && pa.zero(a.dw_dws, (size_t)a.Cmid * a.ks * a.ks * sizeof(float));
        ok = ok && pa.zero(a.dw_dwt, (size_t)a.Cmid * a.kt * sizeof(float));
        ok = ok && pa.zero(a.dw_pwl, (size_t)a.Cout * a.Cmid * sizeof(float));
        if (w.pb) ok = ok && pa.zero(w.pb, (size_t)a.B * a.Cout * a.Cmid * sizeof(float));
        ok = ok && pa.zero(w.gacc, pw_fold_floats(a.Cin) * sizeof(float));
        ok = ok && pa.zero(w.tacc, pw_wgrad_tacc_floats(a.Cmid, a.Cin) * sizeof(float));
        if (!ok) return dwn_set_error(-2, "block_backward: workspace arena and dw_* buffers must be 16-byte aligned");
        TRY(k_prep(pa, dt, s));
    }

    LoadDesc xin = ld_plain(a.x_has_pe ? a.x : a.a0, a.Cin);     // block input including its positional encoding
    ResGeom gm = geom_of(a);
    // residual + the two linear BNs (bn4 = conv_pwl.1.bn, bnsc = bn_sc.bn)
    PROF(DWN_FAM_RESID_BWD, k_residual_bwd_reduce(xin, a.y4, a.dout, a.bn4.coef, a.bnsc.coef, a.drop_scale, gm, w.st4, w.stsc, dt, s));
    TRY(k_bn_bwd_finalize2(bwd_job(w.st4, (double)Mout, a.bn4, w.abc4, a.Cout),
                           bwd_job(w.stsc, (double)Mout, a.bnsc, w.abcsc, a.Cout), s));
    PROF(DWN_FAM_RESID_BWD, k_residual_bwd_dy4(a.y4, a.dout, w.abc4, a.drop_scale, gm, a.dy4, dt, s));
    // conv_pwl backward: du = dy4 @ W2 (+ SE gate gradient), dW2 = dy4^T @ u
    void* du = a.buf_a;
    if (w.pb) {
        // per-sample products P_b = dy4_b^T z3_b -> dW2 and the SE gate gradient without touching du
        {
            GemmTN g = tn_base(ld_plain(a.dy4, a.Cout), LD_PLAIN, ld_plain(a.z3, a.Cmid), LD_PLAIN, (int)Mout, a.Cout,
                               a.Cmid, w.pb, a.Cmid, 1);
            g.rows_per_sample = S_out; g.dw_sample_stride = (i64)a.Cout * a.Cmid;
            PROF(DWN_FAM_PWL_WGRAD, launch_gemm_tn(g, dt, s));
        }
        PROF(DWN_FAM_PWL_WGRAD, k_pwl_bwd_reduce(w.pb, a.se_gate, a.w_pwl, a.B, a.Cout, a.Cmid, a.dw_pwl, dg, s));
        TRY(k_se_mlp_bwd(dg, a.se_gate, a.se_hidpre, a.se_pmean, a.se_wr, a.se_we, a.B, a.Cmid, a.se_r,
                         1.0f / (float)S_out, w.dgp, w.dhp, w.dps, a.dse_wr, a.dse_br, a.dse_we, a.dse_be, s));
        // dh3 = (du*gate + dpS) * silu'(bn3(y3)) with du = dy4 . W2 recomputed in the GEMM, plus the bn3 backward sums
        GemmNN g = nn_base(ld_plain(a.dy4, a.Cout), LD_PLAIN, w.wpwl, a.Cout, du, a.Cmid, (int)Mout, a.Cmid, a.Cout, 1);
        g.epi = EPI_DH3; g.y3 = a.y3; g.ldy3 = a.Cmid; g.gate3 = a.se_gate; g.dps3 = w.dps; g.dg_ld = a.Cmid;
        g.coef3 = a.bn3.coef; g.coef3_ld = a.Cmid; g.rows_per_sample = S_out;
        g.stats = w.st3; g.stat_nchan = a.Cmid;
        PROF(DWN_FAM_PWL_DGRAD, launch_gemm_nn(g, dt, s));
    } else {
    {
        GemmNN g = nn_base(ld_plain(a.dy4, a.Cout), LD_PLAIN, w.wpwl, a.Cout, du, a.Cmid, (int)Mout, a.Cmid, a.Cout, 1);
        g.epi = EPI_DG; g.y3 = a.z3; g.ldy3 = a.Cmid; g.s3 = nullptr; g.t3 = nullptr; g.dg = dg;
        g.dg_ld = a.Cmid; g.rows_per_sample = S_out;
        PROF(DWN_FAM_PWL_DGRAD, launch_gemm_nn(g, dt, s));
    }
    {
        LoadDesc u = ld_plain(a.z3, a.Cmid);
        u.gate = a.se_gate; u.gate_ld = a.Cmid; u.rows_per_sample = S_out;
        GemmTN g = tn_base(ld_plain(a.dy4, a.Cout), LD_PLAIN, u, LD_GATE, (int)Mout, a.Cout, a.Cmid, a.dw_pwl, a.Cmid, 1);
        PROF(DWN_FAM_PWL_WGRAD, launch_gemm_tn(g, dt, s));
    }
    // SE backward
    TRY(k_se_mlp_bwd(dg, a.se_gate, a.se_hidpre, a.se_pmean, a.se_wr, a.se_we, a.B, a.Cmid, a.se_r,
                     1.0f / (float)S_out, w.dgp, w.dhp, w.dps, a.dse_wr, a.dse_br, a.dse_we, a.dse_be, s));
    // bn3 backward sums over dh3 = (du*gate + dpS) * silu'(h3)
    LoadDesc d3; memset(&d3, 0, sizeof(d3));
    d3.p = du; d3.q = a.y3; d3.ld = a.Cmid; d3.v1 = w.ident3; d3.v2 = w.ident3 + a.Cmid; d3.v3 = w.ident3 + 2 * a.Cmid;
    d3.v4 = a.bn3.coef; d3.v5 = a.bn3.coef + a.Cmid; d3.gate = a.se_gate; d3.gate2 = w.dps; d3.gate_ld = a.Cmid;
    d3.rows_per_sample = S_out;
    // ... and dh3 replaces du in place, so the temporal kernel reads (dh3, y3) with the plain BN-backward affine
    PROF(DWN_FAM_BN3_REDUCE, k_bn3_bwd_reduce(d3, a.bn3.coef, Mout, a.Cmid, w.st3, du, dt, s));
    }
    TRY(k_bn_bwd_finalize(w.st3, (double)Mout, a.bn3.coef, a.bn3.dgamma, a.bn3.dbeta, w.abc3, a.Cmid, s));
    // temporal dw backward
    {
        DwTemporalBwd d; memset(&d, 0, sizeof(d));
        d.dy = ld_affine2(du, a.y3, a.Cmid, w.abc3, a.Cmid);
        d.dy_kind = LD_PLAIN;                           // y3 is recomputed from y2 inside the kernel (one E-wide pass less)
        d.y2 = ld_ycoef(a.y2, a.Cmid, a.bn2.coef, a.Cmid);
        d.w = w.wdwt; d.dh2 = a.buf_b; d.dw = a.dw_dwt; d.B = a.B; d.T = a.T; d.HW = a.Hout * a.Wout; d.C = a.Cmid;
        d.kt = a.kt; d.stats = w.st2;
        PROF(DWN_FAM_DWT_BWD, launch_dw_temporal_bwd(d, dt, s));
    }
    TRY(k_bn_bwd_finalize(w.st2, (double)Mout, a.bn2.coef, a.bn2.dgamma, a.bn2.dbeta, w.abc2, a.Cmid, s));
    // spatial dw backward (du is dead: reuse buf_a for dh1)
    void* dh1 = a.buf_a;
    {
        DwSpatialBwd d; memset(&d, 0, sizeof(d));
        d.dy = ld_affine2(a.buf_b, a.y2, a.Cmid, w.abc2, a.Cmid);
        d.y1 = ld_ycoef(y1_free ? nullptr : a.y1, a.Cmid, a.bn1.coef, a.Cmid);
        if (y1_free) { d.a0 = xin.p; d.a0_ld = a.Cin; d.w1 = w.wpw; d.Cin = a.Cin; }      // y1 rebuilt from the block input
        d.w = w.wdws; d.dh1 = dh1; d.dw = a.dw_dws; d.planes = a.B * a.T; d.Hin = a.Hin; d.Win = a.Win;
        d.Hout = a.Hout; d.Wout = a.Wout; d.C = a.Cmid; d.stride = a.stride; d.ks = a.ks; d.stats = w.st1;
        PROF(DWN_FAM_DWS_BWD, launch_dw_spatial_bwd(d, dt, s));
    }
    TRY(k_bn_bwd_finalize(w.st1, (double)Min, a.bn1.coef, a.bn1.dgamma, a.bn1.dbeta, w.abc1, a.Cmid, s));
    // conv_pw backward WITHOUT y1.  dy1 = A1*dh1 + A2*y1 + A3 is linear and y1 = a0.W1^T, so the y1 terms fold into Cin x Cin
    // matrices on either side:  da0 = [dh1 | a0] . [diag(A1) W1 ; G] + r3  (Bp, r3: k_pw_bwd_prep) and
    // dW1 = diag(A1) (dh1^T a0) + diag(A2) W1 (a0^T a0) + A3 (1^T a0)  (raw products in tacc, folded by k_pw_wgrad_fold)
    // On a stride-1 block (identity shortcut map, output channels = input channels tiled once or twice) the shortcut branch's
    // gradient is folded into this GEMM and its result IS dx: no pass over (da0, x, dout) -> dx.
    // (one-pass kernel only: as an epilogue of the two-GEMM path's data-gradient GEMM it cost 30-60 us where the pass it replaces
    // takes 20-35.)  On a strided block the epilogue gathers: only the rows the nearest map samples carry the shortcut's terms.
    const bool identity_sc = a.stride == 1 && a.Hin == a.Hout && a.Win == a.Wout;
    const bool dx_folded = a.Cout % a.Cin == 0 && a.Cout / a.Cin <= 2 && dwn_pw_bwd_fused_supported(dt, Min, a.Cmid, a.Cin) &&
                           (identity_sc || (gm.hinv && gm.winv && a.Hin + a.Win <= 512));
    TRY(pw_backward(dt, dh1, xin.p, a.w_pw, w.abc1, a.Cmid, a.Cin, Min, w.bp, w.gacc, w.r3, w.tacc, dx_folded ? a.dx : a.da0, a.dw_pw,
                    dx_folded ? a.dout : nullptr, w.abcsc, a.Cout, (dx_folded && !identity_sc) ? &gm : nullptr, s));
    if (dx_folded) return 0;
    PROF(DWN_FAM_RESID_BWD, k_residual_bwd_dx(xin, a.da0, a.dout, w.abcsc, gm, a.dx, dt, s));
    return 0;
}

// Which of the intermediates dwn_block_forward will write for these arguments: bit 0 = y1, bit 1 = y3.  Training writes
// both; the eval-mode forward skips y1 where the stencil rebuilds it (block_fwd_rc) and y3 where the temporal pass emits z3
// directly — the caller need not allocate what is not written (and may pass NULL for it).
int dwn_block_forward_writes(const dwn_block_args* ap) {
    const dwn_block_args& a = *ap;
    if (a.training) return block_y1_free(a) ? 2 : 3;
    if (block_fwd_rc(a)) return 0;
    if (DWN_EVAL_CHAIN && a.dtype == DWN_BF16) {        // the chained stencil's rebuilt-input form (dwn_block_forward, eval_chain)
        DwSpatialFwd f; memset(&f, 0, sizeof(f));
        f.planes = a.B * a.T; f.Hin = a.Hin; f.Win = a.Win; f.Hout = a.Hout; f.Wout = a.Wout; f.C = a.Cmid; f.stride = a.stride; f.ks = a.ks;
        f.in.ld = a.Cmid; f.a0_ld = a.Cin; f.Cin = a.Cin;
        if (dw_spatial_fwd_rc_walk_supported(f, a.dtype)) return 0;
    }
    return 1;
}

// ------------------------------------------------------------------------------------------------ pool
int dwn_pool_forward(const dwn_pool_args* a, int device, void* stream) {
    ENTER(device);
    if (a->C % 8) return dwn_set_error(-2, "pool: C must be a multiple of 8");
    return k_pool_fwd(a->x, a->out, a->BT, a->HW, a->C, a->dtype, (hipStream_t)stream);
}
int dwn_pool_backward(const dwn_pool_args* a, int device, void* stream) {
    ENTER(device);
    return k_pool_bwd(a->dout, a->dx, a->BT, a->HW, a->C, a->dtype, (hipStream_t)stream);
}

// ------------------------------------------------------------------------------------------------ cortex
namespace {
struct CortexWs { void* wp; double *st, *stsc; float *abc, *abcsc; void *dy, *dxmain; char *zb, *ze; size_t bytes; };
CortexWs carve_cortex(const dwn_cortex_args& a, int backward, void* base, size_t cap) {
    CortexWs w; memset(&w, 0, sizeof(w));
    Carver c(base, cap);
    const size_t ts = tsize(a.dtype);
    const i64 M = (i64)a.B * a.T;
    w.wp = c.take<char>((size_t)a.C * (a.Cin / a.groups) * ts);
    c.take<char>(0);
    size_t z0 = (c.off + 255) & ~(size_t)255;
    w.st = c.take<double>(nstat(a.C));
    w.stsc = c.take<double>(nstat(backward ? a.C : a.Cin));
    size_t z1 = c.off;
    if (backward) {
        w.abc = c.take<float>(3 * (size_t)a.C);
        w.abcsc = c.take<float>(3 * (size_t)a.C);
        w.dy = c.take<char>((size_t)M * a.C * ts);
        w.dxmain = c.take<char>((size_t)M * a.Cin * ts);
    }
    w.bytes = c.off + 256;
    if (base) { w.zb = (char*)base + z0; w.ze = (char*)base + z1; }
    return w;
}
}  // namespace

size_t dwn_cortex_workspace_bytes(const dwn_cortex_args* a, int backward) {
    return carve_cortex(*a, backward, nullptr, 0).bytes;
}
int dwn_cortex_forward(const dwn_cortex_args* ap, int device, void* stream) {
    ENTER(device);
    const dwn_cortex_args& a = *ap;
    hipStream_t s = (hipStream_t)stream;
    if (a.Cin % (8 * a.groups) || a.C % (8 * a.groups)) return dwn_set_error(-2, "cortex: channels per group must be multiples of 8");
    CortexWs w = carve_cortex(a, 0, a.ws, a.ws_bytes);
    if (w.bytes > a.ws_bytes) return dwn_set_error(-6, "cortex_forward: workspace too small");
    const int M = a.B * a.T, Kg = a.Cin / a.groups, Ng = a.C / a.groups, dt = a.dtype, tr = a.training;
    {
        PrepArgs pa;
        bool ok = pa.zero(w.zb, ((size_t)(w.ze - w.zb) + 15) & ~(size_t)15);
        ok = ok && pa.packw(a.w, w.wp, 1, a.C, Kg, 0, a.C, Kg);
        if (!ok) return dwn_set_error(-2, "cortex_forward: workspace arena must be 16-byte aligned");
        TRY(k_prep(pa, dt, s));
    }
    GemmNN g = nn_base(ld_plain(a.x, a.Cin), LD_PLAIN, w.wp, Kg, a.y, a.C, M, Ng, Kg, a.groups);
    g.stats = tr ? w.st : nullptr; g.stat_nchan = a.C; g.f32_split = f32_split_of(a.f32_products, !tr);
    PROF(DWN_FAM_CORTEX_FWD, launch_gemm_nn(g, dt, s));
    if (tr) {
        TRY(k_colstats(ld_plain(a.x, a.Cin), LD_PLAIN, M, a.Cin, w.stsc, dt, s));
        TRY(k_bn_finalize_train2(fin_job(w.st, a.C, (double)M, a.bn, a.C), fin_job(w.stsc, a.Cin, (double)M, a.bnsc, a.C),
                                 a.momentum, a.eps, s));
    } else {
        TRY(bn_finalize(w.st, a.C, (double)M, a.bn, a.C, tr, a.momentum, a.eps, s));
        TRY(bn_finalize(w.stsc, a.Cin, (double)M, a.bnsc, a.C, tr, a.momentum, a.eps, s));
    }
    return k_cortex_residual_fwd(a.y, a.x, a.bn.coef, a.bnsc.coef, a.drop_scale, M, a.T, a.Cin, a.C, a.groups, a.out,
                                 dt, s);
}
int dwn_cortex_backward(const dwn_cortex_args* ap, int device, void* stream) {
    ENTER(device);
    const dwn_cortex_args& a = *ap;
    hipStream_t s = (hipStream_t)stream;
    if (!a.training) return dwn_set_error(-7, "cortex_backward: only training-mode backward is built");
    CortexWs w = carve_cortex(a, 1, a.ws, a.ws_bytes);
    if (w.bytes > a.ws_bytes) return dwn_set_error(-6, "cortex_backward: workspace too small");
    const int M = a.B * a.T, Kg = a.Cin / a.groups, Ng = a.C / a.groups, dt = a.dtype;
    {
        PrepArgs pa;
        bool ok = pa.zero(w.zb, ((size_t)(w.ze - w.zb) + 15) & ~(size_t)15);
        ok = ok && pa.packw(a.w, w.wp, a.groups, Ng, Kg, 1, Kg, Ng);       // per group W^T [Kg][Ng]
        // dw is accumulated with atomics by the weight-gradient product: cleared here, in the same launch (as dwn_block_backward
        // clears its dw_*), not by the caller
        ok = ok && pa.zero(a.dw, (size_t)a.C * Kg * sizeof(float));
        if (!ok) return dwn_set_error(-2, "cortex_backward: workspace arena and dw must be 16-byte aligned (C * Cin / groups a multiple of 4)");
        TRY(k_prep(pa, dt, s));
    }
    TRY(k_cortex_bwd_reduce(a.y, a.x, a.dout, a.dout_mask, a.dout_mask_ld, a.bn.coef, a.bnsc.coef, a.drop_scale, M, a.T,
                            a.Cin, a.C, a.groups, w.st, w.stsc, dt, s));
    TRY(k_bn_bwd_finalize2(bwd_job(w.st, (double)M, a.bn, w.abc, a.C), bwd_job(w.stsc, (double)M, a.bnsc, w.abcsc, a.C), s));
    TRY(k_cortex_bwd_dy(a.y, a.dout, a.dout_mask, a.dout_mask_ld, a.bn.coef, w.abc, a.drop_scale, M, a.T, a.C, a.groups,
                        w.dy, dt, s));
    {
        GemmNN g = nn_base(ld_plain(w.dy, a.C), LD_PLAIN, w.wp, Ng, w.dxmain, a.Cin, M, Kg, Ng, a.groups);
        PROF(DWN_FAM_CORTEX_BWD, launch_gemm_nn(g, dt, s));
    }
    {
        GemmTN g = tn_base(ld_plain(w.dy, a.C), LD_PLAIN, ld_plain(a.x, a.Cin), LD_PLAIN, M, Ng, Kg, a.dw, Kg, a.groups);
        PROF(DWN_FAM_CORTEX_BWD, launch_gemm_tn(g, dt, s));
    }
    return k_cortex_bwd_dx(w.dxmain, a.x, a.dout, a.dout_mask, a.dout_mask_ld, w.abcsc, M, a.T, a.Cin, a.C, a.dx, dt, s);
}

// ------------------------------------------------------------------------------------------------ readout
namespace {
struct ReadoutWs { void* wp; void* dz; void* xd; size_t bytes; int Npad, Rg, Rp, ldp, ldt; };
ReadoutWs carve_readout(const dwn_readout_args& a, int backward, void* base, size_t cap) {
    ReadoutWs w; memset(&w, 0, sizeof(w));
    Carver c(base, cap);
    const size_t ts = tsize(a.dtype);
    const i64 M = (i64)a.B * a.T;
    w.Npad = (a.n_out + a.groups - 1) / a.groups * a.groups;
    w.Rg = w.Npad / a.groups;
    w.Rp = (w.Rg + 63) / 64 * 64;      // the data-gradient product contracts over Rp: whole k-tiles (LDS-DMA variant)
    const int Kg = a.Cin / a.groups;
    w.ldp = Kg; w.ldt = w.Rp;          // (row padding of 8 / 64 / 72 elements measured: no effect, so none)
    w.wp = c.take<char>(backward ? (a.wt ? 0 : (size_t)a.groups * Kg * w.ldt * ts) : (size_t)w.Npad * w.ldp * ts);
    if (backward) w.dz = c.take<char>((size_t)M * a.groups * w.Rp * ts);
    if (a.drop_mask) w.xd = c.take<char>((size_t)M * a.Cin * ts);      // x * dropout mask, materialised once
    w.bytes = c.off + 256;
    return w;
}
}  // namespace

size_t dwn_readout_workspace_bytes(const dwn_readout_args* a, int backward) {
    return carve_readout(*a, backward, nullptr, 0).bytes;
}
size_t dwn_readout_wt_bytes(const dwn_readout_args* a) {
    ReadoutWs w = carve_readout(*a, 0, nullptr, 0);
    return (size_t)a->groups * (a->Cin / a->groups) * w.ldt * tsize(a->dtype);
}
int dwn_readout_forward(const dwn_readout_args* ap, int device, void* stream) {
    ENTER(device);
    const dwn_readout_args& a = *ap;
    hipStream_t s = (hipStream_t)stream;
    if (a.Cin % (8 * a.groups)) return dwn_set_error(-2, "readout: in-channels per group must be a multiple of 8");
    ReadoutWs w = carve_readout(a, 0, a.ws, a.ws_bytes);
    if (w.bytes > a.ws_bytes) return dwn_set_error(-6, "readout_forward: workspace too small");
    const int M = a.B * a.T, Kg = a.Cin / a.groups, dt = a.dtype;
    TRY(k_pack_weight_dual(a.w, w.wp, a.wt, a.groups, w.Rg, Kg, w.Rp, w.ldp, w.ldt, dt, s));   // both operand layouts, one read of the weight
    LoadDesc x = ld_plain(a.x, a.Cin);
    const int kind = LD_PLAIN;
    if (a.drop_mask) {
        // Dropout1d (dwiseneuro.py:275): the masked input (8 MB) is materialised once — applying the per-(sample,
        // channel) mask inside the GEMM's operand loader put a dependent mask load in front of every A chunk
        LoadDesc xm = x;
        xm.gate = a.drop_mask; xm.gate_ld = a.Cin; xm.rows_per_sample = a.T;
        TRY(k_ew_apply(xm, LD_GATE, w.xd, a.Cin, M, a.Cin, dt, s));
        x = ld_plain(w.xd, a.Cin);
    }
    GemmNN g = nn_base(x, kind, w.wp, w.ldp, nullptr, 0, M, w.Rg, Kg, a.groups);
    g.epi = EPI_READOUT; g.bias = a.bias; g.sp_beta = a.softplus_beta; g.out_nct = a.out; g.Tn = a.T; g.n_valid = a.n_out;
    g.f32_split = f32_split_of(a.f32_products, false);            // the caller says (it knows whether this is an inference forward)
    PROF(DWN_FAM_READOUT_FWD, launch_gemm_nn(g, dt, s));
    return 0;
}
int dwn_readout_backward(const dwn_readout_args* ap, int device, void* stream) {
    ENTER(device);
    const dwn_readout_args& a = *ap;
    hipStream_t s = (hipStream_t)stream;
    ReadoutWs w = carve_readout(a, 1, a.ws, a.ws_bytes);
    if (w.bytes > a.ws_bytes) return dwn_set_error(-6, "readout_backward: workspace too small");
    const int M = a.B * a.T, Kg = a.Cin / a.groups, dt = a.dtype;
    const void* wt = a.wt;                       // per group W^T [Kg][Rp], zero padded: kept from the forward, or packed now
    if (!wt) { TRY(k_pack_weight_dual(a.w, nullptr, w.wp, a.groups, w.Rg, Kg, w.Rp, w.ldp, w.ldt, dt, s)); wt = w.wp; }
    TRY(k_readout_dz(a.dout, a.out, a.softplus_beta, a.B, a.T, a.n_out, w.Rg, w.Rp, a.groups, w.dz, a.dbias, dt, s));
    LoadDesc dz = ld_plain(w.dz, (i64)a.groups * w.Rp);
    {
        GemmNN g = nn_base(dz, LD_PLAIN, wt, w.ldt, a.dx, a.Cin, M, Kg, w.Rp, a.groups);
        PROF(DWN_FAM_READOUT_BWD, launch_gemm_nn(g, dt, s));
    }
    LoadDesc x = ld_plain(a.x, a.Cin);
    const int kind = LD_PLAIN;
    if (a.drop_mask) {
        LoadDesc xm = x;
        xm.gate = a.drop_mask; xm.gate_ld = a.Cin; xm.rows_per_sample = a.T;
        TRY(k_ew_apply(xm, LD_GATE, w.xd, a.Cin, M, a.Cin, dt, s));      // masked input for the weight gradient
        x = ld_plain(w.xd, a.Cin);
        // grad wrt the un-dropped input: dx *= mask (in place)
        LoadDesc dxm = ld_plain(a.dx, a.Cin);
        dxm.gate = a.drop_mask; dxm.gate_ld = a.Cin; dxm.rows_per_sample = a.T;
        TRY(k_ew_apply(dxm, LD_GATE, a.dx, a.Cin, M, a.Cin, dt, s));
    }
    GemmTN g = tn_base(dz, LD_PLAIN, x, kind, M, w.Rg, Kg, a.dw, Kg, a.groups);
    g.R_load = w.Rp;
    // every element of dw [Npad][Kg] is produced here: written with plain stores when the tiles alone fill the chip (the real
    // readouts: 1984 tiles), zeroed + accumulated otherwise — the caller does not clear dw
    g.overwrite = ((size_t)a.dw & 15) == 0 && ((size_t)a.groups * w.Rg * Kg) % 4 == 0;
    if (!g.overwrite) TRY(k_fill_f32(a.dw, 0.f, a.groups * w.Rg * Kg, s));
    PROF(DWN_FAM_READOUT_BWD, launch_gemm_tn(g, dt, s));
    return 0;
}

// ------------------------------------------------------------------------------------------------ loss / optimizer
int dwn_poisson_loss_forward(const float* pred, const float* target, const float* w, long long per_sample,
                             long long total, float eps, double* loss_acc, int device, void* stream) {
    ENTER(device);
    return k_poisson_fwd(pred, target, w, per_sample, total, eps, loss_acc, (hipStream_t)stream);
}
int dwn_poisson_loss_backward(const float* pred, const float* target, const float* w, const float* gscale,
                              long long per_sample, long long total, float eps, float* dpred, int device,
                              void* stream) {
    ENTER(device);
    return k_poisson_bwd(pred, target, w, gscale, per_sample, total, eps, dpred, (hipStream_t)stream);
}
int dwn_f64_to_f32(const double* src, float* dst, int n, int device, void* stream) {
    ENTER(device);
    return k_f64_to_f32(src, dst, n, (hipStream_t)stream);
}
int dwn_adamw_ema_multi(const dwn_tensor_entry* list, int ntensors, int max_blocks, double lr, double beta1,
                        double beta2, double eps, double weight_decay, long long step, double ema_decay,
                        double grad_scale, int device, void* stream) {
    ENTER(device);
    if (step < 1) return dwn_set_error(-2, "adamw: step must be >= 1");
    const double bc1 = 1.0 - pow(beta1, (double)step);
    const double bc2 = 1.0 - pow(beta2, (double)step);
    return k_adamw_ema(list, ntensors, max_blocks, (float)(1.0 - lr * weight_decay), (float)(1.0 - beta1), (float)beta2,
                       (float)(1.0 - beta2), (float)eps, (float)(lr / bc1), (float)sqrt(bc2), (float)ema_decay,
                       (float)(1.0 - ema_decay), (float)grad_scale, (hipStream_t)stream);
}
int dwn_ema_lerp_multi(const dwn_tensor_entry* list, int ntensors, int max_blocks, double decay, int device,
                       void* stream) {
    ENTER(device);
    return k_ema_lerp(list, ntensors, max_blocks, (float)decay, (float)(1.0 - decay), (hipStream_t)stream);
}

int dwn_pw_bwd_fused_supported(int dtype, long long M, int E, int Cin) {
    return pw_bwd_fused_supported(dtype, M, E, Cin) ? 1 : 0;
}
size_t dwn_pw_backward_workspace_bytes(int E, int Cin, int dtype) {
    return (size_t)Cin * (E + Cin) * tsize(dtype) + 256 + (pw_fold_floats(Cin) + pw_wgrad_tacc_floats(E, Cin)) * sizeof(float) + 256;
}
int dwn_pw_backward(const dwn_pw_bwd_args* a, int dtype, int device, void* stream) {
    ENTER(device);
    if (!a || !a->dh1 || !a->a0 || !a->w_pw || !a->abc || !a->da0 || !a->dw || !a->ws)
        return dwn_set_error(-1, "pw_backward: null pointer");
    if (a->E <= 0 || a->Cin <= 0 || a->E % 8 || a->Cin % 8 || a->M <= 0 || a->M > 0x7fffffffLL)
        return dwn_set_error(-2, "pw_backward: E and Cin must be positive multiples of 8, 0 < M < 2^31");
    if (a->ws_bytes < dwn_pw_backward_workspace_bytes(a->E, a->Cin, dtype)) return dwn_set_error(-6, "pw_backward: workspace too small");
    hipStream_t s = (hipStream_t)stream;
    Carver c(a->ws, a->ws_bytes);
    void* bp = c.take<char>((size_t)a->Cin * (a->E + a->Cin) * tsize(dtype));
    const size_t nz = pw_fold_floats(a->Cin) + pw_wgrad_tacc_floats(a->E, a->Cin);
    float* gacc = c.take<float>(nz);
    float* r3 = gacc + (size_t)a->Cin * a->Cin;
    float* tacc = gacc + pw_fold_floats(a->Cin);
    TRY(k_zero(gacc, nz * sizeof(float), s));
    ResGeom rg; memset(&rg, 0, sizeof(rg));
    rg.hinv = a->res_hinv; rg.winv = a->res_winv; rg.Hin = a->res_Hin; rg.Win = a->res_Win; rg.Hout = a->res_Hout; rg.Wout = a->res_Wout;
    return pw_backward(dtype, a->dh1, a->a0, a->w_pw, a->abc, a->E, a->Cin, a->M, bp, gacc, r3, tacc, a->da0, a->dw, a->res,
                       a->res_abc, a->res_C, a->res_hinv ? &rg : nullptr, s);
}

int dwn_assemble_inputs(const dwn_clip_desc* descs, int B, int T, int H0, int W0, int H, int W, float pad_fill,
                        float* x, int device, void* stream) {
    ENTER(device);
    if (!descs || !x) return dwn_set_error(-1, "assemble_inputs: null pointer");
    if (B <= 0 || T <= 0 || H0 <= 0 || W0 <= 0) return dwn_set_error(-2, "assemble_inputs: B, T, H0, W0 must be positive");
    if (H < H0 || W < W0) return dwn_set_error(-2, "assemble_inputs: output frame smaller than the video");
    return k_assemble_inputs(descs, B, T, H0, W0, H, W, pad_fill, x, (hipStream_t)stream);
}

int dwn_assemble_targets(const dwn_clip_desc* descs, int B, int T, float* const* targets, const int* n_neurons,
                         int n_mice, int max_neurons, float* mice_weights, int device, void* stream) {
    ENTER(device);
    if (!descs || !targets || !n_neurons || !mice_weights) return dwn_set_error(-1, "assemble_targets: null pointer");
    if (B <= 0 || T <= 0 || n_mice <= 0 || max_neurons <= 0)
        return dwn_set_error(-2, "assemble_targets: B, T, n_mice, max_neurons must be positive");
    if (B > 65535 || n_mice > 65535) return dwn_set_error(-2, "assemble_targets: B and n_mice are grid dimensions (<= 65535)");
    return k_assemble_targets(descs, B, T, targets, n_neurons, n_mice, max_neurons, mice_weights, (hipStream_t)stream);
}

}  // extern "C"
