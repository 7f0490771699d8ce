"""GPU parity of the MFMA GEMM kernels (dwn_gemm_nn / dwn_gemm_tn) through the C-ABI.

Reference arithmetic: the 1x1x1 / grouped k=1 convolutions of src/models/dwiseneuro.py:91,118,207,276 are
row-major GEMMs; the oracle states them as ``x @ W.T`` (oracle.pointwise).  Tolerances: fp32 path 1e-3
(north star), bf16 storage path 4e-2 relative L2 (inputs are rounded to bf16 on both sides, so the
observed error is ~1e-3; the bound is the separately stated bf16 tolerance of SURVEY.md §7g).
"""
import ctypes as C

import pytest
import torch

pytestmark = pytest.mark.gpu

from tests.gpu_helpers import dev, load_desc, read_stats, rel, stats_buffer, stream, tol  # noqa: E402


@pytest.fixture(scope="module")
def L():
    import sensorium_amd._lib as lib
    return lib


def _dt(L, dtype):
    return L.DWN_BF16 if dtype == torch.bfloat16 else L.DWN_F32


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("M,N,K", [(128, 64, 64), (300, 448, 64), (257, 24, 8), (1000, 128, 448), (64, 256, 1792),
                                   (513, 896, 128), (1300, 136, 72), (640, 56, 128), (2000, 448, 40)])
def test_gemm_nn_plain_with_stats(L, dtype, M, N, K):
    torch.manual_seed(M + N + K)
    a = torch.randn(M, K, device=dev()).to(dtype)
    b = (torch.randn(N, K, device=dev()) / K ** 0.5).to(dtype)
    # asymmetric integer-valued check as well (catches swapped row/col maps that random data would hide by tolerance)
    c = torch.empty(M, N, dtype=dtype, device=dev())
    st = stats_buffer(N)
    g = L.GemmNNArgs()
    g.a = load_desc(L, a, K)
    g.a_kind = L.LD_PLAIN
    g.b = b.data_ptr(); g.ldb = K; g.c = c.data_ptr(); g.ldc = N
    g.M, g.N, g.K, g.groups = M, N, K, 1
    g.stats = st.data_ptr(); g.stat_nchan = N; g.epi = L.EPI_STORE
    L.check(L.lib.dwn_gemm_nn(C.byref(g), _dt(L, dtype), 0, stream()), "gemm_nn")
    torch.cuda.synchronize()
    ref = a.double() @ b.double().t()
    assert rel(c, ref) < tol(dtype, 1e-5, 6e-3)
    s0, s1 = read_stats(st, N)
    cf = c.double()
    assert rel(s0, cf.sum(0)) < 1e-4 and rel(s1, (cf * cf).sum(0)) < 1e-4


@pytest.mark.parametrize("bn", [128, 256])
@pytest.mark.parametrize("M,N,K,groups", [(256, 256, 64, 1), (300, 264, 72, 1), (1000, 128, 448, 1), (777, 1000, 200, 1), (257, 24, 8, 1),
                                          (200, 48, 32, 2), (900, 1000, 328, 2), (2100, 1792, 256, 1)])
def test_gemm_nn_xl_kernel(L, M, N, K, groups, bn):
    """The 256-row LDS-DMA kernel (dwn_gemm_xl.hip: conv_pw of the 256-channel blocks, src/models/dwiseneuro.py:91) forced onto
    ragged shapes: M / N / K tails, groups, both tile widths, with the BatchNorm sums."""
    torch.manual_seed(M + N + K)
    dtype = torch.bfloat16
    a = torch.randn(M, groups * K, device=dev()).to(dtype)
    b = (torch.randn(groups * N, K, device=dev()) / K ** 0.5).to(dtype)
    c = torch.full((M, groups * N), float("nan"), dtype=dtype, device=dev())
    st = stats_buffer(groups * N)
    g = L.GemmNNArgs()
    g.a = load_desc(L, a, groups * K)
    g.a_kind = L.LD_PLAIN
    g.b = b.data_ptr(); g.ldb = K; g.c = c.data_ptr(); g.ldc = groups * N
    g.M, g.N, g.K, g.groups = M, N, K, groups
    g.stats = st.data_ptr(); g.stat_nchan = groups * N; g.epi = L.EPI_STORE
    g.variant = L.NN_XL256 if bn == 256 else L.NN_XL128
    L.check(L.lib.dwn_gemm_nn(C.byref(g), _dt(L, dtype), 0, stream()), "gemm_nn")
    torch.cuda.synchronize()
    ref = torch.cat([a[:, i * K:(i + 1) * K].double() @ b[i * N:(i + 1) * N].double().t() for i in range(groups)], 1)
    assert not torch.isnan(c.float()).any()
    assert rel(c, ref) < 6e-3
    s0, s1 = read_stats(st, groups * N)
    cf = c.double()
    assert rel(s0, cf.sum(0)) < 1e-4 and rel(s1, (cf * cf).sum(0)) < 1e-4


@pytest.mark.parametrize("form", ["plain", "gate", "cat"])
@pytest.mark.parametrize("M,N,K,K2", [(128, 128, 128, 128), (1000, 256, 384, 128), (2560, 128, 896, 128), (5120, 256, 1792, 256),
                                      (40960, 256, 1792, 256)])
def test_gemm_nn_a_direct_kernel(L, form, M, N, K, K2):
    """The A-direct kernel (dwn_gemm_kd.hip: conv_pwl with the SE gate on its input, src/models/dwiseneuro.py:117-120, and
    conv_pw's K-concatenated data gradient, :90-91 backward, of the 128- / 256-channel blocks) forced onto small and ragged-M
    shapes and at block 8's shape: equal to the 128-row kernel BIT FOR BIT (same k order, same roundings), BatchNorm sums
    included, and within the bf16 bound of float64."""
    torch.manual_seed(M + N + K)
    dtype = torch.bfloat16
    rows = 256 if M % 256 == 0 else 128 if M % 128 == 0 else 0        # rows per sample (the gate form needs whole tiles per sample)
    if form == "gate" and not rows:
        pytest.skip("gate form: rows per sample must be a multiple of 128")
    a = torch.randn(M, K, device=dev()).to(dtype)
    Kt = K + (K2 if form == "cat" else 0)
    b = (torch.randn(N, Kt, device=dev()) / Kt ** 0.5).to(dtype)
    a2 = torch.randn(M, K2, device=dev()).to(dtype)
    bias = torch.randn(N, device=dev())
    gate = torch.rand(M // rows if rows else 1, K, device=dev()) + 0.25
    out = {}
    for variant in (L.NN_KD, L.NN_TILE128):
        c = torch.full((M, N), float("nan"), dtype=dtype, device=dev())
        st = stats_buffer(N)
        g = L.GemmNNArgs()
        g.a = load_desc(L, a, K)
        g.a_kind = L.LD_PLAIN
        if form == "gate":
            g.a.gate = gate.data_ptr(); g.a.gate_ld = K; g.a.rows_per_sample = rows
            g.a_kind = L.LD_GATE
        g.b = b.data_ptr(); g.ldb = Kt; g.c = c.data_ptr(); g.ldc = N
        g.M, g.N, g.K, g.groups = M, N, Kt, 1
        g.stats = st.data_ptr(); g.stat_nchan = N; g.epi = L.EPI_STORE
        if form == "cat":
            g.epi = L.EPI_STORE_CAT; g.a2 = a2.data_ptr(); g.a2_ld = K2; g.K1 = K; g.bias = bias.data_ptr()
        g.variant = variant
        L.check(L.lib.dwn_gemm_nn(C.byref(g), _dt(L, dtype), 0, stream()), "gemm_nn")
        torch.cuda.synchronize()
        out[variant] = (c, read_stats(st, N))
    c, (s0, s1) = out[L.NN_KD]
    c_ref, (r0, r1) = out[L.NN_TILE128]
    assert not torch.isnan(c.float()).any()
    assert torch.equal(c.view(torch.int16), c_ref.view(torch.int16))
    af = a.double()
    if form == "gate":
        af = (a.float() * gate.repeat_interleave(rows, 0)).to(dtype).double()      # rounded as the loader rounds
    ref = af @ b[:, :K].double().t()
    if form == "cat":
        ref = ref + a2.double() @ b[:, K:].double().t() + bias.double()
    assert rel(c, ref) < 6e-3
    cf = c.double()
    assert rel(s0, cf.sum(0)) < 1e-4 and rel(s1, (cf * cf).sum(0)) < 1e-4
    assert rel(s0, r0) < 1e-5 and rel(s1, r1) < 1e-5


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_gemm_nn_exact_small_integers(L, dtype):
    """A = I-like / asymmetric B with small integers: exact in both dtypes, catches any transposed tile map."""
    M, N, K = 256, 128, 64
    a = torch.zeros(M, K, device=dev())
    a[torch.arange(M), torch.arange(M) % K] = 1.0
    a[:, 0] += (torch.arange(M, device=dev()) % 3).float()
    b = ((torch.arange(N, device=dev())[:, None] * 3 + torch.arange(K, device=dev())[None, :]) % 7 - 3).float()
    c = torch.empty(M, N, dtype=dtype, device=dev())
    g = L.GemmNNArgs()
    a_t, b_t = a.to(dtype), b.to(dtype)
    g.a = load_desc(L, a_t, K); g.a_kind = L.LD_PLAIN
    g.b = b_t.data_ptr(); g.ldb = K; g.c = c.data_ptr(); g.ldc = N
    g.M, g.N, g.K, g.groups = M, N, K, 1
    L.check(L.lib.dwn_gemm_nn(C.byref(g), _dt(L, dtype), 0, stream()), "gemm_nn")
    torch.cuda.synchronize()
    assert torch.equal(c.float(), a @ b.t())


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_gemm_nn_loaders(L, dtype):
    """PE, BN+SiLU+gate and BN-backward-affine prologues against plain torch math."""
    torch.manual_seed(3)
    B, T, H, W, K, N = 2, 3, 5, 6, 24, 40
    M = B * T * H * W
    x = torch.randn(M, K, device=dev()).to(dtype)
    w = (torch.randn(N, K, device=dev()) / K ** 0.5).to(dtype)
    pt, ph, pw = (torch.randn(s, K, device=dev()) for s in (T, H, W))
    out = torch.empty(M, N, dtype=dtype, device=dev())

    def run(desc, kind):
        g = L.GemmNNArgs()
        g.a = desc; g.a_kind = kind
        g.b = w.data_ptr(); g.ldb = K; g.c = out.data_ptr(); g.ldc = N
        g.M, g.N, g.K, g.groups = M, N, K, 1
        L.check(L.lib.dwn_gemm_nn(C.byref(g), _dt(L, dtype), 0, stream()), "gemm_nn")
        torch.cuda.synchronize()
        return out.clone()

    def rt(v):   # operand rounding of the MFMA input
        return v.to(dtype).double()

    # PE
    got = run(load_desc(L, x, K, pe_t=pt, pe_h=ph, pe_w=pw, pT=T, pH=H, pW=W, pe_ld=K), L.LD_PE)
    enc = (pt[:, None, None, :] + ph[None, :, None, :] + pw[None, None, :, :]).expand(B, T, H, W, K).reshape(M, K)
    ref = rt(x.float() + enc) @ w.double().t()
    assert rel(got, ref) < tol(dtype, 1e-5, 6e-3)
    # BN + SiLU + per-sample gate
    sc, sh = torch.rand(K, device=dev()) + 0.5, torch.randn(K, device=dev())
    gate = torch.rand(B, K, device=dev())
    got = run(load_desc(L, x, K, v1=sc, v2=sh, act=1, gate=gate, gate_ld=K, rows_per_sample=T * H * W), L.LD_BNACT)
    h = x.float() * sc + sh
    u = h * torch.sigmoid(h) * gate.repeat_interleave(T * H * W, 0)
    assert rel(got, rt(u) @ w.double().t()) < tol(dtype, 1e-5, 6e-3)
    # AFFINE2
    y = torch.randn(M, K, device=dev()).to(dtype)
    a1, a2, a3 = (torch.randn(K, device=dev()) for _ in range(3))
    got = run(load_desc(L, x, K, q=y, v1=a1, v2=a2, v3=a3), L.LD_AFFINE2)
    v = a1 * x.float() + a2 * y.float() + a3
    assert rel(got, rt(v) @ w.double().t()) < tol(dtype, 1e-5, 6e-3)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_gemm_nn_grouped(L, dtype):
    torch.manual_seed(5)
    M, groups, Kg, Ng = 200, 2, 32, 48
    x = torch.randn(M, groups * Kg, device=dev()).to(dtype)
    w = (torch.randn(groups * Ng, Kg, device=dev()) / Kg ** 0.5).to(dtype)
    out = torch.empty(M, groups * Ng, dtype=dtype, device=dev())
    g = L.GemmNNArgs()
    g.a = load_desc(L, x, groups * Kg); g.a_kind = L.LD_PLAIN
    g.b = w.data_ptr(); g.ldb = Kg; g.c = out.data_ptr(); g.ldc = groups * Ng
    g.M, g.N, g.K, g.groups = M, Ng, Kg, groups
    L.check(L.lib.dwn_gemm_nn(C.byref(g), _dt(L, dtype), 0, stream()), "gemm_nn")
    torch.cuda.synchronize()
    ref = torch.cat([x[:, i * Kg:(i + 1) * Kg].double() @ w[i * Ng:(i + 1) * Ng].double().t() for i in range(groups)], 1)
    assert rel(out, ref) < tol(dtype, 1e-5, 6e-3)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("M,R,Cc", [(256, 64, 448), (1000, 448, 64), (333, 24, 8), (4100, 256, 136)])
def test_gemm_tn(L, dtype, M, R, Cc):
    torch.manual_seed(M)
    p = torch.randn(M, R, device=dev()).to(dtype)
    q = torch.randn(M, Cc, device=dev()).to(dtype)
    dw = torch.zeros(R, Cc, device=dev())
    g = L.GemmTNArgs()
    g.p = load_desc(L, p, R); g.p_kind = L.LD_PLAIN
    g.q = load_desc(L, q, Cc); g.q_kind = L.LD_PLAIN
    g.M, g.R, g.Cc = M, R, Cc
    g.dw = dw.data_ptr(); g.lddw = Cc; g.groups = 1; g.nsplit = 0
    L.check(L.lib.dwn_gemm_tn(C.byref(g), _dt(L, dtype), 0, stream()), "gemm_tn")
    torch.cuda.synchronize()
    ref = p.double().t() @ q.double()
    assert rel(dw, ref) < 2e-5


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_gemm_tn_exact(L, dtype):
    M, R, Cc = 96, 32, 48
    p = ((torch.arange(M, device=dev())[:, None] * 2 + torch.arange(R, device=dev())[None, :]) % 5 - 2).float()
    q = ((torch.arange(M, device=dev())[:, None] + 3 * torch.arange(Cc, device=dev())[None, :]) % 7 - 3).float()
    dw = torch.zeros(R, Cc, device=dev())
    pt_, qt_ = p.to(dtype), q.to(dtype)
    g = L.GemmTNArgs()
    g.p = load_desc(L, pt_, R); g.p_kind = L.LD_PLAIN
    g.q = load_desc(L, qt_, Cc); g.q_kind = L.LD_PLAIN
    g.M, g.R, g.Cc = M, R, Cc
    g.dw = dw.data_ptr(); g.lddw = Cc; g.groups = 1; g.nsplit = 0
    L.check(L.lib.dwn_gemm_tn(C.byref(g), _dt(L, dtype), 0, stream()), "gemm_tn")
    torch.cuda.synchronize()
    assert torch.equal(dw, p.t() @ q)


def test_gemm_nn_rejects_unsupported_requests(L):
    """The C-ABI fails loudly (non-zero code + dwn_last_error) instead of computing something else."""
    M, N, K = 256, 128, 128
    a = torch.zeros(M, K, device=dev(), dtype=torch.bfloat16)
    b = torch.zeros(N, K, device=dev(), dtype=torch.bfloat16)
    c = torch.empty(M, N, device=dev(), dtype=torch.bfloat16)
    dummy = torch.zeros(N, device=dev())

    def base():
        g = L.GemmNNArgs()
        g.a = load_desc(L, a, K); g.a_kind = L.LD_PLAIN
        g.b = b.data_ptr(); g.ldb = K; g.c = c.data_ptr(); g.ldc = N
        g.M, g.N, g.K, g.groups = M, N, K, 1
        g.epi = L.EPI_STORE
        return g

    g = base()                                   # per-sample weights need whole 128-row tiles per sample
    g.b_sample_stride = N * K; g.b_rows_per_sample = 96
    assert L.lib.dwn_gemm_nn(C.byref(g), L.DWN_BF16, 0, stream()) != 0
    assert b"per-sample" in L.lib.dwn_last_error()
    g = base()                                   # the dg epilogue reads the activated z3: s3 / t3 are reserved
    g.epi = L.EPI_DG; g.y3 = c.data_ptr(); g.ldy3 = N; g.dg = dummy.data_ptr(); g.dg_ld = N; g.rows_per_sample = 128
    g.s3 = dummy.data_ptr(); g.t3 = dummy.data_ptr()
    assert L.lib.dwn_gemm_nn(C.byref(g), L.DWN_BF16, 0, stream()) != 0
    g = base()                                   # K must be a multiple of the 16-byte vector
    g.K = 12
    assert L.lib.dwn_gemm_nn(C.byref(g), L.DWN_BF16, 0, stream()) != 0
    torch.cuda.synchronize()


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_gemm_nn_per_sample_weights(L, dtype):
    """Rows of sample b multiply weight matrix b (the SE gate folded into conv_pwl's weights)."""
    nb, rps, N, K = 3, 256, 64, 192
    M = nb * rps
    torch.manual_seed(11)
    a = torch.randn(M, K, device=dev()).to(dtype)
    w = (torch.randn(nb, N, K, device=dev()) / K ** 0.5).to(dtype)
    c = torch.empty(M, N, dtype=dtype, device=dev())
    g = L.GemmNNArgs()
    g.a = load_desc(L, a, K); g.a_kind = L.LD_PLAIN
    g.b = w.data_ptr(); g.ldb = K; g.c = c.data_ptr(); g.ldc = N
    g.M, g.N, g.K, g.groups = M, N, K, 1
    g.epi = L.EPI_STORE; g.b_sample_stride = N * K; g.b_rows_per_sample = rps
    L.check(L.lib.dwn_gemm_nn(C.byref(g), _dt(L, dtype), 0, stream()), "gemm_nn")
    torch.cuda.synchronize()
    ref = torch.cat([a[i * rps:(i + 1) * rps].double() @ w[i].double().t() for i in range(nb)])
    assert rel(c, ref) < tol(dtype, 1e-5, 6e-3)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("B,S,R,Cc", [(4, 256, 64, 448), (3, 200, 24, 40), (32, 1280, 128, 136), (2, 64, 8, 8)])
def test_gemm_tn_per_sample_products(L, dtype, B, S, R, Cc):
    """rows_per_sample > 0: B separate [R][Cc] products P_b = P_b^T Q_b (conv_pwl backward, dwn_api.hip)."""
    torch.manual_seed(S)
    M = B * S
    p = torch.randint(-2, 3, (M, R), device=dev()).float().to(dtype)
    q = torch.randint(-2, 3, (M, Cc), device=dev()).float().to(dtype)
    stride = R * Cc + 16                                   # padded sample stride
    dw = torch.zeros(B * stride, device=dev())
    g = L.GemmTNArgs()
    g.p = load_desc(L, p, R); g.p_kind = L.LD_PLAIN
    g.q = load_desc(L, q, Cc); g.q_kind = L.LD_PLAIN
    g.M, g.R, g.Cc = M, R, Cc
    g.dw = dw.data_ptr(); g.lddw = Cc; g.groups = 1; g.nsplit = 0
    g.rows_per_sample = S; g.dw_sample_stride = stride
    L.check(L.lib.dwn_gemm_tn(C.byref(g), _dt(L, dtype), 0, stream()), "gemm_tn")
    torch.cuda.synchronize()
    got = dw.view(B, stride)
    ref = torch.einsum("bsr,bsc->brc", p.float().view(B, S, R), q.float().view(B, S, Cc))      # small integers: exact
    assert torch.equal(got[:, :R * Cc].reshape(B, R, Cc), ref)
    assert not got[:, R * Cc:].any()
    g.M = M - 1
    assert L.lib.dwn_gemm_tn(C.byref(g), _dt(L, dtype), 0, stream()) < 0       # M must be whole samples


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("B,S,N,K", [(4, 256, 448, 64), (3, 200, 40, 24), (2, 640, 896, 128), (2, 128, 136, 256),
                                     (5, 100, 64, 16)])
def test_gemm_nn_dh3_epilogue(L, dtype, B, S, N, K):
    """DWN_EPI_DH3: dh3 = (round(A.B^T)*gate + dps) * silu'(scale*y3 + shift) stored, plus the two BatchNorm-backward
    sums — against the same formula in fp64 on the rounded product (whole-tile and ragged shapes, resident and k-loop)."""
    torch.manual_seed(N + K)
    M = B * S
    a = (torch.randn(M, K, device=dev()) * 0.5).to(dtype)
    b = (torch.randn(N, K, device=dev()) * 0.2).to(dtype)
    y3 = torch.randn(M, N, device=dev()).to(dtype)
    gate = torch.rand(B, N, device=dev())
    dps = torch.randn(B, N, device=dev()) * 0.1
    coef = torch.stack([torch.rand(N, device=dev()) + 0.5, torch.randn(N, device=dev()) * 0.3,
                        torch.randn(N, device=dev()) * 0.2, torch.rand(N, device=dev()) + 0.5]).contiguous()
    out = torch.full((M, N), float("nan"), device=dev()).to(dtype)
    st = stats_buffer(N)
    g = L.GemmNNArgs()
    g.a = load_desc(L, a, K); g.a_kind = L.LD_PLAIN
    g.b = b.data_ptr(); g.ldb = K; g.c = out.data_ptr(); g.ldc = N
    g.M, g.N, g.K, g.groups = M, N, K, 1
    g.epi = L.EPI_DH3
    g.y3 = y3.data_ptr(); g.ldy3 = N; g.gate3 = gate.data_ptr(); g.dps3 = dps.data_ptr(); g.dg_ld = N
    g.coef3 = coef.data_ptr(); g.coef3_ld = N; g.rows_per_sample = S
    g.stats = st.data_ptr(); g.stat_nchan = N
    L.check(L.lib.dwn_gemm_nn(C.byref(g), _dt(L, dtype), 0, stream()), "gemm_nn dh3")
    torch.cuda.synchronize()
    du = (a.double() @ b.double().t()).to(dtype).double()                      # the product is rounded to the storage type
    yd = y3.double()
    h = yd * coef[0].double() + coef[1].double()
    sg = torch.sigmoid(h)
    gb = gate.double().repeat_interleave(S, 0)
    pb = dps.double().repeat_interleave(S, 0)
    dh = (du * gb + pb) * (sg * (1 + h * (1 - sg)))
    t = 1e-5 if dtype == torch.float32 else 1e-2
    assert torch.isfinite(out.float()).all()
    assert rel(out, dh) < t
    r = out.double()                                                             # the sums use the stored (rounded) values
    s0, s1 = read_stats(st, N)
    want0 = r.sum(0)
    want1 = (r * (yd - coef[2].double()) * coef[3].double()).sum(0)
    scale = r.abs().sum(0).max()
    assert float((s0 - want0).abs().max() / scale) < 1e-5
    assert float((s1 - want1).abs().max() / scale) < 1e-5
    g.gate3 = None
    assert L.lib.dwn_gemm_nn(C.byref(g), _dt(L, dtype), 0, stream()) < 0


@pytest.mark.parametrize("M,E,Cin,dtype", [(128, 448, 64, torch.bfloat16), (128 * 37, 448, 64, torch.bfloat16),
                                           (128 * 513, 448, 64, torch.bfloat16),        # one-pass kernel (64-channel blocks)
                                           (128 * 40, 384, 64, torch.bfloat16),         # ... of the expansion-6 student
                                           (128 * 37 + 64, 448, 64, torch.bfloat16),    # same widths, ragged M: two GEMMs
                                           (5000, 896, 128, torch.bfloat16), (3000, 1792, 256, torch.bfloat16),
                                           (5000, 384, 64, torch.bfloat16),             # E not a multiple of the 128-row tile
                                           (5000, 448, 64, torch.float32), (3000, 896, 128, torch.float32)])
def test_pw_backward_without_y1(L, M, E, Cin, dtype):
    """dwn_pw_backward: conv_pw data and weight gradient from (dh1, a0, W1, abc) alone — y1 = a0 . W1^T is never read, its
    terms are folded into Cin x Cin matrices (include/dwn.h).  Against float64: dy1 = A1*dh1 + A2*(a0 W1^T) + A3;
    da0 = dy1 W1; dW = dy1^T a0, with W1 as rounded to the storage dtype."""
    fused = L.lib.dwn_pw_bwd_fused_supported(_dt(L, dtype), M, E, Cin)
    assert fused == (1 if (dtype == torch.bfloat16 and E in (448, 384) and Cin == 64 and M % 128 == 0) else 0)
    g = torch.Generator(device="cuda").manual_seed(M + E)
    dh1 = torch.randn(M, E, generator=g, device=dev()).to(dtype)
    a0 = (torch.randn(M, Cin, generator=g, device=dev()) + 0.3).to(dtype)           # a non-zero column mean: the A3 term counts
    w1 = (torch.randn(E, Cin, generator=g, device=dev()) * 0.1).contiguous()
    abc = torch.randn(3, E, generator=g, device=dev()).contiguous()
    da0 = torch.full((M, Cin), float("nan"), device=dev()).to(dtype)
    dw = torch.full((E, Cin), float("nan"), device=dev())                            # overwritten
    nws = L.lib.dwn_pw_backward_workspace_bytes(E, Cin, _dt(L, dtype))
    ws = torch.empty(nws, dtype=torch.uint8, device=dev())
    a = L.PwBwdArgs()
    a.dh1, a.a0, a.w_pw, a.abc = dh1.data_ptr(), a0.data_ptr(), w1.data_ptr(), abc.data_ptr()
    a.da0, a.dw, a.M, a.E, a.Cin, a.ws, a.ws_bytes = da0.data_ptr(), dw.data_ptr(), M, E, Cin, ws.data_ptr(), nws
    L.check(L.lib.dwn_pw_backward(C.byref(a), _dt(L, dtype), 0, stream()), "pw_backward")
    torch.cuda.synchronize()
    w1r = w1.to(dtype).double()
    y1 = a0.double() @ w1r.t()
    dy = abc[0].double() * dh1.double() + abc[1].double() * y1 + abc[2].double()
    want_da = dy @ w1r
    want_dw = dy.t() @ a0.double()
    assert torch.isfinite(da0.float()).all() and torch.isfinite(dw).all()
    assert rel(da0, want_da) < (8e-3 if dtype == torch.bfloat16 else 2e-5)         # bf16: output rounding + G kept in bf16
    assert rel(dw, want_dw) < (2e-4 if dtype == torch.bfloat16 else 2e-5)           # bf16: only W1's Gram row is rounded
    # the three terms separately: each column of dW
    for sel in range(3):
        ab = torch.zeros_like(abc); ab[sel] = abc[sel]
        a.abc = ab.data_ptr()
        L.check(L.lib.dwn_pw_backward(C.byref(a), _dt(L, dtype), 0, stream()), "pw_backward")
        torch.cuda.synchronize()
        dy = ab[0].double() * dh1.double() + ab[1].double() * y1 + ab[2].double()
        assert rel(dw, dy.t() @ a0.double()) < (2e-3 if dtype == torch.bfloat16 else 2e-5), sel
    a.abc = abc.data_ptr()
    # ... and with a stride-1 shortcut branch folded in (res = the block's output gradient, 1x or 2x the input channels)
    dy = abc[0].double() * dh1.double() + abc[1].double() * y1 + abc[2].double()
    for mult in (1, 2):
        rc = mult * Cin
        if not fused:            # built into the one-pass kernel only
            res = torch.zeros(M, rc, device=dev()).to(dtype)
            rabc = torch.zeros(3, rc, device=dev())
            a.res, a.res_abc, a.res_C = res.data_ptr(), rabc.data_ptr(), rc
            assert L.lib.dwn_pw_backward(C.byref(a), _dt(L, dtype), 0, stream()) == -3
            continue
        res = torch.randn(M, rc, generator=g, device=dev()).to(dtype)
        rabc = torch.randn(3, rc, generator=g, device=dev()).contiguous()
        a.res, a.res_abc, a.res_C = res.data_ptr(), rabc.data_ptr(), rc
        L.check(L.lib.dwn_pw_backward(C.byref(a), _dt(L, dtype), 0, stream()), "pw_backward")
        torch.cuda.synchronize()
        want = dy @ w1r
        for j in range(mult):
            sl = slice(j * Cin, (j + 1) * Cin)
            want = want + rabc[0, sl].double() * res[:, sl].double() + rabc[1, sl].double() * a0.double() + rabc[2, sl].double()
        assert rel(da0, want) < (8e-3 if dtype == torch.bfloat16 else 2e-5), mult
        assert rel(dw, want_dw) < (2e-4 if dtype == torch.bfloat16 else 2e-5)       # the weight gradient does not see it
    if fused:
        a.res_C = 3 * Cin
        assert L.lib.dwn_pw_backward(C.byref(a), _dt(L, dtype), 0, stream()) < 0
    a.res, a.res_abc, a.res_C = None, None, 0
    a.ws_bytes = nws - 1
    assert L.lib.dwn_pw_backward(C.byref(a), _dt(L, dtype), 0, stream()) == -6
    a.ws_bytes, a.dw = nws, None
    assert L.lib.dwn_pw_backward(C.byref(a), _dt(L, dtype), 0, stream()) < 0


@pytest.mark.parametrize("E,mult", [(448, 1), (448, 2), (384, 1)])
def test_pw_backward_gathered_shortcut(L, E, mult):
    """dwn_pw_backward with the shortcut branch of a STRIDED block in the one-pass kernel's epilogue: only the rows the nearest map
    samples (here every second row and column) get  A1sc*dout[rout] + A2sc*a0 + A3sc  (include/dwn.h dwn_pw_bwd_args.res_hinv)."""
    Cin, F, Hin, Win = 64, 5, 32, 32
    Hout, Wout = 16, 16
    M, rc = F * Hin * Win, mult * Cin
    dtype = torch.bfloat16
    assert L.lib.dwn_pw_bwd_fused_supported(L.DWN_BF16, M, E, Cin) == 1
    g = torch.Generator(device="cuda").manual_seed(E + mult)
    dh1 = torch.randn(M, E, generator=g, device=dev()).to(dtype)
    a0 = (torch.randn(M, Cin, generator=g, device=dev()) + 0.3).to(dtype)
    w1 = (torch.randn(E, Cin, generator=g, device=dev()) * 0.1).contiguous()
    abc = torch.randn(3, E, generator=g, device=dev()).contiguous()
    res = torch.randn(F * Hout * Wout, rc, generator=g, device=dev()).to(dtype)
    rabc = torch.randn(3, rc, generator=g, device=dev()).contiguous()
    hinv = torch.full((Hin,), -1, dtype=torch.int32); hinv[::2] = torch.arange(Hout, dtype=torch.int32)
    winv = torch.full((Win,), -1, dtype=torch.int32); winv[::2] = torch.arange(Wout, dtype=torch.int32)
    hinv_d, winv_d = hinv.to(dev()), winv.to(dev())
    da0 = torch.full((M, Cin), float("nan"), device=dev()).to(dtype)
    dw = torch.full((E, Cin), float("nan"), device=dev())
    nws = L.lib.dwn_pw_backward_workspace_bytes(E, Cin, L.DWN_BF16)
    ws = torch.empty(nws, dtype=torch.uint8, device=dev())
    a = L.PwBwdArgs()
    a.dh1, a.a0, a.w_pw, a.abc = dh1.data_ptr(), a0.data_ptr(), w1.data_ptr(), abc.data_ptr()
    a.da0, a.dw, a.M, a.E, a.Cin, a.ws, a.ws_bytes = da0.data_ptr(), dw.data_ptr(), M, E, Cin, ws.data_ptr(), nws
    a.res, a.res_abc, a.res_C = res.data_ptr(), rabc.data_ptr(), rc
    a.res_hinv, a.res_winv = hinv_d.data_ptr(), winv_d.data_ptr()
    a.res_Hin, a.res_Win, a.res_Hout, a.res_Wout = Hin, Win, Hout, Wout
    L.check(L.lib.dwn_pw_backward(C.byref(a), L.DWN_BF16, 0, stream()), "pw_backward")
    torch.cuda.synchronize()
    w1r = w1.to(dtype).double()
    y1 = a0.double() @ w1r.t()
    dy = abc[0].double() * dh1.double() + abc[1].double() * y1 + abc[2].double()
    want = (dy @ w1r).view(F, Hin, Win, Cin).clone()
    a0v = a0.double().view(F, Hin, Win, Cin)
    resv = res.double().view(F, Hout, Wout, rc)
    for j in range(mult):
        sl = slice(j * Cin, (j + 1) * Cin)
        want[:, ::2, ::2] += rabc[0, sl].double() * resv[..., sl] + rabc[1, sl].double() * a0v[:, ::2, ::2] + rabc[2, sl].double()
    assert torch.isfinite(da0.float()).all()
    assert rel(da0, want.view(M, Cin)) < 8e-3
    # the rows the map skips are exactly the plain data gradient
    a.res = None
    da1 = torch.empty_like(da0); a.da0 = da1.data_ptr()
    L.check(L.lib.dwn_pw_backward(C.byref(a), L.DWN_BF16, 0, stream()), "pw_backward")
    torch.cuda.synchronize()
    skip = torch.ones(F, Hin, Win, dtype=torch.bool, device=dev()); skip[:, ::2, ::2] = False
    # (not bit for bit: G is accumulated with fp32 atomics in arrival order and then rounded to bf16)
    assert rel(da0.view(F, Hin, Win, Cin)[skip], da1.view(F, Hin, Win, Cin)[skip].double()) < 2e-3
    assert rel(da0.view(F, Hin, Win, Cin)[~skip], da1.view(F, Hin, Win, Cin)[~skip].double()) > 5e-2     # the sampled rows did change
    assert rel(dw, dy.t() @ a0.double()) < 2e-4


def test_conv_pw_batchnorm_statistics_from_the_gram_matrix(L):
    """dwn_conv_pw_bn_stats: BatchNorm-1's batch statistics of y1 = a0 . W1^T from a0^T a0 and 1^T a0 (y1 never computed), against
    float64 statistics of the product of the same rounded operands: mean / variance to 1e-6 (of the standard deviation / of the
    variance), running statistics like nn.BatchNorm3d, and the shortcut BatchNorm's raw sums of a0."""
    import ctypes as C
    torch.manual_seed(5)
    for dtype, M, E, Cin in ((torch.bfloat16, 40000, 448, 64), (torch.bfloat16, 9001, 896, 128), (torch.float32, 5000, 192, 64)):
        # inputs with per-channel offsets (the positional encoding adds constants of order one) and unequal scales
        a0 = (torch.randn(M, Cin, device=dev()) * (0.5 + torch.rand(Cin, device=dev())) + torch.randn(Cin, device=dev())).to(dtype)
        w = torch.randn(E, Cin, device=dev()) / Cin ** 0.5
        gamma = torch.rand(E, device=dev()) + 0.5
        beta = torch.randn(E, device=dev()) * 0.2
        rm = torch.randn(E, device=dev()) * 0.1
        rv = torch.rand(E, device=dev()) + 0.5
        rm0, rv0 = rm.clone(), rv.clone()
        nbt = torch.zeros(1, dtype=torch.int64, device=dev())
        coef = torch.empty(4 * E, device=dev())
        sc = torch.zeros(32 * 2 * Cin, dtype=torch.float64, device=dev())
        ws = torch.empty(L.lib.dwn_conv_pw_bn_stats_workspace_bytes(Cin), dtype=torch.uint8, device=dev())
        bn = L.BN()
        bn.gamma = gamma.data_ptr(); bn.beta = beta.data_ptr(); bn.running_mean = rm.data_ptr(); bn.running_var = rv.data_ptr()
        bn.num_batches_tracked = nbt.data_ptr(); bn.coef = coef.data_ptr()
        L.check(L.lib.dwn_conv_pw_bn_stats(a0.data_ptr(), Cin, M, w.data_ptr(), E, Cin, C.byref(bn), 0.1, 1e-5, sc.data_ptr(),
                                           ws.data_ptr(), ws.numel(), _dt(L, dtype), 0, stream()), "conv_pw_bn_stats")
        torch.cuda.synchronize()
        y = a0.double() @ w.to(dtype).double().t()
        mean, var = y.mean(0), y.var(0, unbiased=False)
        c = coef.view(4, E).double()
        assert float(((c[2] - mean).abs() / var.sqrt()).max()) < 1e-6
        invstd = 1.0 / (var + 1e-5).sqrt()
        assert float(((c[3] - invstd).abs() / invstd).max()) < (1e-6 if dtype == torch.float32 else 3e-6)
        assert rel(c[0], gamma.double() * invstd) < 1e-6 and rel(c[1], beta.double() - mean * gamma.double() * invstd) < 1e-5
        assert rel(rm, 0.9 * rm0.double() + 0.1 * mean) < 1e-6
        assert rel(rv, 0.9 * rv0.double() + 0.1 * var * M / (M - 1)) < 1e-6
        assert int(nbt) == 1
        s0, s1 = read_stats(sc, Cin)
        assert rel(s0, a0.double().sum(0)) < 1e-6 and rel(s1, (a0.double() ** 2).sum(0)) < 1e-6


def test_conv_pw_batchnorm_statistics_from_the_gram_matrix_at_the_metric_batch(L):
    """Round-5 advisor: the Gram route to BatchNorm-1's statistics obtains the variance by cancellation, w^T (G / n - mu mu^T) w, and
    was only tested to M = 40000 with independent channels.  Here: the row count of block 0 at the metric batch (M = 32 * 32 * 36 * 64
    = 2 359 296) and of blocks 1-3 (589 824), CORRELATED channels (a random mixing of 8 latent factors + noise) with LARGE means
    (offsets of three standard deviations: (mean^2 + var) / var ~ 10 amplifies the relative error of the raw sums) — against float64
    statistics of the product of the same rounded operands.  Mean to 1e-6 sigma, invstd to 3e-6: the raw products are added with
    fp64 atomics since round 6 (fp32 atomics measured 1e-5 here)."""
    import ctypes as C
    torch.manual_seed(11)
    for M, E, Cin in ((2359296, 448, 64), (589824, 448, 64), (147456, 896, 128)):
        lat = torch.randn(M, 8, device=dev())
        mix = torch.randn(8, Cin, device=dev())
        scale = 0.5 + torch.rand(Cin, device=dev())
        a0 = ((lat @ mix) * 0.6 + torch.randn(M, Cin, device=dev()) * 0.5) * scale + 3.0 * scale * torch.sign(torch.randn(Cin, device=dev()))
        a0 = a0.to(torch.bfloat16)
        del lat
        w = torch.randn(E, Cin, device=dev()) / Cin ** 0.5
        gamma = torch.rand(E, device=dev()) + 0.5
        beta = torch.randn(E, device=dev()) * 0.2
        rm, rv = torch.zeros(E, device=dev()), torch.ones(E, device=dev())
        nbt = torch.zeros(1, dtype=torch.int64, device=dev())
        coef = torch.empty(4 * E, device=dev())
        sc = torch.zeros(32 * 2 * Cin, dtype=torch.float64, device=dev())
        ws = torch.empty(L.lib.dwn_conv_pw_bn_stats_workspace_bytes(Cin), dtype=torch.uint8, device=dev())
        bn = L.BN()
        bn.gamma = gamma.data_ptr(); bn.beta = beta.data_ptr(); bn.running_mean = rm.data_ptr(); bn.running_var = rv.data_ptr()
        bn.num_batches_tracked = nbt.data_ptr(); bn.coef = coef.data_ptr()
        L.check(L.lib.dwn_conv_pw_bn_stats(a0.data_ptr(), Cin, M, w.data_ptr(), E, Cin, C.byref(bn), 0.1, 1e-5, sc.data_ptr(),
                                           ws.data_ptr(), ws.numel(), L.DWN_BF16, 0, stream()), "conv_pw_bn_stats")
        torch.cuda.synchronize()
        # float64 statistics of the product, in row chunks (the product of the largest case is 8.5 GB in float64)
        wd = w.to(torch.bfloat16).double()
        s1 = torch.zeros(E, dtype=torch.float64, device=dev())
        for r0 in range(0, M, 262144):
            s1 += (a0[r0:r0 + 262144].double() @ wd.t()).sum(0)
        mean = s1 / M
        s2 = torch.zeros(E, dtype=torch.float64, device=dev())
        for r0 in range(0, M, 262144):
            s2 += ((a0[r0:r0 + 262144].double() @ wd.t() - mean) ** 2).sum(0)
        var = s2 / M
        c = coef.view(4, E).double()
        assert float((mean ** 2 / var).max()) > 5.0                    # the case does amplify
        assert float(((c[2] - mean).abs() / var.sqrt()).max()) < 1e-6, (M, float(((c[2] - mean).abs() / var.sqrt()).max()))
        invstd = 1.0 / (var + 1e-5).sqrt()
        assert float(((c[3] - invstd).abs() / invstd).max()) < 3e-6, (M, float(((c[3] - invstd).abs() / invstd).max()))
        assert rel(rv, 0.9 + 0.1 * var * M / (M - 1)) < 3e-6
