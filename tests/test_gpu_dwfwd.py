"""Chained row-walk spatial depth-wise forward kernels (sensorium_amd/csrc/dwn_dwfwd.hip; reference op
src/models/dwiseneuro.py:96-102) against the library's second implementation, the pair kernel (dwn_dw_spatial_fwd_args.impl = 1),
through the C-ABI: y2 BIT-identical, BatchNorm-2 sums to summation order, both strides.  (The pair kernel is pinned to the
oracle by tests/test_gpu_block.py.)"""
import ctypes as C

import pytest
import torch

pytestmark = pytest.mark.gpu

import sensorium_amd._lib as L  # noqa: E402
from tests.gpu_helpers import dev  # noqa: E402

BF = torch.bfloat16


def _both(planes, Hin, Win, Cc, stride, rows_band=0, seed=0):
    d = dev()
    s = torch.cuda.current_stream().cuda_stream
    g = torch.Generator(device=d); g.manual_seed(seed)
    Hout, Wout = (Hin - 1) // stride + 1, (Win - 1) // stride + 1
    x = torch.randn(planes * Hin * Win, Cc, device=d, generator=g).to(BF)
    coef = torch.cat([torch.rand(Cc, device=d, generator=g) + 0.5, torch.randn(Cc, device=d, generator=g) * 0.3])
    w = torch.randn(9, Cc, device=d, generator=g) / 3.0
    out = {}
    for mode in ("old", "new"):
        y2 = torch.full((planes * Hout * Wout, Cc), float("nan"), dtype=BF, device=d)
        st = torch.zeros(32 * 2 * Cc, dtype=torch.float64, device=d)
        a = L.DwSpatialFwdArgs()
        di = L.LoadDesc()
        di.p = x.data_ptr(); di.ld = Cc; di.rows_per_sample = 1; di.v1 = coef.data_ptr(); di.v2 = coef[Cc:].data_ptr(); di.act = 1
        a.inp = di
        a.w = w.data_ptr(); a.out = y2.data_ptr(); a.planes = planes; a.Hin = Hin; a.Win = Win; a.Hout = Hout
        a.Wout = Wout; a.C = Cc; a.stride = stride; a.ks = 3; a.stats = st.data_ptr()
        a.impl = 1 if mode == "old" else 0
        a.rows_band = rows_band if mode == "new" else 0
        L.check(L.lib.dwn_dw_spatial_fwd(C.byref(a), L.DWN_BF16, d.index, s), "dwn_dw_spatial_fwd")
        torch.cuda.synchronize()
        out[mode] = (y2, st.view(32, 2, Cc).sum(0))
    return out["old"], out["new"]


@pytest.mark.parametrize("case", [(3, 18, 32, 64, 1), (5, 9, 16, 128, 1), (7, 5, 8, 64, 1), (2, 3, 32, 72, 1), (9, 1, 8, 64, 1),
                                  (1, 20, 16, 64, 1), (130, 9, 16, 448, 1), (131, 5, 8, 448, 1), (1, 2, 32, 64, 1), (4, 7, 8, 200, 1),
                                  # stride 2 (on by default since round 2; same cases as tests/test_gpu_dwbwd.py)
                                  (3, 36, 64, 64, 2), (7, 9, 16, 64, 2), (2, 4, 64, 72, 2), (129, 18, 32, 448, 2), (5, 18, 32, 128, 2),
                                  (9, 1, 16, 64, 2), (3, 7, 32, 64, 2), (3, 5, 16, 64, 2), (130, 9, 16, 448, 2)])
def test_fwd_walk_matches_pair_kernel(case):
    (y0, s0), (y1, s1) = _both(*case)
    assert not torch.isnan(y1.float()).any()
    assert torch.equal(y0.view(torch.int16), y1.view(torch.int16))
    assert float(((s0 - s1).abs() / (s0.abs() + 1e-2 * s0.abs().mean())).max()) < 1e-5


@pytest.mark.parametrize("stride,rows_band", [(1, 1), (1, 2), (1, 4), (1, 7), (2, 1), (2, 2), (2, 3), (2, 4)])
def test_fwd_walk_band_heights(stride, rows_band):
    """rows_band = output rows per chunk (rounded up to a built chunk height)."""
    H, W = (18, 32) if stride == 1 else (36, 64)
    (y0, s0), (y1, s1) = _both(3, H, W, 64, stride, rows_band=rows_band)
    assert torch.equal(y0.view(torch.int16), y1.view(torch.int16))
    assert float(((s0 - s1).abs() / (s0.abs() + 1e-2 * s0.abs().mean())).max()) < 1e-5


# ---- rebuilt-input mode (round 5): conv_pw + spat_covn_dw in one pass — the stencil rebuilds its y1 rows from the block input a0 ----
def _stored_vs_rebuilt(planes, Hin, Win, Cc, stride, rows_band=0, seed=0, cin=64):
    d = dev()
    s = torch.cuda.current_stream().cuda_stream
    g = torch.Generator(device=d); g.manual_seed(seed)
    Hout, Wout = (Hin - 1) // stride + 1, (Win - 1) // stride + 1
    a0 = torch.randn(planes * Hin * Win, cin, device=d, generator=g).to(BF)
    w1 = (torch.randn(Cc, cin, device=d, generator=g) / cin ** 0.5).to(BF)
    # y1 as the product path stores it: the real conv_pw GEMM
    y1 = torch.empty(planes * Hin * Win, Cc, dtype=BF, device=d)
    gm = L.GemmNNArgs()
    da = L.LoadDesc(); da.p = a0.data_ptr(); da.ld = cin; da.rows_per_sample = 1
    gm.a = da; gm.a_kind = L.LD_PLAIN; gm.b = w1.data_ptr(); gm.ldb = cin; gm.c = y1.data_ptr(); gm.ldc = Cc
    gm.M, gm.N, gm.K, gm.groups = planes * Hin * Win, Cc, cin, 1
    gm.epi = L.EPI_STORE
    L.check(L.lib.dwn_gemm_nn(C.byref(gm), L.DWN_BF16, d.index, s), "gemm_nn")
    coef = torch.cat([torch.rand(Cc, device=d, generator=g) + 0.5, torch.randn(Cc, device=d, generator=g) * 0.3])
    w = torch.randn(9, Cc, device=d, generator=g) / 3.0
    out = {}
    for mode in ("stored", "rebuilt"):
        y2 = torch.full((planes * Hout * Wout, Cc), float("nan"), dtype=BF, device=d)
        st = torch.zeros(32 * 2 * Cc, dtype=torch.float64, device=d)
        a = L.DwSpatialFwdArgs()
        di = L.LoadDesc()
        di.p = y1.data_ptr() if mode == "stored" else None
        di.ld = Cc; di.rows_per_sample = 1; di.v1 = coef.data_ptr(); di.v2 = coef[Cc:].data_ptr(); di.act = 1
        a.inp = di
        a.w = w.data_ptr(); a.out = y2.data_ptr(); a.planes = planes; a.Hin = Hin; a.Win = Win; a.Hout = Hout
        a.Wout = Wout; a.C = Cc; a.stride = stride; a.ks = 3; a.stats = st.data_ptr()
        a.rows_band = rows_band
        if mode == "rebuilt":
            a.a0 = a0.data_ptr(); a.a0_ld = cin; a.w1 = w1.data_ptr(); a.Cin = cin
            assert L.lib.dwn_dw_spatial_fwd_rc_supported(C.byref(a), L.DWN_BF16) == 1
        L.check(L.lib.dwn_dw_spatial_fwd(C.byref(a), L.DWN_BF16, d.index, s), "dwn_dw_spatial_fwd")
        torch.cuda.synchronize()
        out[mode] = (y2, st.view(32, 2, Cc).sum(0))
    return out["stored"], out["rebuilt"]


@pytest.mark.parametrize("case", [(3, 18, 32, 64, 1), (5, 9, 16, 128, 1), (7, 5, 8, 64, 1), (9, 1, 8, 64, 1), (1, 20, 16, 64, 1),
                                  (2, 3, 32, 192, 1), (130, 9, 16, 448, 1), (131, 5, 8, 448, 1), (33, 18, 32, 448, 1),
                                  (3, 36, 64, 64, 2), (7, 9, 16, 64, 2), (2, 4, 64, 128, 2), (129, 18, 32, 448, 2), (5, 18, 32, 128, 2),
                                  (9, 1, 16, 64, 2), (3, 7, 32, 64, 2), (3, 5, 16, 64, 2), (130, 9, 16, 448, 2), (40, 36, 64, 448, 2)])
def test_rebuilt_input_matches_stored_input(case):
    """The same chained stencil, its input read from HBM (conv_pw's stored bf16 output) vs rebuilt as a0 . W1^T by MFMA and
    rounded the same way: y2 BIT-identical, BatchNorm-2 sums to summation order."""
    (y0, s0), (y1, s1) = _stored_vs_rebuilt(*case)
    assert not torch.isnan(y1.float()).any()
    assert torch.equal(y0.view(torch.int16), y1.view(torch.int16))
    assert float(((s0 - s1).abs() / (s0.abs() + 1e-2 * s0.abs().mean())).max()) < 1e-5


@pytest.mark.parametrize("case", [(3, 18, 32, 128, 1), (5, 9, 16, 128, 1), (7, 5, 8, 64, 1), (130, 9, 16, 896, 1), (33, 18, 32, 192, 1),
                                  (3, 36, 64, 64, 2), (5, 18, 32, 128, 2), (7, 9, 16, 64, 2), (129, 18, 32, 896, 2), (9, 1, 16, 64, 2)])
def test_rebuilt_input_128_channels(case):
    """Cin = 128 (blocks 4-6): four k-steps per MFMA tile, the W1 slice in LDS instead of registers."""
    (y0, s0), (y1, s1) = _stored_vs_rebuilt(*case, cin=128)
    assert not torch.isnan(y1.float()).any()
    assert torch.equal(y0.view(torch.int16), y1.view(torch.int16))
    assert float(((s0 - s1).abs() / (s0.abs() + 1e-2 * s0.abs().mean())).max()) < 1e-5


@pytest.mark.parametrize("stride,rows_band", [(1, 4), (1, 8), (2, 2), (2, 4)])
def test_rebuilt_input_chunk_heights(stride, rows_band):
    H, W = (18, 32) if stride == 1 else (36, 64)
    (y0, s0), (y1, s1) = _stored_vs_rebuilt(5, H, W, 128, stride, rows_band=rows_band, seed=2)
    assert torch.equal(y0.view(torch.int16), y1.view(torch.int16))
    assert float(((s0 - s1).abs() / (s0.abs() + 1e-2 * s0.abs().mean())).max()) < 1e-5
