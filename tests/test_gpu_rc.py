"""spat_covn_dw without a materialised conv_pw output (dwn_dw_spatial_fwd_rc; reference src/models/dwiseneuro.py:90-102)
against the materialised path through the C-ABI (dwn_gemm_nn -> dwn_dw_spatial_fwd), which the block tests pin to the
oracle.  With round_y1 = 1 the rebuilt y1 tile is rounded to bf16 exactly like the stored tensor, the MFMA k-order is the
GEMM's and the stencil arithmetic is the pair kernel's: y2 must be BIT-IDENTICAL; the BatchNorm-2 sums agree to fp32
summation order."""
import ctypes as C

import pytest
import torch

pytestmark = pytest.mark.gpu

import sensorium_amd._lib as L  # noqa: E402
from tests.gpu_helpers import dev  # noqa: E402

BF = torch.bfloat16


def _desc(p, ld, **kw):
    d = L.LoadDesc()
    d.p = p.data_ptr(); d.ld = ld; d.rows_per_sample = 1
    for k, v in kw.items():
        setattr(d, k, v.data_ptr() if isinstance(v, torch.Tensor) else v)
    return d


def _run(planes, Hin, Win, Cin, E, stride, rows_band=0, round_y1=1, seed=0):
    d = dev()
    s = torch.cuda.current_stream().cuda_stream
    g = torch.Generator(device=d); g.manual_seed(seed)
    Hout, Wout = (Hin - 1) // stride + 1, (Win - 1) // stride + 1
    Min, Mout = planes * Hin * Win, planes * Hout * Wout
    a0 = torch.randn(Min, Cin, device=d, generator=g).to(BF)
    w1 = torch.randn(E, Cin, device=d, generator=g) / Cin ** 0.5
    coef = torch.cat([torch.rand(E, device=d, generator=g) + 0.5, torch.randn(E, device=d, generator=g) * 0.3])
    wdw = torch.randn(9, E, device=d, generator=g) / 3.0
    w1p = torch.empty(E, Cin, dtype=BF, device=d)
    L.check(L.lib.dwn_pack_weight(w1.data_ptr(), w1p.data_ptr(), 1, E, Cin, 0, E, Cin, L.DWN_BF16, d.index, s), "pack")
    y1 = torch.empty(Min, E, dtype=BF, device=d)
    gm = L.GemmNNArgs()
    gm.a = _desc(a0, Cin); gm.a_kind = L.LD_PLAIN; gm.b = w1p.data_ptr(); gm.ldb = Cin; gm.c = y1.data_ptr(); gm.ldc = E
    gm.M, gm.N, gm.K, gm.groups = Min, E, Cin, 1
    gm.stats = None; gm.stat_nchan = E; gm.epi = L.EPI_STORE
    L.check(L.lib.dwn_gemm_nn(C.byref(gm), L.DWN_BF16, d.index, s), "nn")
    y2r = torch.empty(Mout, E, dtype=BF, device=d)
    str_ = torch.zeros(32 * 2 * E, dtype=torch.float64, device=d)
    fa = L.DwSpatialFwdArgs()
    fa.inp = _desc(y1, E, v1=coef, v2=coef[E:], act=1)
    fa.w = wdw.data_ptr(); fa.out = y2r.data_ptr(); fa.planes = planes; fa.Hin = Hin; fa.Win = Win; fa.Hout = Hout
    fa.Wout = Wout; fa.C = E; fa.stride = stride; fa.ks = 3; fa.stats = str_.data_ptr(); fa.rows_band = 0
    L.check(L.lib.dwn_dw_spatial_fwd(C.byref(fa), L.DWN_BF16, d.index, s), "dws")
    blob = torch.zeros(L.lib.dwn_dw_spatial_rc_blob_bytes(E, Cin), dtype=torch.uint8, device=d)
    L.check(L.lib.dwn_dw_spatial_rc_prep(w1.data_ptr(), wdw.data_ptr(), coef.data_ptr(), E, Cin, blob.data_ptr(), d.index, s), "prep")
    y2 = torch.full((Mout, E), float("nan"), dtype=BF, device=d)
    st = torch.zeros(32 * 2 * E, dtype=torch.float64, device=d)
    ra = L.DwSpatialRcFwdArgs()
    ra.a0 = a0.data_ptr(); ra.a0_ld = Cin; ra.blob = blob.data_ptr(); ra.out = y2.data_ptr()
    ra.planes = planes; ra.Hin = Hin; ra.Win = Win; ra.Hout = Hout; ra.Wout = Wout; ra.Cin = Cin; ra.E = E
    ra.stride = stride; ra.stats = st.data_ptr(); ra.rows_band = rows_band; ra.round_y1 = round_y1
    L.check(L.lib.dwn_dw_spatial_fwd_rc(C.byref(ra), d.index, s), "rc")
    torch.cuda.synchronize()
    return y2, y2r, st.view(32, 2, E).sum(0), str_.view(32, 2, E).sum(0)


CASES = [
    # planes, Hin, Win, Cin, E, stride
    (3, 18, 32, 64, 128, 1), (3, 36, 64, 64, 64, 2), (2, 9, 16, 128, 192, 1), (2, 18, 32, 128, 128, 2),
    (5, 7, 5, 64, 64, 1), (5, 7, 5, 64, 64, 2), (4, 5, 8, 128, 64, 1), (3, 10, 11, 64, 128, 2),
    (3, 1, 2, 64, 64, 1), (2, 2, 3, 64, 64, 2), (300, 9, 16, 64, 448, 1),
]


@pytest.mark.parametrize("case", CASES)
def test_rc_forward_bit_identical_to_materialised(case):
    y2, y2r, st, str_ = _run(*case)
    assert not torch.isnan(y2.float()).any()
    assert torch.equal(y2.view(torch.int16), y2r.view(torch.int16)), "recomputed-y1 stencil output differs from the stored-y1 path"
    assert float(((st - str_).abs() / (str_.abs() + 1e-3)).max()) < 1e-5


@pytest.mark.parametrize("stride", [1, 2])
@pytest.mark.parametrize("rows_band", [1, 2, 3, 5])
def test_rc_forward_band_heights(stride, rows_band):
    """every band split of the plane (halo rows re-derived per band) gives the same bits"""
    y2, y2r, st, str_ = _run(3, 18, 32, 64, 128, stride, rows_band=rows_band)
    assert torch.equal(y2.view(torch.int16), y2r.view(torch.int16))
    assert float(((st - str_).abs() / (str_.abs() + 1e-3)).max()) < 1e-5


def test_rc_forward_unrounded_y1_is_closer_than_bf16():
    """round_y1 = 0 keeps the rebuilt y1 in fp32: the result differs from the stored-bf16-y1 path by bf16 rounding of y1 only"""
    y2, y2r, _, _ = _run(3, 18, 32, 64, 128, 1, round_y1=0)
    err = float((y2.float() - y2r.float()).norm() / y2r.float().norm())
    assert 0 < err < 1e-2


def test_rc_unsupported_configuration_is_refused():
    assert L.lib.dwn_dw_spatial_rc_supported(L.DWN_BF16, 64, 448, 3, 1, 18, 32) == 1
    assert L.lib.dwn_dw_spatial_rc_supported(L.DWN_F32, 64, 448, 3, 1, 18, 32) == 0
    assert L.lib.dwn_dw_spatial_rc_supported(L.DWN_BF16, 256, 1792, 3, 1, 5, 8) == 0
    assert L.lib.dwn_dw_spatial_rc_supported(L.DWN_BF16, 64, 440, 3, 1, 18, 32) == 0
    # planes whose one-row tile fits the LDS but neither register geometry: refused by the SAME predicate the launcher uses
    assert L.lib.dwn_dw_spatial_rc_supported(L.DWN_BF16, 64, 448, 3, 1, 18, 130) == 0
    assert L.lib.dwn_dw_spatial_rc_supported(L.DWN_BF16, 64, 448, 3, 2, 18, 160) == 0
    assert L.lib.dwn_dw_spatial_rc_supported(L.DWN_BF16, 64, 448, 3, 1, 18, 64) == 1
    ra = L.DwSpatialRcFwdArgs()
    dummy = torch.zeros(64, dtype=torch.uint8, device=dev())
    ra.a0 = dummy.data_ptr(); ra.blob = dummy.data_ptr(); ra.out = dummy.data_ptr(); ra.a0_ld = 256
    ra.planes = 1; ra.Hin = 5; ra.Win = 8; ra.Hout = 5; ra.Wout = 8; ra.Cin = 256; ra.E = 1792; ra.stride = 1
    rc = L.lib.dwn_dw_spatial_fwd_rc(C.byref(ra), dev().index, torch.cuda.current_stream().cuda_stream)
    assert rc == -3 and b"unsupported" in L.lib.dwn_last_error()
