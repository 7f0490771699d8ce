"""temp_covn_dw with the eval-mode epilogue (dwn_dw_temporal_fwd, z_scale / z_shift / pooled): z3 = SiLU(BN3(y3)) and the
SqueezeExcite pooling sums (reference src/models/dwiseneuro.py:105-111, 9-22, 38-39) straight from the temporal pass, against
the two-pass form (plain temporal forward, then BatchNorm + SiLU + per-sample sums in torch on the stored y3)."""
import ctypes as C

import pytest
import torch

pytestmark = pytest.mark.gpu

import sensorium_amd._lib as L  # noqa: E402
from tests.gpu_helpers import dev  # noqa: E402


def _desc(p, ld, **kw):
    d = L.LoadDesc()
    d.p = p.data_ptr(); d.ld = ld; d.rows_per_sample = 1
    for k, v in kw.items():
        setattr(d, k, v.data_ptr() if isinstance(v, torch.Tensor) else v)
    return d


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32])
@pytest.mark.parametrize("B,T,HW,Cc,kt", [(3, 8, 40, 64, 5), (2, 6, 35, 24, 5), (4, 5, 9, 128, 3), (2, 16, 144, 192, 5)])
def test_temporal_forward_eval_epilogue(dtype, B, T, HW, Cc, kt):
    g = torch.Generator(device=dev()).manual_seed(B * 1000 + HW)
    M = B * T * HW
    y2 = torch.randn(M, Cc, device=dev(), generator=g).to(dtype)
    coef2 = torch.cat([torch.rand(Cc, device=dev(), generator=g) + 0.5, torch.randn(Cc, device=dev(), generator=g) * 0.3])
    coef3 = torch.cat([torch.rand(Cc, device=dev(), generator=g) + 0.5, torch.randn(Cc, device=dev(), generator=g) * 0.3])
    w = torch.randn(kt, Cc, device=dev(), generator=g) / kt
    dt = L.DWN_BF16 if dtype == torch.bfloat16 else L.DWN_F32
    s = torch.cuda.current_stream().cuda_stream

    def run(z):
        out = torch.full((M, Cc), float("nan"), device=dev()).to(dtype)
        pooled = torch.zeros(B, Cc, dtype=torch.int64, device=dev())      # 64-bit fixed point, units of 2^-32 (include/dwn.h)
        a = L.DwTemporalFwdArgs()
        a.inp = _desc(y2, Cc, v1=coef2, v2=coef2[Cc:], act=1)
        a.w = w.data_ptr(); a.out = out.data_ptr(); a.B = B; a.T = T; a.HW = HW; a.C = Cc; a.kt = kt
        if z:
            a.z_scale = coef3.data_ptr(); a.z_shift = coef3[Cc:].data_ptr(); a.pooled = pooled.data_ptr()
        L.check(L.lib.dwn_dw_temporal_fwd(C.byref(a), dt, 0, s), "dwt_fwd")
        torch.cuda.synchronize()
        return out, pooled

    y3, _ = run(False)
    z3, pooled_fix = run(True)
    _, again = run(True)
    assert torch.equal(pooled_fix, again), "integer pooling sums must not depend on the arrival order"
    pooled = pooled_fix.double() / 2.0 ** 32
    h = y3.float() * coef3[:Cc] + coef3[Cc:]
    want = (h * torch.sigmoid(h)).to(dtype)
    # same arithmetic on the same rounded y3; the kernel's sigmoid is exp + rcp (1 ulp-level differences)
    tol = 2e-2 if dtype == torch.bfloat16 else 2e-6
    assert not torch.isnan(z3.float()).any()
    assert float((z3.float() - want.float()).abs().max() / want.float().abs().max()) < tol
    want_pool = z3.double().view(B, T * HW, Cc).sum(1)           # sums of the values as stored
    assert float((pooled - want_pool).abs().max() / want_pool.abs().max()) < 1e-5
    # argument errors: statistics and the eval epilogue exclude each other
    a = L.DwTemporalFwdArgs()
    a.inp = _desc(y2, Cc, v1=coef2, v2=coef2[Cc:], act=1)
    a.w = w.data_ptr(); a.out = z3.data_ptr(); a.B = B; a.T = T; a.HW = HW; a.C = Cc; a.kt = kt
    a.z_scale = coef3.data_ptr()
    assert L.lib.dwn_dw_temporal_fwd(C.byref(a), dt, 0, s) < 0
