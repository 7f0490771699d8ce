"""float64 reference of conv_pw -> BatchNorm-1 + SiLU -> spat_covn_dw and of its backward (reference ops:
src/models/dwiseneuro.py:90-102), channels-last, on the GPU, by padded-slice arithmetic and torch autograd — the checker the
kernel-level tests of the rebuilt-input stencils compare with (tests/test_gpu_dwfwd.py, test_gpu_dwbwd.py).  Test infrastructure
only."""
import torch


def conv_pw_f64(a0: torch.Tensor, w1: torch.Tensor) -> torch.Tensor:
    """y1 = a0 . W1^T in float64 from the (bf16) operands as given — the unrounded product."""
    return a0.double() @ w1.double().t()


def _dw3x3(z: torch.Tensor, w: torch.Tensor, stride: int) -> torch.Tensor:
    """z [P, H, W, C] float64, w [9, C] tap-major (dy * 3 + dx) -> [P, Hout, Wout, C]; zero padding 1."""
    P, H, W, Cc = z.shape
    Hout, Wout = (H - 1) // stride + 1, (W - 1) // stride + 1
    zp = torch.nn.functional.pad(z, (0, 0, 1, 1, 1, 1))
    out = None
    for dy in range(3):
        for dx in range(3):
            t = zp[:, dy:dy + stride * (Hout - 1) + 1:stride, dx:dx + stride * (Wout - 1) + 1:stride, :] * w[dy * 3 + dx]
            out = t if out is None else out + t
    return out


def dw_spatial_fwd_f64(y1, scale, shift, w, planes, Hin, Win, stride):
    """y1 [planes*Hin*Win, C] -> y2 [planes*Hout*Wout, C] (float64): dwS * SiLU(scale * y1 + shift)."""
    Cc = y1.shape[1]
    h = y1.double().view(planes, Hin, Win, Cc) * scale.double() + shift.double()
    y2 = _dw3x3(h * torch.sigmoid(h), w.double(), stride)
    return y2.reshape(-1, Cc)


def dw_spatial_bwd_f64(y1, scale, shift, mean, invstd, g, w, planes, Hin, Win, stride):
    """Backward of the above for the output gradient g [planes*Hout*Wout, C]: returns (dh1 = dL/d(BN1 output) [rows, C],
    dW [C, 9], sum dh1 [C], sum dh1 * yhat1 [C]) in float64, yhat1 = (y1 - mean) * invstd."""
    Cc = y1.shape[1]
    Hout, Wout = (Hin - 1) // stride + 1, (Win - 1) // stride + 1
    y1 = y1.double()
    h = (y1.view(planes, Hin, Win, Cc) * scale.double() + shift.double()).requires_grad_(True)
    wd = w.double().clone().requires_grad_(True)
    y2 = _dw3x3(h * torch.sigmoid(h), wd, stride)
    y2.backward(g.double().view(planes, Hout, Wout, Cc))
    dh1 = h.grad.reshape(-1, Cc)
    yhat = (y1 - mean.double()) * invstd.double()
    return dh1, wd.grad.t().contiguous(), dh1.sum(0), (dh1 * yhat).sum(0)


def rel_l2(a: torch.Tensor, b: torch.Tensor) -> float:
    return float((a.double() - b.double()).norm() / b.double().norm().clamp_min(1e-300))
