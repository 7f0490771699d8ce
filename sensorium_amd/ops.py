"""torch.autograd.Function wrappers over the C-ABI composites of libdwiseneuro_hip.so.

One Function per reference module on the hot path (src/models/dwiseneuro.py): stem, [PositionalEncoding3d +
InvertedResidual3d], pool, ShuffleLayer, Readout, and MicePoissonLoss (src/losses.py).  PyTorch is plumbing
here: it owns device memory (every buffer, saved tensor and workspace is a torch tensor), the stream, and the
autograd graph; all arithmetic runs in the hand-written HIP kernels.  There is no fallback path: tensors must
live on a GPU and the shared library must be present.
"""
from __future__ import annotations

import ctypes as C
import math
import os
from typing import Optional

import numpy as np
import torch

from . import _lib as L

_DT = {torch.float32: L.DWN_F32, torch.bfloat16: L.DWN_BF16}


def _ptr(t: Optional[torch.Tensor]):
    return None if t is None else t.data_ptr()


def _stream(device: torch.device):
    return torch.cuda.current_stream(device).cuda_stream


def _require_gpu(t: torch.Tensor, what: str):
    if not t.is_cuda:
        raise RuntimeError(f"sensorium_amd.{what}: tensors must be on a GPU (got {t.device}); "
                           "the HIP path has no CPU fallback")


def _bn_struct(bn: torch.nn.modules.batchnorm._BatchNorm, coef: torch.Tensor, dgamma=None, dbeta=None) -> L.BN:
    s = L.BN()
    s.gamma = bn.weight.data_ptr()
    s.beta = bn.bias.data_ptr()
    s.running_mean = bn.running_mean.data_ptr()
    s.running_var = bn.running_var.data_ptr()
    s.num_batches_tracked = bn.num_batches_tracked.data_ptr()
    s.coef = coef.data_ptr()
    s.dgamma = _ptr(dgamma)
    s.dbeta = _ptr(dbeta)
    return s


def _check_bn(bn):
    if bn.momentum is None or not bn.affine or not bn.track_running_stats:
        raise RuntimeError("sensorium_amd: BatchNorm must be affine with running stats and a fixed momentum")


def _f32_products(mod, inference_readout: bool = False) -> int:
    """How this module's fp32 GEMMs multiply (include/dwn.h DWN_F32_*).  ``DwiseNeuro.set_fp32_eval_products`` stamps the
    choice on its sub-modules: "bf16x3" (default: eval-mode forward as three bf16 products, 5.8e-7 from native; training always
    native) or "native" (fp32 MFMA everywhere)."""
    native = getattr(mod, "_dwn_fp32_native", False)
    if inference_readout:               # the readout's C struct has no training flag: the caller says what this forward is
        return L.F32_NATIVE if native else L.F32_SPLIT3
    return L.F32_NATIVE if native else L.F32_AUTO


def _ddp_flush() -> None:
    from . import ddp
    if ddp._LIVE:
        ddp.flush_ready_all()


def _ws(nbytes: int, device) -> torch.Tensor:
    return torch.empty(int(nbytes), dtype=torch.uint8, device=device)


# conv_pwl backward implementation forced for a process (the block / model parity tests are re-run under both): "new" =
# per-sample products + recompute epilogue, "old" = materialised du; unset = the library chooses by shape
def _env_choice(name: str, table: dict) -> int:
    v = os.environ.get(name, "")
    if v not in table:
        raise ValueError(f"{name}={v!r}: expected one of {sorted(k for k in table if k)} (or unset)")
    return table[v]


_PWL_BWD = _env_choice("DWN_PWL_BWD", {"": 0, "new": 1, "old": 2})
# dwn_block_args.y1_mode: "" = the library leaves y1 (conv_pw's output) unmaterialised in bf16 training where both stencils rebuild it
# from the block input; DWN_Y1=materialise forces the stored-y1 path (the parity tests run both; same-box A/B runs)
_Y1_MODE = _env_choice("DWN_Y1", {"": 0, "free": 0, "materialise": 1, "all": 2})


def grad_out(param: torch.Tensor, zero: bool = False) -> torch.Tensor:
    """The fp32 tensor a backward kernel writes ``param``'s gradient into.  Under data parallelism ``GradBuckets`` gives
    every parameter a slice of a flat all-reduce bucket (``param._dwn_grad_slot``): the gradient is produced *there* — a
    fresh view object, so autograd adopts it as ``param.grad`` without a copy and the bucket needs no gather pass.  When a
    gradient is already being accumulated (``param.grad`` set: argus ``iter_size`` > 1) or no bucket exists, a new tensor in
    the parameter's own shape (same memory layout as the kernels' 2-D views, so autograd can take ownership)."""
    slot = getattr(param, "_dwn_grad_slot", None)
    if slot is not None and param.grad is None and not getattr(param, "_dwn_slot_out", False):
        # handed out ONCE per backward pass: a second producer of the same parameter's gradient (a module applied twice, two
        # forwards summed into one loss) gets a tensor of its own — two kernels writing the same slot before AccumulateGrad
        # adds them would leave 2x the last contribution.  GradBuckets' hook clears the mark when the gradient has arrived.
        param._dwn_slot_out = True
        flat, off = slot
        g = flat[off:off + param.numel()].view(param.shape)
        return g.zero_() if zero else g
    if zero:
        return torch.zeros_like(param, dtype=torch.float32)
    return torch.empty_like(param, dtype=torch.float32)


# ------------------------------------------------------------------------------------------------
# index maps / positional-encoding tables (host side, cached by the modules)
# ------------------------------------------------------------------------------------------------
def nearest_src_index(out_size: int, in_size: int) -> np.ndarray:
    """F.interpolate(mode='nearest') source index (reference: dwiseneuro.py:127-129)."""
    scale = np.float32(in_size) / np.float32(out_size)
    src = np.floor(np.arange(out_size, dtype=np.float32) * scale).astype(np.int64)
    return np.minimum(src, in_size - 1).astype(np.int32)


def inverse_index(src: np.ndarray, in_size: int) -> np.ndarray:
    inv = np.full(in_size, -1, dtype=np.int32)
    inv[src] = np.arange(len(src), dtype=np.int32)
    if len(np.unique(src)) != len(src):
        raise RuntimeError("nearest-neighbour shortcut map is not injective (stride < 1?)")
    return inv


def pe_axis_tables(channels: int, inv_freq: torch.Tensor, t: int, h: int, w: int):
    """Separable positional-encoding tables PT[t][C], PH[h][C], PW[w][C] (reference: dwiseneuro.py:163-182):
    enc[c](t,h,w) = PT[t][c] + PH[h][c] + PW[w][c] where every channel depends on exactly one axis."""
    ch = int(math.ceil(channels / 6) * 2)
    if ch % 2:
        ch += 1
    inv = inv_freq.detach().float().cpu()
    tabs = []
    for axis, size in enumerate((t, h, w)):
        arg = inv[:, None] * torch.arange(size).float()[None, :]
        emb = torch.cat([arg.sin(), arg.cos()], dim=0)              # [ch, size]
        tab = torch.zeros(size, channels, dtype=torch.float32)
        lo = axis * ch
        n = max(0, min(ch, channels - lo))
        if n > 0:
            tab[:, lo:lo + n] = emb[:n].t()
        tabs.append(tab.contiguous())
    return tabs


# ------------------------------------------------------------------------------------------------
# stem
# ------------------------------------------------------------------------------------------------
class StemFn(torch.autograd.Function):
    """Conv3d(C_in->C0, 1x1x1, bias=False) + BatchNorm3d on the NCDHW fp32 input (dwiseneuro.py:306-309).
    Returns the channels-last activation [B,T,H,W,C0] in the compute dtype."""

    @staticmethod
    def forward(ctx, x, weight, gamma, beta, mod, dtype, pe=None):
        _require_gpu(x, "StemFn")
        conv, bn = mod.stem[0], mod.stem[1].bn
        _check_bn(bn)
        x = x.contiguous().float()
        B, Cin, T, H, W = x.shape
        C0 = weight.shape[0]
        dev = x.device
        out = torch.empty(B, T, H, W, C0, dtype=dtype, device=dev)
        coef = torch.empty(4 * C0, dtype=torch.float32, device=dev)
        # the input moments (sum x, sum x x^T): what the backward needs instead of the raw conv output (never materialised)
        xmom = torch.empty(72, dtype=torch.float64, device=dev)
        a = L.StemArgs()
        a.dtype = _DT[dtype]; a.training = int(bn.training); a.B = B; a.Cin = Cin; a.C0 = C0; a.S = T * H * W
        a.eps = bn.eps; a.momentum = bn.momentum
        a.x = x.data_ptr(); a.w = weight.data_ptr(); a.bn = _bn_struct(bn, coef)
        a.out = out.data_ptr(); a.xmom = xmom.data_ptr()
        if pe is not None:       # positional encoding of the first block, folded into the stem's output pass
            a.pe_t, a.pe_h, a.pe_w = (t.data_ptr() for t in pe)
        a.T, a.H, a.W = T, H, W
        ws = _ws(L.lib.dwn_stem_workspace_bytes(C.byref(a)), dev)
        a.ws = ws.data_ptr(); a.ws_bytes = ws.numel()
        L.check(L.lib.dwn_stem_forward(C.byref(a), dev.index, _stream(dev)), "dwn_stem_forward")
        ctx.mod = mod; ctx.dtype = dtype; ctx.was_training = bn.training
        ctx.save_for_backward(x, weight, xmom, coef)
        return out

    @staticmethod
    def backward(ctx, dout):
        x, weight, xmom, coef = ctx.saved_tensors
        if not ctx.was_training:
            raise RuntimeError("sensorium_amd: backward through eval-mode BatchNorm is not built")
        bn = ctx.mod.stem[1].bn
        dev = x.device
        B, Cin, T, H, W = x.shape
        C0 = weight.shape[0]
        dout = dout.contiguous()
        dw = grad_out(ctx.mod.stem[0].weight)
        dgamma = grad_out(bn.weight)
        dbeta = grad_out(bn.bias)
        a = L.StemArgs()
        a.dtype = _DT[ctx.dtype]; a.training = 1; a.B = B; a.Cin = Cin; a.C0 = C0; a.S = T * H * W
        a.eps = bn.eps; a.momentum = bn.momentum
        a.x = x.data_ptr(); a.w = weight.data_ptr(); a.bn = _bn_struct(bn, coef, dgamma, dbeta)
        a.xmom = xmom.data_ptr(); a.dout = dout.data_ptr(); a.dw = dw.data_ptr()
        ws = _ws(L.lib.dwn_stem_workspace_bytes(C.byref(a)), dev)
        a.ws = ws.data_ptr(); a.ws_bytes = ws.numel()
        L.check(L.lib.dwn_stem_backward(C.byref(a), dev.index, _stream(dev)), "dwn_stem_backward")
        return None, dw, dgamma, dbeta, None, None, None


# ------------------------------------------------------------------------------------------------
# PositionalEncoding3d + InvertedResidual3d
# ------------------------------------------------------------------------------------------------
_BLOCK_PARAMS = ("w_pw", "g1", "b1", "w_dws", "g2", "b2", "w_dwt", "g3", "b3", "se_wr", "se_br", "se_we", "se_be",
                 "w_pwl", "g4", "b4", "gsc", "bsc")


def _block_args(blk, geom, x, dtype, training, coefs, saved, drop_scale, x_has_pe, a0):
    """Fill the fields shared by forward and backward."""
    B, T, Hin, Win, Cin = x.shape
    a = L.BlockArgs()
    a.dtype = _DT[dtype]; a.training = int(training)
    a.B = B; a.T = T; a.Hin = Hin; a.Win = Win
    a.Hout = (Hin - 1) // blk.spatial_stride + 1
    a.Wout = (Win - 1) // blk.spatial_stride + 1
    a.Cin = Cin; a.Cmid = blk.mid_features; a.Cout = blk.out_features
    a.stride = blk.spatial_stride; a.ks = blk.spatial_kernel; a.kt = blk.temporal_kernel
    a.se_r = blk.se.conv_reduce.out_channels
    bn1 = blk.conv_pw[1].bn
    a.eps = bn1.eps; a.momentum = bn1.momentum
    a.x = x.data_ptr()
    a.x_has_pe = int(x_has_pe)
    a.a0 = _ptr(a0)
    pe_t, pe_h, pe_w, hsrc, wsrc, hinv, winv = geom
    a.pe_t = pe_t.data_ptr(); a.pe_h = pe_h.data_ptr(); a.pe_w = pe_w.data_ptr()
    a.hsrc = hsrc.data_ptr(); a.wsrc = wsrc.data_ptr(); a.hinv = hinv.data_ptr(); a.winv = winv.data_ptr()
    a.w_pw = blk.conv_pw[0].weight.data_ptr()
    a.w_dws = blk.spat_covn_dw[0].weight.data_ptr()
    a.w_dwt = blk.temp_covn_dw[0].weight.data_ptr()
    a.w_pwl = blk.conv_pwl[0].weight.data_ptr()
    a.se_wr = blk.se.conv_reduce.weight.data_ptr(); a.se_br = blk.se.conv_reduce.bias.data_ptr()
    a.se_we = blk.se.conv_expand.weight.data_ptr(); a.se_be = blk.se.conv_expand.bias.data_ptr()
    a.drop_scale = _ptr(drop_scale)
    a.se_pmean = saved["pmean"].data_ptr(); a.se_hidpre = saved["hidpre"].data_ptr()
    a.se_gate = saved["gate"].data_ptr()
    a.f32_products = _f32_products(blk)
    a.y1_mode = getattr(blk, "_dwn_y1_mode", _Y1_MODE)
    return a


class BlockFn(torch.autograd.Function):
    """x -> InvertedResidual3d(x + PositionalEncoding3d) (dwiseneuro.py:136-144, 184-192), channels-last."""

    @staticmethod
    def forward(ctx, x, drop_scale, blk, geom, dtype, x_has_pe, out_pe, *params):
        _require_gpu(x, "BlockFn")
        x = x.contiguous()
        dev = x.device
        B, T, Hin, Win, Cin = x.shape
        s = blk.spatial_stride
        Hout, Wout = (Hin - 1) // s + 1, (Win - 1) // s + 1
        Cmid, Cout = blk.mid_features, blk.out_features
        bns = blk.bn_modules()
        for bn in bns:
            _check_bn(bn)
        training = bns[0].training
        f32 = dict(dtype=torch.float32, device=dev)
        y2 = torch.empty(B, T, Hout, Wout, Cmid, dtype=dtype, device=dev)
        z3 = torch.empty_like(y2)
        a0 = None if x_has_pe else torch.empty_like(x)
        y4 = torch.empty(B, T, Hout, Wout, Cout, dtype=dtype, device=dev)
        out = torch.empty_like(y4)
        coefs = [torch.empty(4 * c, **f32) for c in (Cmid, Cmid, Cmid, Cout, Cout)]
        R = blk.se.conv_reduce.out_channels
        saved = dict(pmean=torch.empty(B, Cmid, **f32), hidpre=torch.empty(B, R, **f32),
                     gate=torch.empty(B, Cmid, **f32))
        a = _block_args(blk, geom, x, dtype, training, coefs, saved, drop_scale, x_has_pe, a0)
        a.out = out.data_ptr()
        # y1 is not written where the stencils rebuild it (eval; bf16 training where both directions are built: the backward then
        # gets no y1 either), y3 not where the eval-mode temporal pass emits z3 directly
        writes = L.lib.dwn_block_forward_writes(C.byref(a))
        y1 = torch.empty(B, T, Hin, Win, Cmid, dtype=dtype, device=dev) if writes & 1 else None
        y3 = torch.empty_like(y2) if writes & 2 else None
        a.y1 = _ptr(y1); a.y2 = y2.data_ptr(); a.y3 = _ptr(y3); a.y4 = y4.data_ptr()
        a.z3 = z3.data_ptr()
        if out_pe is not None:   # next block's positional encoding, folded into this block's residual pass
            a.out_pe_t, a.out_pe_h, a.out_pe_w = (t.data_ptr() for t in out_pe)
        a.bn1, a.bn2, a.bn3, a.bn4, a.bnsc = (_bn_struct(bn, cf) for bn, cf in zip(bns, coefs))
        ws = _ws(L.lib.dwn_block_workspace_bytes(C.byref(a), 0), dev)
        a.ws = ws.data_ptr(); a.ws_bytes = ws.numel()
        L.check(L.lib.dwn_block_forward(C.byref(a), dev.index, _stream(dev)), "dwn_block_forward")
        ctx.blk = blk; ctx.geom = geom; ctx.dtype = dtype; ctx.was_training = training
        if getattr(blk, "_capture", False):        # test hook: expose the raw intermediates
            blk._captured = dict(y1=y1, y2=y2, y3=y3, y4=y4, z3=z3, coefs=coefs, **saved)
        ctx.has_drop = drop_scale is not None
        # the block input *including* its positional encoding is what backward needs
        if not training:              # no backward through eval-mode BatchNorm: nothing to keep
            return out
        tensors = [x if x_has_pe else a0, y1, y2, y3, y4, *coefs, saved["pmean"], saved["hidpre"], saved["gate"], z3]
        if drop_scale is not None:
            tensors.append(drop_scale)
        ctx.save_for_backward(*tensors)
        return out

    @staticmethod
    def backward(ctx, dout):
        if not ctx.was_training:
            raise RuntimeError("sensorium_amd: backward through eval-mode BatchNorm is not built")
        blk, dtype = ctx.blk, ctx.dtype
        t = ctx.saved_tensors
        x, y1, y2, y3, y4 = t[:5]
        coefs = list(t[5:10])
        saved = dict(pmean=t[10], hidpre=t[11], gate=t[12])
        z3 = t[13]
        drop_scale = t[14] if ctx.has_drop else None
        dev = x.device
        dout = dout.contiguous()
        B, T, Hin, Win, Cin = x.shape
        Hout, Wout = y2.shape[2], y2.shape[3]
        Cmid, Cout = blk.mid_features, blk.out_features
        f32 = dict(dtype=torch.float32, device=dev)
        a = _block_args(blk, ctx.geom, x, dtype, True, coefs, saved, drop_scale, True, None)
        a.y1 = _ptr(y1); a.y2 = y2.data_ptr(); a.y3 = y3.data_ptr(); a.y4 = y4.data_ptr()      # (y1 is None on a y1-free block)
        a.z3 = z3.data_ptr()
        bns = blk.bn_modules()
        dg = [grad_out(bn.weight) for bn in bns]
        db = [grad_out(bn.bias) for bn in bns]
        a.bn1, a.bn2, a.bn3, a.bn4, a.bnsc = (_bn_struct(bn, cf, g_, b_) for bn, cf, g_, b_ in zip(bns, coefs, dg, db))
        m_in, m_out = B * T * Hin * Win, B * T * Hout * Wout
        buf_a = torch.empty(max(m_in, m_out) * Cmid, dtype=dtype, device=dev)
        buf_b = torch.empty(m_out * Cmid, dtype=dtype, device=dev)
        dy4 = torch.empty(m_out * Cout, dtype=dtype, device=dev)
        da0 = torch.empty(m_in * Cin, dtype=dtype, device=dev)
        dx = torch.empty_like(x)
        R = blk.se.conv_reduce.out_channels
        # gradients are allocated in the parameters' own shapes (same memory layout as the kernels' 2-D views) so
        # that autograd can take ownership instead of cloning a view
        # (dwn_block_backward clears the four atomically accumulated weight gradients itself, in its prep launch)
        # ... or, under data parallelism, straight into the parameters' slices of the all-reduce buckets (grad_out)
        dw_pw = grad_out(blk.conv_pw[0].weight)
        dw_dws = grad_out(blk.spat_covn_dw[0].weight)
        dw_dwt = grad_out(blk.temp_covn_dw[0].weight)
        dw_pwl = grad_out(blk.conv_pwl[0].weight)
        dse_wr = grad_out(blk.se.conv_reduce.weight); dse_br = grad_out(blk.se.conv_reduce.bias)
        dse_we = grad_out(blk.se.conv_expand.weight); dse_be = grad_out(blk.se.conv_expand.bias)
        a.dout = dout.data_ptr(); a.dx = dx.data_ptr()
        a.buf_a = buf_a.data_ptr(); a.buf_b = buf_b.data_ptr(); a.dy4 = dy4.data_ptr(); a.da0 = da0.data_ptr()
        a.dw_pw = dw_pw.data_ptr(); a.dw_dws = dw_dws.data_ptr(); a.dw_dwt = dw_dwt.data_ptr()
        a.dw_pwl = dw_pwl.data_ptr()
        a.dse_wr = dse_wr.data_ptr(); a.dse_br = dse_br.data_ptr(); a.dse_we = dse_we.data_ptr()
        a.dse_be = dse_be.data_ptr()
        a.pwl_bwd = _PWL_BWD                       # before the workspace is sized: the per-sample path carves B x Cout x Cmid floats
        ws = _ws(L.lib.dwn_block_workspace_bytes(C.byref(a), 1), dev)
        a.ws = ws.data_ptr(); a.ws_bytes = ws.numel()
        L.check(L.lib.dwn_block_backward(C.byref(a), dev.index, _stream(dev)), "dwn_block_backward")
        _ddp_flush()            # this block's kernels are queued: a good moment for the host to start pending gradient exchanges
        grads = (dw_pw, dg[0], db[0], dw_dws, dg[1], db[1], dw_dwt, dg[2], db[2], dse_wr, dse_br, dse_we, dse_be,
                 dw_pwl, dg[3], db[3], dg[4], db[4])
        return (dx, None, None, None, None, None, None) + grads


# ------------------------------------------------------------------------------------------------
# pool
# ------------------------------------------------------------------------------------------------
class PoolFn(torch.autograd.Function):
    """AdaptiveAvgPool3d((None,1,1)) + squeeze (dwiseneuro.py:374,400): [B,T,H,W,C] -> [B,T,C]."""

    @staticmethod
    def forward(ctx, x):
        _require_gpu(x, "PoolFn")
        x = x.contiguous()
        B, T, H, W, Cc = x.shape
        out = torch.empty(B, T, Cc, dtype=x.dtype, device=x.device)
        a = L.PoolArgs()
        a.dtype = _DT[x.dtype]; a.BT = B * T; a.HW = H * W; a.C = Cc; a.x = x.data_ptr(); a.out = out.data_ptr()
        L.check(L.lib.dwn_pool_forward(C.byref(a), x.device.index, _stream(x.device)), "dwn_pool_forward")
        ctx.shape = (B, T, H, W, Cc)
        return out

    @staticmethod
    def backward(ctx, dout):
        B, T, H, W, Cc = ctx.shape
        dout = dout.contiguous()
        dx = torch.empty(B, T, H, W, Cc, dtype=dout.dtype, device=dout.device)
        a = L.PoolArgs()
        a.dtype = _DT[dout.dtype]; a.BT = B * T; a.HW = H * W; a.C = Cc; a.dout = dout.data_ptr(); a.dx = dx.data_ptr()
        L.check(L.lib.dwn_pool_backward(C.byref(a), dout.device.index, _stream(dout.device)), "dwn_pool_backward")
        return dx


# ------------------------------------------------------------------------------------------------
# cortex ShuffleLayer
# ------------------------------------------------------------------------------------------------
class CortexFn(torch.autograd.Function):
    """ShuffleLayer.forward (dwiseneuro.py:228-234) on [B,T,C_in] -> [B,T,C]."""

    @staticmethod
    def forward(ctx, x, drop_scale, layer, dtype, weight, g, b, gsc, bsc):
        _require_gpu(x, "CortexFn")
        x = x.contiguous()
        dev = x.device
        B, T, Cin = x.shape
        Cc = layer.out_features
        bn, bnsc = layer.bn.bn, layer.bn_sc.bn
        _check_bn(bn); _check_bn(bnsc)
        training = bn.training
        y = torch.empty(B, T, Cc, dtype=dtype, device=dev)
        out = torch.empty_like(y)
        coef = torch.empty(4 * Cc, dtype=torch.float32, device=dev)
        coefsc = torch.empty(4 * Cc, dtype=torch.float32, device=dev)
        a = L.CortexArgs()
        a.dtype = _DT[dtype]; a.training = int(training); a.B = B; a.T = T; a.Cin = Cin; a.C = Cc
        a.groups = layer.groups; a.eps = bn.eps; a.momentum = bn.momentum
        a.x = x.data_ptr(); a.out = out.data_ptr(); a.y = y.data_ptr(); a.w = weight.data_ptr()
        a.bn = _bn_struct(bn, coef); a.bnsc = _bn_struct(bnsc, coefsc); a.drop_scale = _ptr(drop_scale)
        a.f32_products = _f32_products(layer)
        ws = _ws(L.lib.dwn_cortex_workspace_bytes(C.byref(a), 0), dev)
        a.ws = ws.data_ptr(); a.ws_bytes = ws.numel()
        L.check(L.lib.dwn_cortex_forward(C.byref(a), dev.index, _stream(dev)), "dwn_cortex_forward")
        ctx.layer = layer; ctx.dtype = dtype; ctx.was_training = training; ctx.has_drop = drop_scale is not None
        tensors = [x, y, coef, coefsc, weight]
        if drop_scale is not None:
            tensors.append(drop_scale)
        ctx.save_for_backward(*tensors)
        return out

    @staticmethod
    def backward(ctx, dout):
        if not ctx.was_training:
            raise RuntimeError("sensorium_amd: backward through eval-mode BatchNorm is not built")
        layer, dtype = ctx.layer, ctx.dtype
        t = ctx.saved_tensors
        x, y, coef, coefsc, weight = t[:5]
        drop_scale = t[5] if ctx.has_drop else None
        dev = x.device
        dout = dout.contiguous()
        B, T, Cin = x.shape
        Cc = layer.out_features
        bn, bnsc = layer.bn.bn, layer.bn_sc.bn
        f32 = dict(dtype=torch.float32, device=dev)
        dgm, dbm, dgs, dbs = grad_out(bn.weight), grad_out(bn.bias), grad_out(bnsc.weight), grad_out(bnsc.bias)
        dw = grad_out(layer.conv.weight)             # cleared by dwn_cortex_backward's prep launch
        dx = torch.empty_like(x)
        a = L.CortexArgs()
        a.dtype = _DT[dtype]; a.training = 1; a.B = B; a.T = T; a.Cin = Cin; a.C = Cc
        a.groups = layer.groups; a.eps = bn.eps; a.momentum = bn.momentum
        a.x = x.data_ptr(); a.y = y.data_ptr(); a.w = weight.data_ptr()
        a.bn = _bn_struct(bn, coef, dgm, dbm); a.bnsc = _bn_struct(bnsc, coefsc, dgs, dbs)
        a.drop_scale = _ptr(drop_scale)
        a.dout = dout.data_ptr(); a.dx = dx.data_ptr(); a.dw = dw.data_ptr()
        ws = _ws(L.lib.dwn_cortex_workspace_bytes(C.byref(a), 1), dev)
        a.ws = ws.data_ptr(); a.ws_bytes = ws.numel()
        L.check(L.lib.dwn_cortex_backward(C.byref(a), dev.index, _stream(dev)), "dwn_cortex_backward")
        return dx, None, None, None, dw, dgm, dbm, dgs, dbs


# ------------------------------------------------------------------------------------------------
# readout
# ------------------------------------------------------------------------------------------------
class ReadoutFn(torch.autograd.Function):
    """Readout.forward (dwiseneuro.py:283-287): Dropout1d -> grouped Conv1d(k=1)+bias -> [:N] -> Softplus(beta).
    x: [B,T,C] (compute dtype) -> [B,N,T] fp32."""

    @staticmethod
    def forward(ctx, x, drop_mask, mod, weight, bias):
        _require_gpu(x, "ReadoutFn")
        x = x.contiguous()
        dev = x.device
        B, T, Cin = x.shape
        n = mod.out_features
        out = torch.empty(B, n, T, dtype=torch.float32, device=dev)
        a = L.ReadoutArgs()
        a.dtype = _DT[x.dtype]; a.B = B; a.T = T; a.Cin = Cin; a.groups = mod.groups; a.n_out = n
        a.softplus_beta = mod.softplus_beta
        a.x = x.data_ptr(); a.w = weight.data_ptr(); a.bias = bias.data_ptr(); a.drop_mask = _ptr(drop_mask)
        a.out = out.data_ptr()
        # eval-mode forward (the module is not training; a no_grad forward of a training module keeps the training numerics)
        a.f32_products = _f32_products(mod, inference_readout=True) if not mod.training else L.F32_NATIVE
        # a backward will follow: the pack pass also writes the weight in the data gradient's layout and backward reuses it
        # (the optimizer only touches the weight after backward)
        wt = None
        if any(ctx.needs_input_grad):
            wt = torch.empty(L.lib.dwn_readout_wt_bytes(C.byref(a)), dtype=torch.uint8, device=dev)
            a.wt = wt.data_ptr()
        ws = _ws(L.lib.dwn_readout_workspace_bytes(C.byref(a), 0), dev)
        a.ws = ws.data_ptr(); a.ws_bytes = ws.numel()
        L.check(L.lib.dwn_readout_forward(C.byref(a), dev.index, _stream(dev)), "dwn_readout_forward")
        ctx.mod = mod; ctx.has_mask = drop_mask is not None
        ctx.wt = wt
        # samples whose loss weight for this mouse is not zero (MouseModel.train_step knows them on the host): the backward
        # then runs on those rows only — the gradient w.r.t. the other rows' predictions is exactly zero (losses.py:15-17)
        ctx.active = getattr(mod, "_dwn_active", None)
        tensors = [x, weight, bias, out]
        if drop_mask is not None:
            tensors.append(drop_mask)
        ctx.save_for_backward(*tensors)
        return out

    @staticmethod
    def backward(ctx, dout):
        mod = ctx.mod
        t = ctx.saved_tensors
        x, weight, bias, out = t[:4]
        drop_mask = t[4] if ctx.has_mask else None
        dev = x.device
        dout = dout.contiguous().float()
        B, T, Cin = x.shape
        n = mod.out_features
        conv = mod.layer[1]
        dw = grad_out(conv.weight)                   # overwritten by dwn_readout_backward (no 64 MB clear per readout)
        db = grad_out(conv.bias, zero=True)
        idx, dx_full = ctx.active, None
        if idx is not None and idx.numel() < B:
            # compact backward: the three products see nb x T rows instead of B x T (ten readouts with one-hot mouse
            # weights: 3-4 of 32 samples each); zero rows contribute exactly nothing to dW / dbias and get dx = 0
            dx_full = torch.zeros_like(x)
            if idx.numel() == 0:
                return dx_full, None, None, dw.zero_(), db
            x, out, dout = x.index_select(0, idx), out.index_select(0, idx), dout.index_select(0, idx)
            if drop_mask is not None:
                drop_mask = drop_mask.index_select(0, idx)
            B = idx.numel()
        dx = torch.empty_like(x)
        a = L.ReadoutArgs()
        a.dtype = _DT[x.dtype]; a.B = B; a.T = T; a.Cin = Cin; a.groups = mod.groups; a.n_out = n
        a.softplus_beta = mod.softplus_beta
        a.x = x.data_ptr(); a.w = weight.data_ptr(); a.bias = bias.data_ptr(); a.drop_mask = _ptr(drop_mask)
        a.out = out.data_ptr(); a.dout = dout.data_ptr(); a.dx = dx.data_ptr(); a.dw = dw.data_ptr()
        a.dbias = db.data_ptr()
        if ctx.wt is not None:
            a.wt = ctx.wt.data_ptr()
        ws = _ws(L.lib.dwn_readout_workspace_bytes(C.byref(a), 1), dev)
        a.ws = ws.data_ptr(); a.ws_bytes = ws.numel()
        L.check(L.lib.dwn_readout_backward(C.byref(a), dev.index, _stream(dev)), "dwn_readout_backward")
        if dx_full is not None:
            dx_full.index_copy_(0, idx, dx)
            dx = dx_full
        return dx, None, None, dw, db


# ------------------------------------------------------------------------------------------------
# MicePoissonLoss
# ------------------------------------------------------------------------------------------------
class PoissonLossFn(torch.autograd.Function):
    """sum_{b,n,t} w[b] * (x - y*log(x + eps)) for one mouse (losses.py:14-20); w already normalised.
    The reduction runs in a double accumulator on the device and is returned as an fp32 scalar."""

    @staticmethod
    def forward(ctx, pred, target, w, eps):
        _require_gpu(pred, "PoissonLossFn")
        pred = pred.contiguous().float()
        target = target.contiguous().float()
        w = w.contiguous().float()
        dev = pred.device
        acc = torch.zeros(1, dtype=torch.float64, device=dev)
        out = torch.empty((), dtype=torch.float32, device=dev)
        per_sample = pred[0].numel()
        L.check(L.lib.dwn_poisson_loss_forward(pred.data_ptr(), target.data_ptr(), w.data_ptr(), per_sample,
                                               pred.numel(), eps, acc.data_ptr(), dev.index, _stream(dev)),
                "dwn_poisson_loss_forward")
        L.check(L.lib.dwn_f64_to_f32(acc.data_ptr(), out.data_ptr(), 1, dev.index, _stream(dev)), "dwn_f64_to_f32")
        ctx.eps = eps
        ctx.save_for_backward(pred, target, w)
        return out

    @staticmethod
    def backward(ctx, gout):
        pred, target, w = ctx.saved_tensors
        dev = pred.device
        gscale = gout.contiguous().float()
        dpred = torch.empty_like(pred)
        L.check(L.lib.dwn_poisson_loss_backward(pred.data_ptr(), target.data_ptr(), w.data_ptr(),
                                                gscale.data_ptr(), pred[0].numel(), pred.numel(), ctx.eps,
                                                dpred.data_ptr(), dev.index, _stream(dev)),
                "dwn_poisson_loss_backward")
        return dpred, None, None, None
