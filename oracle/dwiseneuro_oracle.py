"""CPU oracle for the DwiseNeuro hot path (TEST INFRASTRUCTURE — never the product path).

This file is a from-scratch *functional* restatement of the arithmetic of lRomul/sensorium's
``src/models/dwiseneuro.py`` / ``src/losses.py`` / ``src/ema.py`` / ``src/predictors.py`` /
``src/metrics.py``.  It is written in channels-last ("NDHWC": rows = (b, t, h, w), columns =
channels) index math with explicit stencils, explicit batch-norm, explicit index maps and a
closed-form positional encoding, i.e. it shares no structure with the reference's ``nn.Module``
tree; it consumes the reference's ``state_dict`` (same key names) so both can be driven by the same
weights.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import this
module.  ``sensorium_amd`` (the product) must never import it.

Pinning: ``oracle/make_golden.py`` (run in the build container, where ``/root/reference`` exists)
drives the *real* reference modules (loaded by file path) and this oracle with identical weights and
inputs, asserts agreement, and commits small fixtures under ``tests/golden/``.  The reference itself
has no tests / golden vectors, so those generated fixtures are the pin.

Every function cites the reference file:line it restates (paths relative to /root/reference).
All functions are dtype-generic (float32 for parity runs, float64 for "ground truth" checks) and
are differentiable through torch autograd, which is how oracle gradients are obtained.
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch

Tensor = torch.Tensor
BN_EPS = 1e-5          # torch.nn.BatchNorm{1,3}d default, used by dwiseneuro.py:16
BN_MOMENTUM = 0.1      # torch default


# --------------------------------------------------------------------------------------------
# index maps (bit-exact requirements)
# --------------------------------------------------------------------------------------------
def nearest_src_index(out_size: int, in_size: int) -> np.ndarray:
    """Source index of ``F.interpolate(mode="nearest")`` as used at dwiseneuro.py:127-129.

    ATen computes ``src = min(int(floorf(dst * scale)), in_size - 1)`` with
    ``scale = float(in_size) / out_size`` in float32 (9 -> 5 gives [0, 1, 3, 5, 7]).
    """
    scale = np.float32(in_size) / np.float32(out_size)
    dst = np.arange(out_size, dtype=np.float32)
    src = np.floor(dst * scale).astype(np.int64)
    return np.minimum(src, in_size - 1)


def strided_out_size(in_size: int, stride: int) -> int:
    """``ceil(in/stride)`` (dwiseneuro.py:128); equals the k=3,p=1 conv output size floor((in-1)/s)+1."""
    return -(-in_size // stride)


def tile_channel_index(out_channels: int, in_channels: int) -> np.ndarray:
    """``torch.tile(x, ceil(out/in))[:, :out]`` (dwiseneuro.py:130-132, 221-224): out[c] = in[c mod C_in]."""
    return np.arange(out_channels, dtype=np.int64) % in_channels


def shuffle_source_index(channels: int, groups: int) -> np.ndarray:
    """ChannelShuffle of dwiseneuro.py:212-219 (view(b,g,C/g,t) -> transpose(1,2) -> reshape):
    ``out[j] = in[(j mod g) * (C/g) + j div g]``."""
    j = np.arange(channels, dtype=np.int64)
    return (j % groups) * (channels // groups) + j // groups


def pe_num_channels(channels: int) -> int:
    """dwiseneuro.py:151-154: ch = 2*ceil(C/6), made even."""
    ch = int(math.ceil(channels / 6) * 2)
    if ch % 2:
        ch += 1
    return ch


def pe_inv_freq(channels: int) -> Tensor:
    """dwiseneuro.py:155: inv_freq_k = 10000^(-2k/ch), k < ch/2 (float32, as the registered buffer)."""
    ch = pe_num_channels(channels)
    return 1.0 / (10000 ** (torch.arange(0, ch, 2).float() / ch))


def pe_axis_tables(channels: int, t: int, h: int, w: int, inv_freq: Optional[Tensor] = None,
                   dtype=torch.float32) -> Tuple[Tensor, Tensor, Tensor]:
    """Separable form of the cached encoding of dwiseneuro.py:163-182.

    The reference concatenates, along channels, [sin(f*pos_T); cos(f*pos_T)], then the same for H,
    then W, and truncates to C channels.  Channel c therefore depends on exactly one axis, so
    ``enc[c, t, h, w] = PT[t, c] + PH[h, c] + PW[w, c]`` with two of the three terms exactly 0.
    Returns (PT [t, C], PH [h, C], PW [w, C]).
    """
    if inv_freq is None:
        inv_freq = pe_inv_freq(channels)
    inv_freq = inv_freq.float()
    ch = pe_num_channels(channels)
    half = ch // 2
    tables = []
    for axis, size in enumerate((t, h, w)):
        pos = torch.arange(size).float()
        arg = inv_freq[:, None] * pos[None, :]                   # einsum("i,j->ij") :167-172
        emb = torch.cat([arg.sin(), arg.cos()], dim=0)           # stack+flatten :159-161 -> [ch, size]
        tab = torch.zeros(size, channels, dtype=torch.float32)
        lo = axis * ch
        for k in range(ch):
            c = lo + k
            if c < channels:
                tab[:, c] = emb[k]
        assert half * 2 == ch
        tables.append(tab.to(dtype))
    return tuple(tables)  # type: ignore[return-value]


def pe_table(channels: int, t: int, h: int, w: int, inv_freq: Optional[Tensor] = None,
             dtype=torch.float32) -> Tensor:
    """Full encoding, channels-last [t, h, w, C] (dwiseneuro.py:176-181)."""
    pt, ph, pw = pe_axis_tables(channels, t, h, w, inv_freq, dtype)
    return pt[:, None, None, :] + ph[None, :, None, :] + pw[None, None, :, :]


# --------------------------------------------------------------------------------------------
# elementwise pieces
# --------------------------------------------------------------------------------------------
def silu(x: Tensor) -> Tensor:
    """nn.SiLU (dwiseneuro.py:359): x * sigmoid(x)."""
    return x * torch.sigmoid(x)


def softplus(x: Tensor, beta: float, threshold: float = 20.0) -> Tensor:
    """nn.Softplus(beta) (dwiseneuro.py:282): log1p(exp(beta x))/beta, identity where beta*x > threshold."""
    bx = x * beta
    soft = torch.log1p(torch.exp(torch.clamp(bx, max=threshold))) / beta
    return torch.where(bx > threshold, x, soft)


def batch_norm(x: Tensor, prefix: str, sd: Dict[str, Tensor], training: bool,
               new_stats: Optional[Dict[str, Tensor]] = None, channel_index: Optional[np.ndarray] = None
               ) -> Tensor:
    """BatchNorm over all leading dims of a channels-last tensor (BatchNormAct, dwiseneuro.py:9-22).

    training: biased batch variance normalises; running stats get momentum-0.1 updates with the
    *unbiased* variance (torch semantics); ``new_stats`` (if given) receives the updated buffers.
    """
    gamma, beta = sd[prefix + ".weight"].to(x.dtype), sd[prefix + ".bias"].to(x.dtype)
    flat = x.reshape(-1, x.shape[-1])
    if training:
        mean = flat.mean(0)
        var = flat.var(0, unbiased=False)
        if new_stats is not None:
            n = flat.shape[0]
            with torch.no_grad():
                unbiased = var * (n / max(n - 1, 1))
                rm = sd[prefix + ".running_mean"].to(x.dtype)
                rv = sd[prefix + ".running_var"].to(x.dtype)
                new_stats[prefix + ".running_mean"] = (1 - BN_MOMENTUM) * rm + BN_MOMENTUM * mean
                new_stats[prefix + ".running_var"] = (1 - BN_MOMENTUM) * rv + BN_MOMENTUM * unbiased
                new_stats[prefix + ".num_batches_tracked"] = sd[prefix + ".num_batches_tracked"] + 1
    else:
        mean = sd[prefix + ".running_mean"].to(x.dtype)
        var = sd[prefix + ".running_var"].to(x.dtype)
    return (x - mean) * torch.rsqrt(var + BN_EPS) * gamma + beta


# --------------------------------------------------------------------------------------------
# convolutions as explicit channels-last math
# --------------------------------------------------------------------------------------------
def pointwise(x: Tensor, weight: Tensor) -> Tensor:
    """1x1x1 conv (dwiseneuro.py:91,118,306) == row-major GEMM: y[m, o] = sum_c x[m, c] W[o, c]."""
    w2 = weight.reshape(weight.shape[0], weight.shape[1]).to(x.dtype)
    return x @ w2.t()


# "stencil": explicit shifted sums (independent restatement, used for parity).  "library": the same op through
# torch.nn.functional.conv3d — the third-party kernel the reference itself calls — used only as the *fast* CPU
# baseline in bench.py; tests/test_oracle_golden.py checks the two agree.
DW_IMPL = "stencil"


def dw_spatial(x: Tensor, weight: Tensor, stride: int) -> Tensor:
    """Depth-wise (1,k,k) conv, stride (1,s,s), pad k//2 (dwiseneuro.py:96-100).  x: [B,T,H,W,C]."""
    k = weight.shape[-1]
    p = k // 2
    if DW_IMPL == "library":
        y = torch.nn.functional.conv3d(x.permute(0, 4, 1, 2, 3), weight.to(x.dtype), stride=(1, stride, stride),
                                       padding=(0, p, p), groups=x.shape[-1])
        return y.permute(0, 2, 3, 4, 1)
    b, t, h, w, c = x.shape
    ho = (h + 2 * p - k) // stride + 1
    wo = (w + 2 * p - k) // stride + 1
    xp = torch.zeros(b, t, h + 2 * p, w + 2 * p, c, dtype=x.dtype)
    xp[:, :, p:p + h, p:p + w] = x
    wk = weight.reshape(c, k, k).to(x.dtype)
    out = torch.zeros(b, t, ho, wo, c, dtype=x.dtype)
    for dy in range(k):
        for dx in range(k):
            win = xp[:, :, dy:dy + (ho - 1) * stride + 1:stride, dx:dx + (wo - 1) * stride + 1:stride]
            out = out + win * wk[:, dy, dx]
    return out


def dw_temporal(x: Tensor, weight: Tensor) -> Tensor:
    """Depth-wise (k,1,1) conv along T, pad k//2 (dwiseneuro.py:105-109).  x: [B,T,H,W,C]."""
    k = weight.shape[2]
    p = k // 2
    if DW_IMPL == "library":
        y = torch.nn.functional.conv3d(x.permute(0, 4, 1, 2, 3), weight.to(x.dtype), padding=(p, 0, 0),
                                       groups=x.shape[-1])
        return y.permute(0, 2, 3, 4, 1)
    b, t, h, w, c = x.shape
    xp = torch.zeros(b, t + 2 * p, h, w, c, dtype=x.dtype)
    xp[:, p:p + t] = x
    wk = weight.reshape(c, k).to(x.dtype)
    out = torch.zeros_like(x)
    for dt in range(k):
        out = out + xp[:, dt:dt + t] * wk[:, dt]
    return out


# --------------------------------------------------------------------------------------------
# blocks
# --------------------------------------------------------------------------------------------
def squeeze_excite(x: Tensor, prefix: str, sd: Dict[str, Tensor]) -> Tensor:
    """SqueezeExcite3d (dwiseneuro.py:38-43): x * sigmoid(W2 silu(W1 mean_{T,H,W}(x) + b1) + b2)."""
    b = x.shape[0]
    c = x.shape[-1]
    pooled = x.reshape(b, -1, c).mean(1)
    w1 = sd[prefix + ".conv_reduce.weight"].reshape(-1, c).to(x.dtype)
    b1 = sd[prefix + ".conv_reduce.bias"].to(x.dtype)
    w2 = sd[prefix + ".conv_expand.weight"].reshape(c, -1).to(x.dtype)
    b2 = sd[prefix + ".conv_expand.bias"].to(x.dtype)
    hid = silu(pooled @ w1.t() + b1)
    gate = torch.sigmoid(hid @ w2.t() + b2)
    return x * gate[:, None, None, None, :]


def inverted_residual(x: Tensor, prefix: str, sd: Dict[str, Tensor], stride: int, training: bool,
                      drop_scale: Optional[Tensor], new_stats, taps: Optional[dict] = None) -> Tensor:
    """InvertedResidual3d.forward (dwiseneuro.py:136-144) on a channels-last tensor [B,T,H,W,C].

    ``drop_scale``: optional per-sample DropPath factor (mask/keep_prob, dwiseneuro.py:46-54).
    ``taps``: optional dict receiving the raw (pre-BN) intermediate tensors for per-kernel tests.
    """
    shortcut = x
    y1 = pointwise(x, sd[prefix + ".conv_pw.0.weight"])
    z1 = silu(batch_norm(y1, prefix + ".conv_pw.1.bn", sd, training, new_stats))
    y2 = dw_spatial(z1, sd[prefix + ".spat_covn_dw.0.weight"], stride)
    z2 = silu(batch_norm(y2, prefix + ".spat_covn_dw.1.bn", sd, training, new_stats))
    y3 = dw_temporal(z2, sd[prefix + ".temp_covn_dw.0.weight"])
    z3 = silu(batch_norm(y3, prefix + ".temp_covn_dw.1.bn", sd, training, new_stats))
    u = squeeze_excite(z3, prefix + ".se", sd)
    y4 = pointwise(u, sd[prefix + ".conv_pwl.0.weight"])
    o4 = batch_norm(y4, prefix + ".conv_pwl.1.bn", sd, training, new_stats)
    if drop_scale is not None:
        o4 = o4 * drop_scale.to(o4.dtype)[:, None, None, None, None]
    # interpolate_shortcut, dwiseneuro.py:125-134
    _, _, h, w, c = shortcut.shape
    c_out = y4.shape[-1]
    if stride > 1:
        hs = torch.from_numpy(nearest_src_index(strided_out_size(h, stride), h))
        ws = torch.from_numpy(nearest_src_index(strided_out_size(w, stride), w))
        shortcut = shortcut[:, :, hs][:, :, :, ws]
    if c != c_out:
        shortcut = shortcut[..., torch.from_numpy(tile_channel_index(c_out, c))]
    sc = batch_norm(shortcut, prefix + ".bn_sc.bn", sd, training, new_stats)
    if taps is not None:
        taps.update(y1=y1, y2=y2, y3=y3, y4=y4)
    return o4 + sc


def core_forward(x: Tensor, sd: Dict[str, Tensor], strides: Sequence[int], training: bool,
                 drop_scales: Optional[Sequence[Optional[Tensor]]], new_stats,
                 block_inputs: Optional[list] = None) -> Tensor:
    """DepthwiseCore.forward (dwiseneuro.py:337-340): stem, then [PE, InvertedResidual3d] x N."""
    y0 = pointwise(x, sd["core.stem.0.weight"])
    x = batch_norm(y0, "core.stem.1.bn", sd, training, new_stats)
    for i, stride in enumerate(strides):
        _, t, h, w, c = x.shape
        inv_freq = sd.get(f"core.blocks.{2 * i}.inv_freq")
        x = x + pe_table(c, t, h, w, inv_freq, x.dtype)            # PositionalEncoding3d :184-192
        if block_inputs is not None:
            block_inputs.append(x)
        ds = None if drop_scales is None else drop_scales[i]
        x = inverted_residual(x, f"core.blocks.{2 * i + 1}", sd, stride, training, ds, new_stats)
    return x


def cortex_layer(x: Tensor, prefix: str, sd: Dict[str, Tensor], groups: int, training: bool,
                 drop_scale: Optional[Tensor], new_stats) -> Tensor:
    """ShuffleLayer.forward (dwiseneuro.py:228-234) on [B, T, C] (channels-last)."""
    w = sd[prefix + ".conv.weight"]                                   # [C_out, C_in/g, 1]
    c_out, cg_in = w.shape[0], w.shape[1]
    cg_out = c_out // groups
    outs = []
    for g in range(groups):
        wg = w[g * cg_out:(g + 1) * cg_out, :, 0].to(x.dtype)
        outs.append(x[..., g * cg_in:(g + 1) * cg_in] @ wg.t())
    y = torch.cat(outs, dim=-1)
    z = silu(batch_norm(y, prefix + ".bn.bn", sd, training, new_stats))
    z = z[..., torch.from_numpy(shuffle_source_index(c_out, groups))]
    if drop_scale is not None:
        z = z * drop_scale.to(z.dtype)[:, None, None]
    c_in = x.shape[-1]
    sc = x if c_in == c_out else x[..., torch.from_numpy(tile_channel_index(c_out, c_in))]
    sc = batch_norm(sc, prefix + ".bn_sc.bn", sd, training, new_stats)
    return z + sc


def readout(x: Tensor, prefix: str, sd: Dict[str, Tensor], groups: int, out_features: int,
            softplus_beta: float, drop_mask: Optional[Tensor]) -> Tensor:
    """Readout.forward (dwiseneuro.py:283-287).  x: [B, T, C] -> [B, N, T] (the reference's NCT output).

    ``drop_mask``: optional [B, C] Dropout1d factor (0 or 1/(1-p)), dwiseneuro.py:275.
    """
    if drop_mask is not None:
        x = x * drop_mask.to(x.dtype)[:, None, :]
    w = sd[prefix + ".layer.1.weight"]
    bias = sd[prefix + ".layer.1.bias"].to(x.dtype)
    n_pad, cg_in = w.shape[0], w.shape[1]
    ng = n_pad // groups
    outs = []
    for g in range(groups):
        wg = w[g * ng:(g + 1) * ng, :, 0].to(x.dtype)
        outs.append(x[..., g * cg_in:(g + 1) * cg_in] @ wg.t())
    z = torch.cat(outs, dim=-1) + bias
    z = z[..., :out_features]
    return softplus(z, softplus_beta).permute(0, 2, 1)


def config_from_state_dict(sd: Dict[str, Tensor]) -> dict:
    """Recover (strides are NOT recoverable) structural sizes from a reference state_dict."""
    n_blocks = len([k for k in sd if k.endswith("conv_pw.0.weight")])
    n_cortex = len([k for k in sd if k.startswith("cortex.layers.") and k.endswith(".conv.weight")])
    n_readouts = len([k for k in sd if k.startswith("readouts.") and k.endswith(".layer.1.weight")])
    return dict(n_blocks=n_blocks, n_cortex=n_cortex, n_readouts=n_readouts)


def forward(sd: Dict[str, Tensor], x_ncdhw: Tensor, *, strides: Sequence[int], readout_outputs: Sequence[int],
            groups: int = 2, softplus_beta: float = 0.07, index: Optional[int] = None,
            training: bool = False, core_drop_scales=None, cortex_drop_scales=None,
            readout_drop_masks=None, new_stats: Optional[dict] = None):
    """DwiseNeuro.forward (dwiseneuro.py:397-405).  ``x_ncdhw``: (B, C, T, H, W) as the reference takes it."""
    x = x_ncdhw.permute(0, 2, 3, 4, 1)                                   # -> [B,T,H,W,C]
    x = core_forward(x, sd, strides, training, core_drop_scales, new_stats)
    x = x.mean(dim=(2, 3))                                               # AdaptiveAvgPool3d((None,1,1)) :374,400
    n_cortex = config_from_state_dict(sd)["n_cortex"]
    for i in range(n_cortex):
        ds = None if cortex_drop_scales is None else cortex_drop_scales[i]
        x = cortex_layer(x, f"cortex.layers.{i}", sd, groups, training, ds, new_stats)
    def one(m: int) -> Tensor:
        dm = None if readout_drop_masks is None else readout_drop_masks[m]
        return readout(x, f"readouts.{m}", sd, groups, readout_outputs[m], softplus_beta, dm)
    if index is None:
        return [one(m) for m in range(len(readout_outputs))]
    return one(index)


# --------------------------------------------------------------------------------------------
# loss / optimizer / EMA / metric / predictor
# --------------------------------------------------------------------------------------------
def mice_poisson_loss(preds: Sequence[Tensor], targets: Sequence[Tensor], mice_weights: Tensor,
                      eps: float = 1e-8) -> Tensor:
    """MicePoissonLoss.forward (losses.py:10-21): sum_m sum_{b,n,t} w[b,m]/sum(w) * (x - y log(x+eps))."""
    w = mice_weights / mice_weights.sum()
    total = preds[0].new_zeros(())
    for m, (x, y) in enumerate(zip(preds, targets)):
        wm = w[:, m]
        if bool((wm != 0).any()):
            nll = x - y * torch.log(x + eps)                     # PoissonNLLLoss(log_input=False, full=False)
            total = total + (nll * wm.to(x.dtype)[:, None, None]).sum()
    return total


def adamw_step(p: Tensor, g: Tensor, m: Tensor, v: Tensor, step: int, lr: float, beta1: float = 0.9,
               beta2: float = 0.999, eps: float = 1e-8, weight_decay: float = 0.05):
    """torch.optim.AdamW single-tensor update (the optimizer named at true_batch_001.py:45-48)."""
    p = p * (1 - lr * weight_decay)
    m = beta1 * m + (1 - beta1) * g
    v = beta2 * v + (1 - beta2) * g * g
    bc1 = 1 - beta1 ** step
    bc2 = 1 - beta2 ** step
    denom = v.sqrt() / math.sqrt(bc2) + eps
    p = p - (lr / bc1) * m / denom
    return p, m, v


def ema_update(ema_v: Tensor, model_v: Tensor, decay: float) -> Tensor:
    """ModelEma.update (ema.py:47-55): e <- decay*e + (1-decay)*m, cast back to e's dtype
    (int64 ``num_batches_tracked`` is truncated by ``copy_``)."""
    return (decay * ema_v + (1.0 - decay) * model_v).to(ema_v.dtype)


def corr(y1: np.ndarray, y2: np.ndarray, axis=-1, eps: float = 1e-8) -> np.ndarray:
    """metrics.py:11-31: Pearson correlation with (std + eps), ddof 0."""
    y1 = (y1 - y1.mean(axis=axis, keepdims=True)) / (y1.std(axis=axis, keepdims=True, ddof=0) + eps)
    y2 = (y2 - y2.mean(axis=axis, keepdims=True)) / (y2.std(axis=axis, keepdims=True, ddof=0) + eps)
    return (y1 * y2).mean(axis=axis)


def window_indexes(index: int, size: int, step: int) -> List[int]:
    """IndexesGenerator(position="last").make_indexes (indexes.py:12-30): size frames ending at index."""
    behind = (size - 1) * step
    return list(range(index - behind, index + 1, step))


def predict_trial(forward_one, inputs: Tensor, n_neurons: int, size: int = 16, step: int = 2) -> np.ndarray:
    """Predictor.predict_trial (predictors.py:37-55) with "ones" blend weights.

    ``forward_one(window[1,C,size,H,W]) -> [N, size]`` is the eval-mode model for one mouse.
    """
    length = inputs.shape[1]
    responses = np.zeros((n_neurons, length), dtype=np.float32)
    counts = np.zeros(length, dtype=np.float32)
    behind = (size - 1) * step
    for index in range(behind, length):
        idx = window_indexes(index, size, step)
        pred = forward_one(inputs[:, idx].unsqueeze(0))
        responses[:, idx] += pred.detach().cpu().numpy().astype(np.float32)
        counts[idx] += 1.0
    responses /= np.clip(counts, 1.0, None)
    return responses


# --------------------------------------------------------------------------------------------
# synthetic weights with the reference's init rule (utils.py:46-56), for fixtures without the reference
# --------------------------------------------------------------------------------------------
def make_state_dict(*, readout_outputs, in_channels=5, core_features=(64, 64, 64, 64, 128, 128, 128, 256, 256),
                    spatial_kernel=3, temporal_kernel=5, expansion_ratio=6, se_reduce_ratio=32,
                    cortex_features=(1024, 2048, 4096), groups=2, seed=0, randomize_bn=False
                    ) -> Dict[str, Tensor]:
    """Build a state_dict with the reference's key names/shapes and init_weights rule:
    conv ~ N(0, sqrt(2/fan_out)), fan_out = prod(k)*out/groups; BN weight 1, bias 0; conv bias 0."""
    rng = np.random.default_rng(seed)
    sd: Dict[str, Tensor] = {}

    def conv(name, shape, conv_groups=1, bias=False):
        fan_out = int(np.prod(shape[2:])) * shape[0] // conv_groups
        sd[name + ".weight"] = torch.from_numpy(
            rng.normal(0, math.sqrt(2.0 / fan_out), size=shape).astype(np.float32))
        if bias:
            sd[name + ".bias"] = torch.zeros(shape[0])
            if randomize_bn:
                sd[name + ".bias"] = torch.from_numpy(rng.normal(0, 0.1, size=shape[0]).astype(np.float32))

    def bn(name, c):
        if randomize_bn:
            sd[name + ".weight"] = torch.from_numpy(rng.uniform(0.5, 1.5, size=c).astype(np.float32))
            sd[name + ".bias"] = torch.from_numpy(rng.normal(0, 0.2, size=c).astype(np.float32))
            sd[name + ".running_mean"] = torch.from_numpy(rng.normal(0, 0.2, size=c).astype(np.float32))
            sd[name + ".running_var"] = torch.from_numpy(rng.uniform(0.5, 1.5, size=c).astype(np.float32))
        else:
            sd[name + ".weight"] = torch.ones(c)
            sd[name + ".bias"] = torch.zeros(c)
            sd[name + ".running_mean"] = torch.zeros(c)
            sd[name + ".running_var"] = torch.ones(c)
        sd[name + ".num_batches_tracked"] = torch.tensor(0, dtype=torch.int64)

    f = list(core_features)
    conv("core.stem.0", (f[0], in_channels, 1, 1, 1))
    bn("core.stem.1.bn", f[0])
    for i in range(len(f)):
        c_in = f[i]
        c_out = f[min(i + 1, len(f) - 1)]
        mid = c_in * expansion_ratio
        sd[f"core.blocks.{2 * i}.inv_freq"] = pe_inv_freq(c_in)
        p = f"core.blocks.{2 * i + 1}"
        conv(p + ".conv_pw.0", (mid, c_in, 1, 1, 1))
        bn(p + ".conv_pw.1.bn", mid)
        conv(p + ".spat_covn_dw.0", (mid, 1, 1, spatial_kernel, spatial_kernel), conv_groups=mid)
        bn(p + ".spat_covn_dw.1.bn", mid)
        conv(p + ".temp_covn_dw.0", (mid, 1, temporal_kernel, 1, 1), conv_groups=mid)
        bn(p + ".temp_covn_dw.1.bn", mid)
        rd = mid // se_reduce_ratio
        conv(p + ".se.conv_reduce", (rd, mid, 1, 1, 1), bias=True)
        conv(p + ".se.conv_expand", (mid, rd, 1, 1, 1), bias=True)
        conv(p + ".conv_pwl.0", (c_out, mid, 1, 1, 1))
        bn(p + ".conv_pwl.1.bn", c_out)
        bn(p + ".bn_sc.bn", c_out)
    prev = f[-1]
    for i, c in enumerate(cortex_features):
        p = f"cortex.layers.{i}"
        conv(p + ".conv", (c, prev // groups, 1), conv_groups=groups)
        bn(p + ".bn.bn", c)
        bn(p + ".bn_sc.bn", c)
        prev = c
    for m, n in enumerate(readout_outputs):
        n_pad = int(math.ceil(n / groups) * groups)
        conv(f"readouts.{m}.layer.1", (n_pad, prev // groups, 1), conv_groups=groups, bias=True)
    return sd
