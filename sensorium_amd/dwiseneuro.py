"""MI355X-native DwiseNeuro: drop-in for ``src/models/dwiseneuro.py`` of lRomul/sensorium.

Same constructor arguments, same ``forward(x, index)`` contract and — because every learnable tensor lives in
a standard ``nn.Conv3d`` / ``nn.Conv1d`` / ``nn.BatchNorm*`` *holder* placed at the reference's attribute
path — the same ``state_dict`` keys, shapes and ordering, so reference checkpoints load with
``strict=True``, ``init_weights`` (src/utils.py:46-63) matches the layers by ``isinstance`` and ``ModelEma``
can ``deepcopy`` the module.  The holders' own ``forward`` is never called: all arithmetic runs in the
hand-written gfx950 kernels behind ``sensorium_amd.ops`` on channels-last activations.

Compute dtype: fp32 by default (parity mode); bf16 storage with fp32 accumulation/statistics when called
under ``torch.autocast`` (the reference trains under fp16 autocast, src/argus_models.py:50 — fp16 and bf16
autocast both select the bf16 path here) or when ``compute_dtype=torch.bfloat16`` is set explicitly.

Constraints of the HIP path (raised loudly, no fallback): channel counts multiples of 8, spatial_kernel 3,
temporal_kernel 3 or 5, CUDA/HIP tensors only, backward only in training mode.
"""
from __future__ import annotations

import math
from typing import Optional, Sequence

import torch
from torch import nn

from . import ops


def _select_dtype(explicit: Optional[torch.dtype]) -> torch.dtype:
    if explicit is not None:
        return explicit
    if torch.is_autocast_enabled():
        return torch.bfloat16
    return torch.float32


class BatchNormAct(nn.Module):
    """Parameter holder for BatchNorm(+activation) (reference: dwiseneuro.py:9-22).  The activation is fused
    into the consuming HIP kernel; ``apply_act`` only records what the reference would apply."""

    def __init__(self, num_features: int, bn_layer=nn.BatchNorm3d, apply_act: bool = True):
        super().__init__()
        self.bn = bn_layer(num_features)
        self.apply_act = apply_act


class SqueezeExcite3d(nn.Module):
    """Holder for the SE 1x1x1 convs (reference: dwiseneuro.py:25-43)."""

    def __init__(self, in_features: int, reduce_ratio: int = 16):
        super().__init__()
        rd_channels = in_features // reduce_ratio
        self.conv_reduce = nn.Conv3d(in_features, rd_channels, (1, 1, 1), bias=True)
        self.conv_expand = nn.Conv3d(rd_channels, in_features, (1, 1, 1), bias=True)


class DropPath(nn.Module):
    """Stochastic depth (reference: dwiseneuro.py:46-67): the per-sample factor mask/keep_prob is generated
    here (RNG is plumbing) and applied inside the residual kernel."""

    def __init__(self, drop_prob: float = 0.0):
        super().__init__()
        self.drop_prob = float(drop_prob)
        self._pooled: Optional[torch.Tensor] = None      # this forward's factors, drawn by DwiseNeuro for all layers at once

    def sample(self, batch: int, device) -> Optional[torch.Tensor]:
        if self.drop_prob == 0.0 or not self.training:
            return None
        if self._pooled is not None:
            scale, self._pooled = self._pooled, None
            if scale.shape[0] == batch and scale.device == device:
                return scale
        keep = 1.0 - self.drop_prob
        scale = torch.empty(batch, dtype=torch.float32, device=device).bernoulli_(keep)
        if keep > 0.0:
            scale.div_(keep)
        return scale

    def extra_repr(self):
        return f"drop_prob={self.drop_prob:0.3f}"


class PositionalEncoding3d(nn.Module):
    """Holder of ``inv_freq`` (reference: dwiseneuro.py:147-192).  The encoding is separable
    (every channel depends on one of t/h/w), so three small tables are added on the fly by the consumer."""

    def __init__(self, channels: int):
        super().__init__()
        self.orig_channels = channels
        ch = int(math.ceil(channels / 6) * 2)
        if ch % 2:
            ch += 1
        self.channels = ch
        inv_freq = 1.0 / (10000 ** (torch.arange(0, ch, 2).float() / ch))
        self.register_buffer("inv_freq", inv_freq)
        self.register_buffer("cached_encoding", None, persistent=False)


class InvertedResidual3d(nn.Module):
    """PE + inverted-residual block (reference: dwiseneuro.py:70-144) executed by ``ops.BlockFn``."""

    def __init__(self, in_features: int, out_features: int, spatial_kernel: int = 3, temporal_kernel: int = 3,
                 spatial_stride: int = 1, expansion_ratio: int = 3, se_reduce_ratio: int = 16,
                 drop_path_rate: float = 0.0, bias: bool = False):
        super().__init__()
        if bias:
            raise NotImplementedError("sensorium_amd: biased convs inside InvertedResidual3d are not built")
        self.in_features = in_features
        self.out_features = out_features
        self.spatial_stride = spatial_stride
        self.spatial_kernel = spatial_kernel
        self.temporal_kernel = temporal_kernel
        mid = in_features * expansion_ratio
        self.mid_features = mid
        stride = (1, spatial_stride, spatial_stride)
        sp, tp = spatial_kernel // 2, temporal_kernel // 2
        self.conv_pw = nn.Sequential(nn.Conv3d(in_features, mid, (1, 1, 1), bias=False), BatchNormAct(mid))
        self.spat_covn_dw = nn.Sequential(
            nn.Conv3d(mid, mid, (1, spatial_kernel, spatial_kernel), stride=stride, padding=(0, sp, sp),
                      groups=mid, bias=False),
            BatchNormAct(mid))
        self.temp_covn_dw = nn.Sequential(
            nn.Conv3d(mid, mid, (temporal_kernel, 1, 1), stride=(1, 1, 1), padding=(tp, 0, 0), groups=mid,
                      bias=False),
            BatchNormAct(mid))
        self.se = SqueezeExcite3d(mid, reduce_ratio=se_reduce_ratio)
        self.conv_pwl = nn.Sequential(nn.Conv3d(mid, out_features, (1, 1, 1), bias=False),
                                      BatchNormAct(out_features, apply_act=False))
        self.drop_path = DropPath(drop_prob=drop_path_rate)
        self.bn_sc = BatchNormAct(out_features, apply_act=False)
        self._geom_cache: dict = {}

    def __deepcopy__(self, memo):
        cache, self._geom_cache = self._geom_cache, {}
        try:
            cls = self.__class__
            new = cls.__new__(cls)
            memo[id(self)] = new
            import copy
            for k, v in self.__dict__.items():
                setattr(new, k, copy.deepcopy(v, memo))
            return new
        finally:
            self._geom_cache = cache

    def bn_modules(self):
        return [self.conv_pw[1].bn, self.spat_covn_dw[1].bn, self.temp_covn_dw[1].bn, self.conv_pwl[1].bn,
                self.bn_sc.bn]

    def parameters_in_kernel_order(self):
        bn = self.bn_modules()
        return (self.conv_pw[0].weight, bn[0].weight, bn[0].bias,
                self.spat_covn_dw[0].weight, bn[1].weight, bn[1].bias,
                self.temp_covn_dw[0].weight, bn[2].weight, bn[2].bias,
                self.se.conv_reduce.weight, self.se.conv_reduce.bias,
                self.se.conv_expand.weight, self.se.conv_expand.bias,
                self.conv_pwl[0].weight, bn[3].weight, bn[3].bias, bn[4].weight, bn[4].bias)

    def geometry(self, pe: PositionalEncoding3d, t: int, h: int, w: int, device):
        """PE tables + nearest-neighbour shortcut index maps for this input size (cached)."""
        key = (t, h, w, str(device))
        geom = self._geom_cache.get(key)
        if geom is None:
            pt, ph, pw = ops.pe_axis_tables(self.in_features, pe.inv_freq, t, h, w)
            s = self.spatial_stride
            ho, wo = (h - 1) // s + 1, (w - 1) // s + 1
            hsrc = ops.nearest_src_index(ho, h)
            wsrc = ops.nearest_src_index(wo, w)
            to = lambda a: torch.as_tensor(a).to(device)
            geom = (pt.to(device), ph.to(device), pw.to(device), to(hsrc), to(wsrc),
                    to(ops.inverse_index(hsrc, h)), to(ops.inverse_index(wsrc, w)))
            self._geom_cache = {key: geom}
        return geom

    def forward(self, x: torch.Tensor, pe: PositionalEncoding3d, dtype: torch.dtype, x_has_pe: bool = False,
                out_pe=None) -> torch.Tensor:
        """``x_has_pe``: the producer of ``x`` already added this block's positional encoding (the stem / the
        previous block's residual kernel do, inside DepthwiseCore).  ``out_pe``: the NEXT block's PE tables, to be
        folded into this block's output pass."""
        b, t, h, w, _ = x.shape
        geom = self.geometry(pe, t, h, w, x.device)
        drop = self.drop_path.sample(b, x.device)
        return ops.BlockFn.apply(x, drop, self, geom, dtype, x_has_pe, out_pe, *self.parameters_in_kernel_order())


class ShuffleLayer(nn.Module):
    """Cortex layer (reference: dwiseneuro.py:195-234) executed by ``ops.CortexFn``."""

    def __init__(self, in_features: int, out_features: int, groups: int = 1, drop_path_rate: float = 0.0):
        super().__init__()
        self.in_features = in_features
        self.out_features = out_features
        self.groups = groups
        self.conv = nn.Conv1d(in_features, out_features, (1,), groups=groups, bias=False)
        self.bn = BatchNormAct(out_features, bn_layer=nn.BatchNorm1d)
        self.drop_path = DropPath(drop_prob=drop_path_rate)
        self.bn_sc = BatchNormAct(out_features, bn_layer=nn.BatchNorm1d, apply_act=False)

    def forward(self, x: torch.Tensor, dtype: torch.dtype) -> torch.Tensor:
        drop = self.drop_path.sample(x.shape[0], x.device)
        return ops.CortexFn.apply(x, drop, self, dtype, self.conv.weight, self.bn.bn.weight, self.bn.bn.bias,
                                  self.bn_sc.bn.weight, self.bn_sc.bn.bias)


class Cortex(nn.Module):
    """Sequence of ShuffleLayers (reference: dwiseneuro.py:237-263)."""

    def __init__(self, in_features: int, features: Sequence[int], groups: int = 1, drop_path_rate: float = 0.0):
        super().__init__()
        self.layers = nn.Sequential()
        prev = in_features
        for num_features in features:
            self.layers.append(ShuffleLayer(prev, num_features, groups=groups, drop_path_rate=drop_path_rate))
            prev = num_features

    def forward(self, x: torch.Tensor, dtype: torch.dtype) -> torch.Tensor:
        for layer in self.layers:
            x = layer(x, dtype)
        return x


class Readout(nn.Module):
    """Per-mouse readout (reference: dwiseneuro.py:266-287) executed by ``ops.ReadoutFn``."""

    def __init__(self, in_features: int, out_features: int, groups: int = 1, softplus_beta: float = 1.0,
                 drop_rate: float = 0.0):
        super().__init__()
        self.in_features = in_features
        self.out_features = out_features
        self.groups = groups
        self.softplus_beta = float(softplus_beta)
        self.drop_rate = float(drop_rate)
        padded = int(math.ceil(out_features / groups) * groups)
        self.layer = nn.Sequential(nn.Dropout1d(p=drop_rate),
                                   nn.Conv1d(in_features, padded, (1,), groups=groups, bias=True))
        self.gate = nn.Softplus(beta=softplus_beta)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        mask = None
        if self.training and self.drop_rate > 0.0:
            keep = 1.0 - self.drop_rate
            mask = torch.empty(x.shape[0], x.shape[2], dtype=torch.float32, device=x.device).bernoulli_(keep)
            mask.div_(keep)
        conv = self.layer[1]
        return ops.ReadoutFn.apply(x, mask, self, conv.weight, conv.bias)


class DepthwiseCore(nn.Module):
    """Stem + [PositionalEncoding3d, InvertedResidual3d] x N (reference: dwiseneuro.py:290-340)."""

    def __init__(self, in_channels: int = 1, features: Sequence[int] = (64, 128, 256, 512),
                 spatial_strides: Sequence[int] = (2, 2, 2, 2), spatial_kernel: int = 3, temporal_kernel: int = 3,
                 expansion_ratio: int = 3, se_reduce_ratio: int = 16, drop_path_rate: float = 0.0):
        super().__init__()
        num_blocks = len(features)
        assert num_blocks and num_blocks == len(spatial_strides)
        self.stem = nn.Sequential(nn.Conv3d(in_channels, features[0], (1, 1, 1), bias=False),
                                  BatchNormAct(features[0], apply_act=False))
        blocks = []
        nxt = features[0]
        for i in range(num_blocks):
            if i < num_blocks - 1:
                nxt = features[i + 1]
            blocks += [
                PositionalEncoding3d(features[i]),
                InvertedResidual3d(features[i], nxt, spatial_kernel=spatial_kernel, temporal_kernel=temporal_kernel,
                                   spatial_stride=spatial_strides[i], expansion_ratio=expansion_ratio,
                                   se_reduce_ratio=se_reduce_ratio,
                                   drop_path_rate=drop_path_rate * i / num_blocks, bias=False),
            ]
        self.blocks = nn.Sequential(*blocks)

    def forward(self, x: torch.Tensor, dtype: torch.dtype) -> torch.Tensor:
        # Every PositionalEncoding3d add is folded into the kernel that *produces* the tensor it is added to:
        # the stem's BN pass for block 0, block i's residual pass for block i+1 (no separate elementwise pass, and
        # the point-wise GEMM prologues stay plain loads).
        bn = self.stem[1].bn
        mods = list(self.blocks)
        pes, blks = mods[0::2], mods[1::2]
        _, _, t, h, w = x.shape
        sizes = []
        for blk in blks:
            sizes.append((h, w))
            s = blk.spatial_stride
            h, w = (h - 1) // s + 1, (w - 1) // s + 1
        tables = [blk.geometry(pe, t, hw[0], hw[1], x.device)[:3] for pe, blk, hw in zip(pes, blks, sizes)]
        x = ops.StemFn.apply(x, self.stem[0].weight, bn.weight, bn.bias, self, dtype, tables[0])
        for i, (pe, blk) in enumerate(zip(pes, blks)):
            x = blk(x, pe, dtype, True, tables[i + 1] if i + 1 < len(blks) else None)
        return x


class DwiseNeuro(nn.Module):
    """Drop-in for the reference ``DwiseNeuro`` (dwiseneuro.py:343-405)."""

    def __init__(self,
                 readout_outputs: Sequence[int],
                 in_channels: int = 5,
                 core_features: Sequence[int] = (64, 64, 64, 64, 128, 128, 128, 256, 256),
                 spatial_strides: Sequence[int] = (2, 1, 1, 1, 2, 1, 1, 2, 1),
                 spatial_kernel: int = 3,
                 temporal_kernel: int = 5,
                 expansion_ratio: int = 6,
                 se_reduce_ratio: int = 32,
                 cortex_features: Sequence[int] = (1024, 2048, 4096),
                 groups: int = 2,
                 softplus_beta: float = 0.07,
                 drop_rate: float = 0.4,
                 drop_path_rate: float = 0.1,
                 compute_dtype: Optional[torch.dtype] = None):
        super().__init__()
        self.compute_dtype = compute_dtype
        self.fp32_eval_products = "bf16x3"     # see set_fp32_eval_products
        self.core = DepthwiseCore(in_channels=in_channels, features=core_features, spatial_strides=spatial_strides,
                                  spatial_kernel=spatial_kernel, temporal_kernel=temporal_kernel,
                                  expansion_ratio=expansion_ratio, se_reduce_ratio=se_reduce_ratio,
                                  drop_path_rate=drop_path_rate)
        self.pool = nn.AdaptiveAvgPool3d((None, 1, 1))          # parameter-free; kept for the module tree / repr
        self.cortex = Cortex(in_features=core_features[-1], features=cortex_features, groups=groups,
                             drop_path_rate=drop_path_rate)
        self.readouts = nn.ModuleList()
        for n in readout_outputs:
            self.readouts.append(Readout(in_features=cortex_features[-1], out_features=n, groups=groups,
                                         softplus_beta=softplus_beta, drop_rate=drop_rate))

    def set_fp32_eval_products(self, mode: str = "bf16x3") -> "DwiseNeuro":
        """How the fp32 path multiplies in the EVAL-mode forward (val_step / predict run fp32, src/argus_models.py:73-99).
        ``"bf16x3"`` (default): every fp32 GEMM operand is split into bf16 hi + lo and multiplied as hi*hi + hi*lo + lo*hi on
        the bf16 matrix cores with fp32 accumulation — 5.8e-7 relative L2 (max 4e-6) from the native products on the full-width
        model, a sixth of the matrix-core time.  ``"native"``: ``v_mfma_f32_16x16x4_f32`` everywhere, so training-mode and
        eval-mode fp32 forwards of the same weights agree to fp32 rounding.  Training always uses the native products.
        (C-ABI: ``f32_products`` of dwn_block_args / dwn_cortex_args / dwn_readout_args.)"""
        if mode not in ("bf16x3", "native"):
            raise ValueError("fp32 eval products: 'bf16x3' or 'native'")
        self.fp32_eval_products = mode
        for m in self.modules():
            if isinstance(m, (InvertedResidual3d, ShuffleLayer, Readout)):
                m._dwn_fp32_native = mode == "native"
        return self

    def _draw_drop_paths(self, batch: int, device) -> None:
        """One draw for every stochastic-depth layer of this forward pass (12 layers: 2 small launches instead of 24).  Same
        distribution as DropPath.sample: factor = Bernoulli(keep) / keep per sample (0 where keep == 0: no 0/0)."""
        layers = [m for m in self.modules() if isinstance(m, DropPath) and m.drop_prob > 0.0 and m.training]
        if not layers:
            return
        key = (tuple(m.drop_prob for m in layers), batch)
        cache = getattr(self, "_dp_keep", None)
        if cache is None or cache[0] != key or cache[1].device != device:      # rebuilt when a drop-path schedule changes a rate
            keep = torch.tensor([1.0 - q for q in key[0]], dtype=torch.float32)
            inv = torch.where(keep > 0, 1.0 / keep.clamp_min(1e-30), torch.zeros_like(keep))
            cache = (key, keep.unsqueeze(1).expand(len(layers), batch).contiguous().to(device), inv.unsqueeze(1).to(device))
            self._dp_keep = cache
        factors = torch.bernoulli(cache[1]).mul_(cache[2])
        for m, row in zip(layers, factors.unbind(0)):
            m._pooled = row

    def trunk(self, x: torch.Tensor) -> torch.Tensor:
        """core -> pool -> cortex; returns channels-last [B, T, C] features in the compute dtype."""
        if x.dim() != 5:
            raise RuntimeError("DwiseNeuro expects (batch, channel, time, height, width)")
        if self.training:
            self._draw_drop_paths(x.shape[0], x.device)
        dtype = _select_dtype(self.compute_dtype)
        x = self.core(x, dtype)                                   # [B,T,h,w,C]
        x = ops.PoolFn.apply(x)                                   # [B,T,C]
        return self.cortex(x, dtype)

    def forward(self, x: torch.Tensor, index: Optional[int] = None):
        feats = self.trunk(x)
        if index is None:
            return [readout(feats) for readout in self.readouts]
        return self.readouts[index](feats)
