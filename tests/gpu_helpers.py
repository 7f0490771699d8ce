"""Helpers shared by the GPU parity tests (imported only by tests)."""
import ctypes as C
import math

import numpy as np
import torch

from oracle import dwiseneuro_oracle as orc


def rel(a, b):
    a = a.detach().double().cpu().reshape(-1)
    b = b.detach().double().cpu().reshape(-1)
    return float((a - b).norm() / (b.norm() + 1e-30))


def tol(dtype, f32=1e-3, bf16=4e-2):
    return f32 if dtype == torch.float32 else bf16


def dev():
    return torch.device("cuda", 0)


def stream():
    return torch.cuda.current_stream().cuda_stream


def stats_buffer(c):
    return torch.zeros(32 * 2 * c, dtype=torch.float64, device=dev())


def read_stats(buf, c):
    s = buf.view(32, 2, c).sum(0)
    return s[0], s[1]


def load_desc(L, p, ld, **kw):
    d = L.LoadDesc()
    d.p = p.data_ptr()
    d.ld = ld
    d.rows_per_sample = 1
    for k, v in kw.items():
        setattr(d, k, v.data_ptr() if isinstance(v, torch.Tensor) else v)
    return d


def sd_to_module(module, sd):
    missing = module.load_state_dict({k: v.clone() for k, v in sd.items()}, strict=True)
    assert not missing.missing_keys and not missing.unexpected_keys
    return module


NUM_NEURONS_ALL = (7863, 7908, 8202, 7939, 8122, 7440, 7928, 8285, 7671, 7495)      # src/constants.py:18,26,38


def synth_inputs(rng, b, t, h, w, readout_outputs):
    """The synthetic clip generator of oracle/make_golden.py (same draw order), restated here because that script imports
    the reference and cannot run on the GPU box: ch0 video 0..255, ch1-4 per-(b,t) scalars broadcast over HxW."""
    x = np.zeros((b, 5, t, h, w), dtype=np.float32)
    x[:, 0] = rng.integers(0, 256, size=(b, t, h, w)).astype(np.float32)
    scale = np.array([10, 5, 20, 20], dtype=np.float32)
    shift = np.array([30, 5, 100, 70], dtype=np.float32)
    beh = np.clip(rng.normal(size=(b, 4, t)).astype(np.float32) * scale[None, :, None] + shift[None, :, None], 0, None)
    x[:, 1:] = beh[:, :, :, None, None]
    targets = [np.maximum(rng.normal(size=(b, n, t)), 0).astype(np.float32) * 10 for n in readout_outputs]
    weights = np.zeros((b, len(readout_outputs)), dtype=np.float32)
    for i in range(b):
        weights[i, i % len(readout_outputs)] = 1.0
    return x, targets, weights


def analytically_zero_grad(name: str) -> bool:
    """The 21 parameters whose gradient is identically zero: a per-channel bias whose only consumers are
    BatchNorms reached through linear ops (stem BN bias; each block's two output BN biases; the shortcut BN bias of the first
    two cortex layers).  What the kernels compute for them is summation noise; relative comparisons skip them."""
    import re as _re
    return bool(name == "core.stem.1.bn.bias" or _re.fullmatch(r"core\.blocks\.\d+\.(conv_pwl\.1\.bn|bn_sc\.bn)\.bias", name)
                or _re.fullmatch(r"cortex\.layers\.[01]\.bn_sc\.bn\.bias", name))
