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
