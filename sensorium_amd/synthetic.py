"""Synthetic clips with the reference's batch structure (shapes from src/datasets.py:172-187,
value ranges from src/inputs.py:26-33 and src/responses.py:25-29).  No dataset is available offline."""
from __future__ import annotations

from typing import Sequence

import numpy as np
import torch


def make_batch(batch: int, frames: int, height: int, width: int, readout_outputs: Sequence[int], seed: int = 20231122,
               device=None):
    """Returns ``input (B,5,T,H,W) fp32`` and ``target = ([ (B,N_m,T) ]*n_mice, mice_weights (B,n_mice))``."""
    rng = np.random.default_rng(seed)
    x = np.zeros((batch, 5, frames, height, width), dtype=np.float32)
    x[:, 0] = rng.integers(0, 256, size=(batch, frames, height, width)).astype(np.float32)   # un-normalised video
    scale = np.array([10, 5, 20, 20], dtype=np.float32)
    shift = np.array([30, 5, 100, 70], dtype=np.float32)
    beh = rng.normal(size=(batch, 4, frames)).astype(np.float32) * scale[None, :, None] + shift[None, :, None]
    x[:, 1:] = np.clip(beh, 0, None)[:, :, :, None, None]            # behaviour / pupil centre: per-(b,t) scalars
    n_mice = len(readout_outputs)
    weights = np.zeros((batch, n_mice), dtype=np.float32)
    weights[np.arange(batch), np.arange(batch) % n_mice] = 1.0        # one-hot mouse (datasets.py:185-186)
    targets = []
    for m, n in enumerate(readout_outputs):
        t = np.maximum(rng.normal(size=(batch, n, frames)), 0).astype(np.float32) * 10.0
        t *= weights[:, m][:, None, None]                            # zero targets for the other mice (datasets.py:172-184)
        targets.append(t)
    xt = torch.from_numpy(x)
    tt = [torch.from_numpy(t) for t in targets]
    wt = torch.from_numpy(weights)
    if device is not None:
        host = wt
        xt, tt, wt = xt.to(device), [t.to(device) for t in tt], wt.to(device)
        wt._dwn_host_version = wt._version   # (an in-place edit of wt afterwards invalidates the host copy)
        wt._dwn_host = host           # the host knows which mouse every sample belongs to: MouseModel.train_step uses it (no read-back)
    return xt, (tt, wt)
