"""MicePoissonLoss drop-in (reference: src/losses.py:5-21) on the HIP Poisson kernels."""
from __future__ import annotations

import torch
from torch import nn

from . import ops


class MicePoissonLoss(nn.Module):
    """sum over mice of the weight-normalised Poisson NLL (log_input=False, full=False).

    ``forward(inputs, targets)`` with ``inputs`` = list of (B, N_m, T) predictions and ``targets`` =
    (list of (B, N_m, T) targets, (B, n_mice) mice_weights), exactly as the reference.  The per-mouse
    ``torch.any(mask)`` host sync of the reference (losses.py:17) is not needed: samples with weight 0
    contribute exactly 0 in the kernel, so the value and gradients are identical.
    """

    def __init__(self, log_input: bool = False, full: bool = False, eps: float = 1e-8):
        super().__init__()
        if log_input or full:
            raise NotImplementedError("sensorium_amd.MicePoissonLoss: only log_input=False, full=False is built")
        self.eps = float(eps)

    def forward(self, inputs, targets):
        target_tensors, mice_weights = targets
        weights = (mice_weights / mice_weights.sum()).float()
        total = None
        for m, (pred, target) in enumerate(zip(inputs, target_tensors)):
            term = ops.PoissonLossFn.apply(pred, target, weights[..., m], self.eps)
            total = term if total is None else total + term
        return total
