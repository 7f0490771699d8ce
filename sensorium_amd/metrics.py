"""Single-trial correlation (reference: src/metrics.py:11-31), the parity metric of the north star."""

import numpy as np
import torch

from .engine import Metric


def corr(y1: np.ndarray, y2: np.ndarray, axis=-1, eps: float = 1e-8, **kwargs) -> np.ndarray:
    y1 = (y1 - y1.mean(axis=axis, keepdims=True)) / (y1.std(axis=axis, keepdims=True, ddof=0) + eps)
    y2 = (y2 - y2.mean(axis=axis, keepdims=True)) / (y2.std(axis=axis, keepdims=True, ddof=0) + eps)
    return (y1 * y2).mean(axis=axis, **kwargs)


class CorrelationMetric(Metric):
    """``val_corr`` / ``val_corr_mouse_<i>`` of scripts/train.py:137-139 — what src/metrics.py:34-82 computes, as a STREAMING
    statistic on the device: per mouse and neuron the five running sums (n, sum p, sum t, sum p^2, sum t^2, sum p*t) over the
    (sample, frame) rows whose mouse weight is not zero, in float64, updated by a handful of reductions per batch.  The
    reference keeps every prediction and target of the epoch on the host (one device->host copy and one ``torch.any`` sync
    per mouse and batch, O(epoch) memory) and calls ``corr`` on the concatenation at the end; the Pearson coefficient with
    the reference's ``eps`` on each standard deviation is a function of those sums alone:
    ``(E[pt] - E[p]E[t]) / ((std p + eps)(std t + eps))`` — equal to ``corr(..., axis=0)`` up to rounding (checked at 1e-6).
    Mice without a weighted row in the epoch are left out, as in the reference."""
    name = "corr"
    better = "max"
    eps = 1e-8

    def __init__(self):
        self.reset()

    def reset(self):
        self.sums = {}                # mouse index -> [count (0-d), sum_p, sum_t, sum_pp, sum_tt, sum_pt] (float64, on the device)

    @torch.no_grad()
    def update(self, step_output: dict):
        predictions = step_output["prediction"]
        targets, mice_weights = step_output["target"]
        for k, (p, t) in enumerate(zip(predictions, targets)):
            rows = (mice_weights[..., k] != 0).to(torch.float64)              # [B]: 1 for this mouse's samples
            p, t = p.to(torch.float64), t.to(torch.float64)
            if p.dim() == 2:                                                  # (B, N): one row per sample
                p, t = p.unsqueeze(-1), t.unsqueeze(-1)
            w = rows.view(-1, 1, 1)
            pw, tw = p * w, t * w
            new = [rows.sum() * p.shape[-1], pw.sum((0, 2)), tw.sum((0, 2)), (pw * p).sum((0, 2)), (tw * t).sum((0, 2)),
                   (pw * t).sum((0, 2))]
            old = self.sums.get(k)
            self.sums[k] = new if old is None else [a + b for a, b in zip(old, new)]

    def compute(self):
        out = {}
        for k, (n, sp, st, spp, stt, spt) in self.sums.items():
            n = float(n)                                                      # the epoch's one read-back per mouse
            if n == 0:
                continue
            mp, mt = sp / n, st / n
            sd_p = (spp / n - mp * mp).clamp_min(0).sqrt()
            sd_t = (stt / n - mt * mt).clamp_min(0).sqrt()
            r = (spt / n - mp * mt) / ((sd_p + self.eps) * (sd_t + self.eps))
            out[k] = float(r.mean())
        return out

    def epoch_complete(self, state):
        per_mouse = self.compute()
        prefix = f"{state.phase}_" if state.phase else ""
        for k, value in per_mouse.items():
            state.metrics[f"{prefix}{self.name}_mouse_{k}"] = value
        state.metrics[prefix + self.name] = float(np.mean(list(per_mouse.values())))
