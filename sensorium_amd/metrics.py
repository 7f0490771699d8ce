"""Single-trial correlation (reference: src/metrics.py:11-31), the parity metric of the north star."""
from collections import defaultdict

import numpy as np
import torch

from .engine import Metric


def corr(y1: np.ndarray, y2: np.ndarray, axis=-1, eps: float = 1e-8, **kwargs) -> np.ndarray:
    y1 = (y1 - y1.mean(axis=axis, keepdims=True)) / (y1.std(axis=axis, keepdims=True, ddof=0) + eps)
    y2 = (y2 - y2.mean(axis=axis, keepdims=True)) / (y2.std(axis=axis, keepdims=True, ddof=0) + eps)
    return (y1 * y2).mean(axis=axis, **kwargs)


class CorrelationMetric(Metric):
    """Validation metric of scripts/train.py:137-139 (reference: src/metrics.py:34-82): per mouse, the rows whose
    mouse weight is non-zero are flattened to (samples*time, neurons) and ``corr`` is averaged over neurons; the
    epoch value is the mean over mice -> ``val_corr`` plus ``val_corr_mouse_<i>``."""
    name = "corr"
    better = "max"

    def __init__(self):
        self.reset()

    def reset(self):
        self.predictions = defaultdict(list)
        self.targets = defaultdict(list)

    def update(self, step_output: dict):
        pred_tensors = step_output["prediction"]
        target_tensors, mice_weights = step_output["target"]
        for mouse_index, (pred, target) in enumerate(zip(pred_tensors, target_tensors)):
            mask = mice_weights[..., mouse_index] != 0.0
            if not bool(torch.any(mask)):
                continue
            pred, target = pred[mask], target[mask]
            if target.dim() == 3:
                pred = pred.transpose(1, 2).reshape(-1, pred.shape[1])
                target = target.transpose(1, 2).reshape(-1, target.shape[1])
            self.predictions[mouse_index].append(pred.float().cpu().numpy())
            self.targets[mouse_index].append(target.float().cpu().numpy())

    def compute(self):
        return {m: corr(np.concatenate(self.predictions[m], axis=0), np.concatenate(self.targets[m], axis=0),
                        axis=0).mean() for m in self.predictions}

    def epoch_complete(self, state):
        with torch.no_grad():
            mice_corr = self.compute()
        prefix = f"{state.phase}_" if state.phase else ""
        for mouse_index, value in mice_corr.items():
            state.metrics[f"{prefix}{self.name}_mouse_{mouse_index}"] = value
        state.metrics[prefix + self.name] = np.mean(list(mice_corr.values()))
