"""Single-trial correlation (reference: src/metrics.py:11-31), the parity metric of the north star."""
import numpy as np


def corr(y1: np.ndarray, y2: np.ndarray, axis=-1, eps: float = 1e-8, **kwargs) -> np.ndarray:
    y1 = (y1 - y1.mean(axis=axis, keepdims=True)) / (y1.std(axis=axis, keepdims=True, ddof=0) + eps)
    y2 = (y2 - y2.mean(axis=axis, keepdims=True)) / (y2.std(axis=axis, keepdims=True, ddof=0) + eps)
    return (y1 * y2).mean(axis=axis, **kwargs)
