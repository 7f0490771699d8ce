"""Small helpers of the reference's ``src/utils.py`` that sit on the training path."""
from __future__ import annotations

import math
import re
from pathlib import Path

import numpy as np
from torch import nn


def get_lr(base_lr: float, batch_size: int, base_batch_size: int = 4) -> float:
    """Linear scaling rule (utils.py:18-19): lr = base_lr * batch / 4."""
    return base_lr * (batch_size / base_batch_size)


def init_weights(module: nn.Module):
    """utils.py:46-63: conv ~ N(0, sqrt(2/fan_out)) with fan_out = prod(kernel)*out_channels/groups, bias 0;
    BatchNorm weight 1, bias 0; Linear uniform(+-1/sqrt(fan_out)).  Works on this package's modules because every
    layer keeps a genuine ``nn.Conv*`` / ``nn.BatchNorm*`` holder at the reference's attribute path."""
    for m in module.modules():
        if isinstance(m, (nn.Conv1d, nn.Conv2d, nn.Conv3d)):
            fan_out = math.prod(m.kernel_size) * m.out_channels // m.groups
            nn.init.normal_(m.weight, 0, math.sqrt(2.0 / fan_out))
            if m.bias is not None:
                nn.init.zeros_(m.bias)
        elif isinstance(m, (nn.BatchNorm1d, nn.BatchNorm2d, nn.BatchNorm3d)):
            nn.init.ones_(m.weight)
            nn.init.zeros_(m.bias)
        elif isinstance(m, nn.Linear):
            init_range = 1.0 / math.sqrt(m.weight.size(0))
            nn.init.uniform_(m.weight, -init_range, init_range)
            if m.bias is not None:
                nn.init.zeros_(m.bias)


def get_best_model_path(dir_path, return_score: bool = False, more_better: bool = True):
    """utils.py:22-43: pick the ``*.pth`` whose trailing ``-<score>.pth`` is best."""
    scored = []
    for model_path in Path(dir_path).glob("*.pth"):
        found = re.search(r"-(\d+(?:\.\d+)?).pth", str(model_path))
        if found is not None:
            scored.append((model_path, float(found.group(0)[1:-4])))
    if not scored:
        return (None, -np.inf if more_better else np.inf) if return_score else None
    scored.sort(key=lambda x: x[1], reverse=more_better)
    return scored[0] if return_score else scored[0][0]
