"""Inputs processors (reference: src/inputs.py:9-48).  ``stack_inputs`` builds the model input of one trial: channel 0 the
video centre-padded to ``size`` (width, height), channels 1-2 behavior, 3-4 pupil_center broadcast over the frame.
Host-side (numpy) like the reference; the training path assembles batches on the device instead (data_gpu.py)."""
from __future__ import annotations

import numpy as np
import torch


class StackInputsProcessor:
    def __init__(self, size, pad_fill_value: int = 0):
        self.size = tuple(size)
        self.pad_fill_value = pad_fill_value

    def __call__(self, frames: np.ndarray, behavior: np.ndarray, pupil_center: np.ndarray) -> torch.Tensor:
        length = frames.shape[-1]
        out = np.full((5, length, self.size[1], self.size[0]), self.pad_fill_value, dtype=np.float32)
        video = np.transpose(frames.astype(np.float32), (2, 0, 1))             # (H, W, L) on disk -> (L, H, W)
        h, w = video.shape[-2:]
        h0, w0 = (self.size[1] - h) // 2, (self.size[0] - w) // 2
        out[0, :, h0:h0 + h, w0:w0 + w] = video
        out[1:3] = behavior[:, :, None, None]
        out[3:] = pupil_center[:, :, None, None]
        return torch.from_numpy(out)


_REGISTRY = {"stack_inputs": StackInputsProcessor}


def get_inputs_processor(name: str, processor_params: dict):
    if name not in _REGISTRY:
        raise ValueError(f"inputs processor '{name}' is not supported (known: {sorted(_REGISTRY)})")
    return _REGISTRY[name](**processor_params)
