"""Host-side inputs processor used by ``Predictor`` when it is built from a checkpoint path (the stored params name it:
``"inputs_processor": ("stack_inputs", {"size": (64, 64), "pad_fill_value": 0.})``, reference src/inputs.py:15-48).

Model input of one trial = 5 planes per frame: the grey-level video centre-padded to ``size`` = (width, height), then the two
behaviour traces and the two pupil-centre traces, each constant over the frame.  Built with torch ops (pad + expand + cat);
the training path never comes here — it assembles whole batches on the device (data_gpu.py / csrc/dwn_data.hip)."""
from __future__ import annotations

from typing import Callable, Dict, Sequence

import numpy as np
import torch
import torch.nn.functional as F


class StackInputsProcessor:
    def __init__(self, size: Sequence[int], pad_fill_value: float = 0):
        self.width, self.height = int(size[0]), int(size[1])
        self.size = (self.width, self.height)
        self.pad_fill_value = pad_fill_value

    def __call__(self, frames: np.ndarray, behavior: np.ndarray, pupil_center: np.ndarray) -> torch.Tensor:
        # trials are stored (H, W, L); the model wants (L, H, W)
        video = torch.from_numpy(np.ascontiguousarray(frames)).to(torch.float32).permute(2, 0, 1)
        n_frames, h, w = video.shape
        top, left = (self.height - h) // 2, (self.width - w) // 2
        if top < 0 or left < 0:
            raise ValueError(f"video {h}x{w} does not fit the {self.height}x{self.width} input")
        video = F.pad(video, (left, self.width - w - left, top, self.height - h - top), value=float(self.pad_fill_value))
        traces = torch.cat([torch.as_tensor(np.asarray(behavior)), torch.as_tensor(np.asarray(pupil_center))]).to(torch.float32)
        if traces.shape != (4, n_frames):
            raise ValueError("behavior and pupil_center must be (2, frames) each")
        planes = traces[:, :, None, None].expand(4, n_frames, self.height, self.width)
        return torch.cat([video[None], planes]).contiguous()


_PROCESSORS: Dict[str, Callable] = {"stack_inputs": StackInputsProcessor}


def get_inputs_processor(name: str, processor_params: dict):
    try:
        factory = _PROCESSORS[name]
    except KeyError:
        raise ValueError(f"inputs processor '{name}' is not supported (known: {sorted(_PROCESSORS)})") from None
    return factory(**processor_params)
