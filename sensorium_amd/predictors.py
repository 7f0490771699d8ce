"""Sliding-window trial prediction (reference: src/predictors.py:20-55, scripts/predict.py:24-50, src/indexes.py).

``Predictor.predict_trial`` keeps the reference's contract — inputs ``(5, L, H, W)``, one model evaluation per end
frame ``index`` over the window ``index-behind : index+1 : step``, accumulate + divide by the overlap count
("ones" blend weights) — but evaluates ``windows_per_batch`` windows per forward (the windows are independent in
eval mode: BatchNorm uses running statistics), removes the per-window device->host sync (one copy per trial), and
accumulates on the device.  ``windows_per_batch=1`` reproduces the reference's launch pattern exactly.
"""
from __future__ import annotations

from typing import List, Optional, Sequence

import numpy as np
import torch


class IndexesGenerator:
    """Frame indexes of one window (reference: src/indexes.py:1-30)."""

    def __init__(self, size: int, step: int, position: str = "last"):
        self.size, self.step = size, step
        if position == "first":
            self.behind, self.ahead = 0, size - 1
        elif position == "middle":
            self.behind = size // 2
            self.ahead = size - self.behind - 1
        elif position == "last":
            self.behind, self.ahead = size - 1, 0
        else:
            raise ValueError("Index position value should be one of {'first', 'middle', 'last'}")
        self.behind *= step
        self.ahead *= step
        self.width = self.behind + self.ahead + 1

    def make_indexes(self, index: int) -> List[int]:
        return list(range(index - self.behind, index + self.ahead + 1, self.step))


class Predictor:
    def __init__(self, model, frame_stack_size: int = 16, frame_stack_step: int = 2, position: str = "last",
                 windows_per_batch: int = 16, use_graph: bool = False):
        """``model``: a ``sensorium_amd.argus_models.MouseModel`` (``predict(input, mouse_index)``).
        ``use_graph``: capture the eval forward of one full window batch into a hipGraph (torch.cuda.CUDAGraph) per
        (mouse, shape) and replay it — the C-ABI neither allocates nor synchronises, so the ~500 launches of one
        forward collapse into one graph launch (~270 tiny forwards per trial per model)."""
        self.model = model
        self.use_graph = bool(use_graph)
        self._graphs: dict = {}
        self.indexes_generator = IndexesGenerator(frame_stack_size, frame_stack_step, position)
        self.blend_weights = np.ones(frame_stack_size, dtype=np.float32)        # get_blend_weights("ones")
        self.windows_per_batch = max(1, int(windows_per_batch))

    @torch.no_grad()
    def predict_trial(self, inputs: torch.Tensor, mouse_index: int, num_neurons: Optional[int] = None) -> np.ndarray:
        """``inputs``: (5, L, H, W) already produced by the inputs processor (src/inputs.py). Returns (N, L) fp32."""
        gen = self.indexes_generator
        device = self.model.device
        inputs = inputs.to(device)
        length = inputs.shape[1]
        ends = list(range(gen.behind, length - gen.ahead))
        responses = None
        counts = torch.zeros(length, dtype=torch.float32, device=device)
        for i in range(0, len(ends), self.windows_per_batch):
            chunk = ends[i:i + self.windows_per_batch]
            idx = torch.tensor([gen.make_indexes(e) for e in chunk], device=device)          # [nw, size]
            windows = inputs[:, idx].permute(1, 0, 2, 3, 4).contiguous()                     # (nw, 5, size, H, W)
            if self.use_graph and len(chunk) == self.windows_per_batch:
                pred = self._graph_forward(windows, mouse_index)
            else:
                pred = self.model.predict(windows, mouse_index)                              # (nw, N, size)
            if responses is None:
                responses = torch.zeros(pred.shape[1], length, dtype=torch.float32, device=device)
            flat_idx = idx.reshape(-1)
            responses.index_add_(1, flat_idx, pred.permute(1, 0, 2).reshape(pred.shape[1], -1).float())
            counts.index_add_(0, flat_idx, torch.ones_like(flat_idx, dtype=torch.float32))
        if responses is None:
            n = num_neurons if num_neurons is not None else 0
            return np.zeros((n, length), dtype=np.float32)
        responses /= counts.clamp(min=1.0)
        return responses.cpu().numpy()


    def _graph_forward(self, windows: torch.Tensor, mouse_index: int) -> torch.Tensor:
        key = (mouse_index, tuple(windows.shape), windows.dtype)
        entry = self._graphs.get(key)
        if entry is None:
            static_in = windows.clone()
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):                      # warm-up outside the capture (allocator, lazy loads)
                self.model.predict(static_in, mouse_index)
            torch.cuda.current_stream().wait_stream(side)
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph):
                static_out = self.model.predict(static_in, mouse_index)
            entry = (graph, static_in, static_out)
            self._graphs[key] = entry
        graph, static_in, static_out = entry
        static_in.copy_(windows)
        graph.replay()
        return static_out


def ensemble_predict_trial(predictors: Sequence[Predictor], inputs: torch.Tensor, mouse_index: int) -> np.ndarray:
    """Mean over fold models (reference: scripts/predict.py:44-50)."""
    preds = [p.predict_trial(inputs, mouse_index) for p in predictors]
    return np.mean(preds, axis=0)
