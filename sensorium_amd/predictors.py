"""Sliding-window trial prediction (reference: src/predictors.py:20-55, scripts/predict.py:24-50, src/indexes.py).

``Predictor`` keeps the reference's surface — ``Predictor(model_path, device, blend_weights)``,
``predict_trial(video, behavior, pupil_center, mouse_index)``, one model evaluation per end frame ``index`` over the window
``index-behind : index+1 : step``, accumulate + divide by the accumulated blend weights — but evaluates
``windows_per_batch`` windows per forward (the windows are independent in eval mode: BatchNorm uses running statistics),
removes the per-window device->host sync (one copy per trial) and accumulates on the device in a fixed order.
``windows_per_batch=1`` reproduces the reference's launch pattern.  ``EnsemblePredictor`` runs all fold models of
scripts/predict.py:44-50 inside one forward / one captured hipGraph per window batch.
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


def get_blend_weights(name: str, size: int) -> np.ndarray:
    """src/predictors.py:13-19"""
    if name == "ones":
        return np.ones(size, dtype=np.float32)
    if name == "linear":
        return np.linspace(0, 1, num=size).astype(np.float32)
    raise ValueError(f"Blend weights '{name}' is not supported")


class Predictor:
    def __init__(self, model, device: str = "cuda:0", blend_weights: str = "ones", *, frame_stack_size: Optional[int] = None,
                 frame_stack_step: Optional[int] = None, position: str = "last", windows_per_batch: int = 32,
                 use_graph: bool = False, fp32_products: Optional[str] = None):
        """``model``: a checkpoint path — the reference's constructor ``Predictor(model_path, device, blend_weights)``
        (src/predictors.py:22-34: ``load_model(path, device=device, optimizer=None, loss=None)``, frame stack and inputs
        processor read from the stored params) — or an already built ``MouseModel`` together with ``frame_stack_size`` /
        ``frame_stack_step``.
        ``fp32_products``: ``"native"`` / ``"bf16x3"`` — how an fp32 model multiplies in this predictor's forwards
        (``DwiseNeuro.set_fp32_eval_products``; ``None`` keeps the model's setting, whose default is "bf16x3").
        ``windows_per_batch``: windows evaluated per forward (1 = the reference's launch pattern).
        ``use_graph``: capture the eval forward of one full window batch into a hipGraph (torch.cuda.CUDAGraph) per
        (mouse, shape) and replay it — the C-ABI neither allocates nor synchronises, so the ~500 launches of one
        forward collapse into one graph launch (~270 tiny forwards per trial per model)."""
        self.inputs_processor = None
        if isinstance(model, (str, bytes)) or hasattr(model, "__fspath__"):
            from .engine import load_model
            from .inputs import get_inputs_processor
            model = load_model(model, device=device, optimizer=None, loss=None)
            params = model.params
            if "inputs_processor" in params:
                self.inputs_processor = get_inputs_processor(*params["inputs_processor"])
            fs = params.get("frame_stack", {})
            frame_stack_size = fs.get("size", frame_stack_size)
            frame_stack_step = fs.get("step", frame_stack_step)
            position = fs.get("position", position)
            if position != "last":
                raise ValueError("Predictor: only frame_stack position 'last' is supported (src/predictors.py:29)")
            rp = params.get("responses_processor", ("identity", {}))
            if rp[0] != "identity":
                raise ValueError("Predictor: only the identity responses processor is supported (src/predictors.py:30)")
        if frame_stack_size is None or frame_stack_step is None:
            frame_stack_size, frame_stack_step = frame_stack_size or 16, frame_stack_step or 2
        self.model = model
        self.model.eval()
        if fp32_products is not None:
            for net in (self.model.nn_module, getattr(getattr(self.model, "model_ema", None), "ema", None)):
                if net is not None:
                    net.set_fp32_eval_products(fp32_products)
        self.use_graph = bool(use_graph)
        self._graphs: dict = {}
        self.frame_stack_size, self.frame_stack_step = int(frame_stack_size), int(frame_stack_step)
        self.indexes_generator = IndexesGenerator(self.frame_stack_size, self.frame_stack_step, position)
        self.blend_weights = get_blend_weights(blend_weights, self.frame_stack_size)
        self.windows_per_batch = max(1, int(windows_per_batch))

    # ---- one window batch: (nw, 5, size, H, W) -> (nw, N, size)
    def _forward(self, windows: torch.Tensor, mouse_index: int) -> torch.Tensor:
        return self.model.predict(windows, mouse_index)

    def _make_inputs(self, video, behavior, pupil_center, mouse_index):
        """Accepts the reference's call ``predict_trial(video, behavior, pupil_center, mouse_index)`` (numpy arrays in the
        on-disk layout, src/predictors.py:36-41) and the pre-processed form ``predict_trial(inputs, mouse_index)`` with
        ``inputs`` = (5, L, H, W) as produced by an inputs processor."""
        if pupil_center is None and (mouse_index is None or behavior is None):
            # pre-processed form: predict_trial(inputs, k) or predict_trial(inputs, mouse_index=k)
            index = behavior if mouse_index is None else mouse_index
            if index is None:
                raise TypeError("predict_trial: mouse_index is required")
            return (video if torch.is_tensor(video) else torch.from_numpy(np.asarray(video))), int(index)
        if mouse_index is None:
            raise TypeError("predict_trial: mouse_index is required")
        if self.inputs_processor is None:
            raise RuntimeError("predict_trial(video, behavior, pupil_center, ...) needs an inputs processor: build the "
                               "Predictor from a checkpoint path or set predictor.inputs_processor")
        return self.inputs_processor(video, behavior, pupil_center), int(mouse_index)

    @torch.no_grad()
    def predict_trial(self, video, behavior=None, pupil_center=None, mouse_index: Optional[int] = None,
                      num_neurons: Optional[int] = None) -> np.ndarray:
        """Returns (N, L) fp32: every window ending at frame ``index`` adds its prediction to the frames it covers, and the
        sum is divided by the accumulated blend weights (src/predictors.py:43-55; like the reference, the predictions themselves
        are not multiplied by the blend weights).  Accumulation happens on the device, one pass per window position (frames
        are distinct within a pass: the blend itself is order-independent), one device-to-host copy per trial.
        Reproducibility: the eval forward is bit-reproducible from call to call — the SqueezeExcite pooling sums inside the
        temporal kernel are 64-bit fixed-point integer adds (any arrival order, same sum) and the blend accumulates in a fixed
        order."""
        inputs, mouse_index = self._make_inputs(video, behavior, pupil_center, mouse_index)
        gen = self.indexes_generator
        device = self.model.device
        inputs = inputs.to(device)
        length = inputs.shape[1]
        ends = list(range(gen.behind, length - gen.ahead))
        responses = None
        counts = torch.zeros(length, dtype=torch.float32, device=device)
        bw = torch.from_numpy(self.blend_weights).to(device)
        for i in range(0, len(ends), self.windows_per_batch):
            chunk = ends[i:i + self.windows_per_batch]
            idx = torch.tensor([gen.make_indexes(e) for e in chunk], device=device)          # [nw, size]
            windows = inputs[:, idx].permute(1, 0, 2, 3, 4).contiguous()                     # (nw, 5, size, H, W)
            if self.use_graph and len(chunk) == self.windows_per_batch:
                pred = self._graph_forward(windows, mouse_index)
            else:
                pred = self._forward(windows, mouse_index)                                   # (nw, N, size)
            if responses is None:
                responses = torch.zeros(pred.shape[1], length, dtype=torch.float32, device=device)
            pred = pred.float()
            for j in range(idx.shape[1]):                 # window position j: the nw frames idx[:, j] are distinct
                responses.index_add_(1, idx[:, j], pred[:, :, j].t())
                counts.index_add_(0, idx[:, j], bw[j].expand(idx.shape[0]))
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
                self._forward(static_in, mouse_index)
            torch.cuda.current_stream().wait_stream(side)
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph):
                static_out = self._forward(static_in, mouse_index)
            entry = (graph, static_in, static_out)
            self._graphs[key] = entry
        graph, static_in, static_out = entry
        static_in.copy_(windows)
        graph.replay()
        return static_out


class EnsemblePredictor(Predictor):
    """The fold ensemble of scripts/predict.py:44-50 (mean of the fold models' trial predictions) with every fold model
    evaluated on the same window batch inside ONE forward / ONE captured hipGraph: the window gather, the host loop and the
    blend are paid once instead of once per fold, and the mean over models is taken on the device (the blend is linear, so
    mean-then-blend equals the reference's blend-then-mean up to summation order)."""

    def __init__(self, models: Sequence, device: str = "cuda:0", blend_weights: str = "ones", **kw):
        models = list(models)
        if not models:
            raise ValueError("EnsemblePredictor: at least one model")
        first = Predictor(models[0], device, blend_weights, **kw)
        self.__dict__.update(first.__dict__)
        self._graphs = {}
        members = [first]
        for m in models[1:]:
            p = Predictor(m, device, blend_weights, **kw)
            if (p.frame_stack_size, p.frame_stack_step) != (first.frame_stack_size, first.frame_stack_step):
                raise ValueError("EnsemblePredictor: fold models disagree on the frame stack")
            members.append(p)
        self.models = [p.model for p in members]

    def _forward(self, windows: torch.Tensor, mouse_index: int) -> torch.Tensor:
        acc = None
        for m in self.models:
            pr = m.predict(windows, mouse_index).float()
            acc = pr if acc is None else acc + pr
        return acc / float(len(self.models))


def ensemble_predict_trial(predictors: Sequence[Predictor], inputs, mouse_index: int) -> np.ndarray:
    """Mean over fold models, one predictor after the other (reference: scripts/predict.py:44-50)."""
    preds = [p.predict_trial(inputs, mouse_index) for p in predictors]
    return np.mean(preds, axis=0)
