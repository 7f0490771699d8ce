"""MouseModel: the train / val / predict step of the reference's ``src/argus_models.py`` (an ``argus.Model``
subclass) restated without the third-party ``argus`` engine (not installable offline).

Same constructor contract (``params`` dict with ``nn_module``/``loss``/``optimizer``/``device``/``amp``/
``iter_size`` exactly as configs/true_batch_001.py:20-61), same method names, arguments, return dict and
numerics-relevant order of operations (argus_models.py:43-99):

    train_step: train() -> zero_grad -> per chunk [to device, autocast, distill targets, forward, loss/iter_size,
                backward, loss.item()] -> optimizer step -> EMA update -> {'prediction','target','loss'}

MI355X-specific differences, none of which change results: autocast selects bf16 (no GradScaler needed — the
scaler object is kept, disabled, for API compatibility); AdamW is the fused multi-tensor HIP kernel with the
parameter EMA folded in; the distillation target fill (argus_models.py:37-41) is vectorised with ``where``
instead of a python loop over ``argwhere``; under ``torch.distributed`` gradients are all-reduced through
``GradBuckets`` while backward is still running.
"""
from __future__ import annotations

import logging
import os
from typing import Optional

import torch
import torch.distributed as dist

from .ddp import GradBuckets
from .dwiseneuro import DwiseNeuro
from .ema import ModelEma
from .engine import Model
from .losses import MicePoissonLoss
from .optim import FusedAdamWEma


def deep_to(obj, device, non_blocking: bool = False):
    if torch.is_tensor(obj):
        return obj.to(device, non_blocking=non_blocking)
    if isinstance(obj, (list, tuple)):
        return type(obj)(deep_to(o, device, non_blocking) for o in obj)
    if isinstance(obj, dict):
        return {k: deep_to(v, device, non_blocking) for k, v in obj.items()}
    return obj


def deep_detach(obj):
    if torch.is_tensor(obj):
        return obj.detach()
    if isinstance(obj, (list, tuple)):
        return type(obj)(deep_detach(o) for o in obj)
    return obj


def deep_chunk(obj, chunks: int):
    if chunks == 1:
        return [obj]
    if torch.is_tensor(obj):
        return list(torch.chunk(obj, chunks, dim=0))
    if isinstance(obj, (list, tuple)):
        parts = [deep_chunk(o, chunks) for o in obj]
        return [type(obj)(p[i] for p in parts) for i in range(chunks)]
    raise TypeError(type(obj))


@torch.no_grad()
def fill_distill_targets(distill_prediction, target, distill_ratio: float):
    """In-place soft-label fill of argus_models.py:35-41: every (sample, mouse) pair with weight 0 gets the
    teacher's prediction as target and the weight ``r/(1-r) * sum(w) / #zeros``.  Vectorised with ``where``
    (the reference loops over ``argwhere`` in python, ~288 iterations per step)."""
    target_tensors, mice_weights = target
    distill_mask = mice_weights == 0.0
    distill_weight = distill_ratio / (1.0 - distill_ratio) * mice_weights.sum() / distill_mask.sum()
    for m, pred in enumerate(distill_prediction):
        sel = distill_mask[:, m]
        target_tensors[m].copy_(torch.where(sel[:, None, None], pred.to(target_tensors[m].dtype), target_tensors[m]))
    mice_weights.copy_(torch.where(distill_mask, distill_weight.to(mice_weights.dtype), mice_weights))


class MouseModel(Model):
    """``fit`` / ``validate`` / ``save`` come from ``engine.Model`` (the argus surface train.py:141-145 uses);
    ``load_model`` finds this class by the ``model_name`` stored in the checkpoint."""
    nn_module = {"dwiseneuro": DwiseNeuro}
    loss = {"mice_poisson": MicePoissonLoss}
    optimizer = {"AdamW": FusedAdamWEma}

    def __init__(self, params: dict):
        self.params = params
        self.logger = logging.getLogger("sensorium_amd")
        name, kwargs = params["nn_module"]
        self.device = torch.device(params.get("device", "cuda:0"))
        self.nn_module = MouseModel.nn_module[name](**kwargs).to(self.device)
        loss_spec = params.get("loss", ("mice_poisson", {}))
        self.loss = None if loss_spec is None else MouseModel.loss[loss_spec[0]](**loss_spec[1])
        self.iter_size = int(params.get("iter_size", 1))
        self.amp = bool(params.get("amp", False))
        self.grad_scaler = torch.amp.GradScaler("cuda", enabled=False)   # bf16 needs no loss scaling
        self._model_ema: Optional[ModelEma] = None
        self.distill_model: Optional[torch.nn.Module] = None
        self.distill_ratio: float = 0.0
        self._opt_spec = params.get("optimizer", ("AdamW", {"lr": 1e-3}))
        self.optimizer = None
        self.buckets: Optional[GradBuckets] = None
        self.prediction_transform = lambda x: x

    # -- setup ------------------------------------------------------------------------------------------
    @property
    def model_ema(self) -> Optional[ModelEma]:
        return self._model_ema

    @model_ema.setter
    def model_ema(self, ema: Optional[ModelEma]):
        """Plain attribute assignment as in the reference (scripts/train.py:53 ``model.model_ema = ModelEma(...)``), at
        any time: an optimizer that already exists keeps its Adam moments and step counts and is re-bound to the new EMA
        copies (or detached when ``ema`` is None)."""
        self._model_ema = ema
        if ema is not None and hasattr(ema.ema, "set_fp32_eval_products"):        # the copy follows the trained module's mode
            ema.ema.set_fp32_eval_products(getattr(self.nn_module, "fp32_eval_products", "bf16x3"))
        if self.optimizer is not None:
            self._bind_ema_to_optimizer()

    def _bind_ema_to_optimizer(self):
        ema = self._model_ema
        if ema is None:
            self.optimizer.bind_ema(None, self.optimizer.ema_decay)
            return
        if self.buckets is not None:
            self.buckets.adopt_ema(ema.ema)       # sharded readout buckets: EMA copies laid out like the parameters
        by_name = dict(ema.ema.named_parameters())
        ema_params = [by_name[n] for n, p in self.nn_module.named_parameters() if p.requires_grad]
        self.optimizer.bind_ema(ema_params, ema.decay, owner=ema)

    def set_ema(self, decay: float):
        """train.py:53; the parameter EMA rides in the optimizer kernel (bound now if the optimizer exists, else when it
        is built)."""
        self.model_ema = ModelEma(self.nn_module, decay=decay)

    def set_fp32_eval_products(self, mode: str = "bf16x3"):
        """DwiseNeuro.set_fp32_eval_products on BOTH networks val_step / predict may evaluate: the trained module and the EMA
        copy (a deepcopy taken when the EMA was set: a mode set on ``nn_module`` afterwards would not reach it)."""
        self.nn_module.set_fp32_eval_products(mode)
        if self._model_ema is not None:
            self._model_ema.ema.set_fp32_eval_products(mode)

    def get_optimizer(self):
        self._ensure_optimizer()
        return self.optimizer

    def _ensure_optimizer(self):
        if self.optimizer is not None:
            return
        if self._opt_spec is None:
            raise RuntimeError("model has no optimizer (loaded with optimizer=None)")
        oname, okwargs = self._opt_spec
        params = [p for p in self.nn_module.parameters() if p.requires_grad]
        # params["ddp_single_rank"]: the exchange machinery on a ONE-rank process group as well (tests: RCCL on a one-GPU box)
        single = bool(self.params.get("ddp_single_rank", False))
        distributed = dist.is_available() and dist.is_initialized() and (dist.get_world_size() > 1 or single)
        if distributed:
            # broadcasts rank 0's parameters and buffers; params["ddp_comm_dtype"] / DWN_DDP_COMM=bf16: exchange bf16 copies
            comm = self.params.get("ddp_comm_dtype", os.environ.get("DWN_DDP_COMM"))
            comm = torch.bfloat16 if comm in (torch.bfloat16, "bf16", "bfloat16") else None
            # params["ddp_shard_optimizer"] / DWN_DDP_SHARD=1: reduce-scatter + sharded AdamW/EMA + all-gather for the readouts
            shard = self.params.get("ddp_shard_optimizer", os.environ.get("DWN_DDP_SHARD", "0") == "1")
            self.buckets = GradBuckets(self.nn_module, comm_dtype=comm, shard_optional=bool(shard), single_rank=single)
            if self._model_ema is not None:                   # ... so the EMA copy taken earlier must follow (val_step uses it)
                self._model_ema.set(self.nn_module)
        self.optimizer = MouseModel.optimizer[oname](params, **okwargs)
        if self.buckets is not None and self.buckets.shard:
            self.optimizer.set_shard_map(self.buckets.owned_range)
        if self._model_ema is not None:
            self._bind_ema_to_optimizer()
        pending = getattr(self, "_pending_optimizer_state", None)
        if pending is not None:                               # load_model(..., optimizer_state) met a lazily built optimizer
            self.optimizer.load_state_dict(pending)
            self._pending_optimizer_state = None

    def train(self):
        self.nn_module.train()

    def eval(self):
        self.nn_module.eval()

    # -- argus_models.py:31-41 -----------------------------------------------------------------------------
    @torch.no_grad()
    def add_distill_predictions(self, input, target):
        if self.distill_model is None or not self.distill_ratio:
            return
        fill_distill_targets(self.distill_model(input), target, self.distill_ratio)

    def _active_samples(self, batch):
        """Per mouse, the device index tensor of the samples whose loss weight is not zero — or None when that is not known
        WITHOUT a device read-back (weights already on the device with no host copy attached) or when it is everybody
        (distillation fills every weight, argus_models.py:37-41).  The readouts' backward then skips the other rows: their
        gradient is exactly zero (losses.py:15-17), and with ten one-hot mice that is 28 of 32 samples per readout."""
        if self.distill_model is not None and self.distill_ratio:
            return None
        try:
            weights = batch[1][1]
        except (TypeError, IndexError, KeyError):
            return None
        if not torch.is_tensor(weights) or weights.dim() != 2 or weights.shape[1] != len(self.nn_module.readouts):
            return None
        host = weights if not weights.is_cuda else getattr(weights, "_dwn_host", None)
        if host is None or host.shape != weights.shape:
            return None
        if weights.is_cuda and getattr(weights, "_dwn_host_version", weights._version) != weights._version:
            return None        # the device weights were edited in place after the host copy was attached: dense backward
        key = (id(host), host._version, str(self.device))
        cached = getattr(self, "_active_cache", None)
        if cached is not None and cached[0] == key:
            return cached[1]
        mask = host != 0
        if bool(mask.all()):
            out = None
        else:
            counts = mask.sum(0).tolist()
            order = torch.cat([mask[:, m].nonzero().flatten() for m in range(mask.shape[1])]).to(self.device)
            out, lo = [], 0
            for c in counts:
                out.append(order[lo:lo + c])
                lo += c
        self._active_cache = (key, out, host)          # (host kept alive: its id is part of the key)
        return out

    # -- argus_models.py:43-71 -----------------------------------------------------------------------------
    def train_step(self, batch, state=None, sync_loss: bool = True) -> dict:
        self._ensure_optimizer()
        self.train()
        if self.buckets is not None:
            self.buckets.zero_grad(self.iter_size)
        else:
            self.optimizer.zero_grad(set_to_none=True)
        loss_value = 0
        for chunk_batch in deep_chunk(batch, self.iter_size):
            active = self._active_samples(chunk_batch)
            input, target = deep_to(chunk_batch, self.device, non_blocking=True)
            for m, readout in enumerate(self.nn_module.readouts):
                readout._dwn_active = None if active is None else active[m]
            with torch.autocast("cuda", dtype=torch.bfloat16, enabled=self.amp):
                self.add_distill_predictions(input, target)
                prediction = self.nn_module(input)
                loss = self.loss(prediction, target)
                loss = loss / self.iter_size
            loss.backward()
            for readout in self.nn_module.readouts:
                readout._dwn_active = None
            loss_value = loss_value + (loss.item() if sync_loss else loss.detach())
        if self.buckets is not None:
            self.buckets.finish()
        self.optimizer.step()
        if self.buckets is not None and self.buckets.shard:
            self.buckets.gather_params()          # awaited by the readouts' forward pre-hook: hidden behind the next core forward
        if self.model_ema is not None:
            # the fused AdamW kernel has already lerped the parameter copies iff it is bound to THIS ModelEma
            folded = self.optimizer.folds_ema_of(self.model_ema)
            if not folded and self.buckets is not None and self.buckets.shard:
                self.buckets.wait_params()        # the lerp below reads every parameter: the all-gather must have landed
            self.model_ema.update(self.nn_module, skip_parameters=folded)
        return {"prediction": self.prediction_transform(deep_detach(prediction)), "target": deep_detach(target),
                "loss": loss_value}

    def sync_for_read(self):
        """Sharded optimizer: make every rank's copy of the parameters and of the EMA network complete (the slices other ranks
        update arrive by all-gather) before they are evaluated or written to a checkpoint.  A collective: every rank calls it."""
        if self.buckets is not None and self.buckets.shard:
            self.buckets.wait_params()
            if self.model_ema is not None:
                self.buckets.gather_ema()

    def needs_sync(self) -> bool:
        """Sharded optimizer: True while this rank's copy of the parameters / EMA network is incomplete, i.e. until EVERY
        rank has called ``sync_for_read()`` after the last training step."""
        b = self.buckets
        if b is None or not b.shard:
            return False
        # the EMA slices only matter while an EMA network exists (gather_params marks them dirty after every step, and only
        # gather_ema / adopt_ema clear the mark: without an EMA nothing ever would)
        return bool(b._param_handles) or (self._model_ema is not None and b.ema_dirty)

    def _eval_module(self):
        self.sync_for_read()
        return self.nn_module if self.model_ema is None else self.model_ema.ema

    # -- argus_models.py:73-87 -----------------------------------------------------------------------------
    def val_step(self, batch, state=None) -> dict:
        self.eval()
        with torch.no_grad():
            input, target = deep_to(batch, self.device, non_blocking=True)
            module = self._eval_module()
            prediction = module(input)
            loss = self.loss(prediction, target)
            return {"prediction": self.prediction_transform(prediction), "target": target, "loss": loss.item()}

    # -- argus_models.py:89-99 -----------------------------------------------------------------------------
    def predict(self, input, mouse_index: Optional[int] = None):
        with torch.no_grad():
            self.eval()
            input = deep_to(input, self.device)
            module = self._eval_module()
            return self.prediction_transform(module(input, mouse_index))
