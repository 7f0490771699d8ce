"""Minimal train/validate runtime with the surface the reference uses from the third-party ``pytorch-argus`` 1.0.0
(requirements.txt:7; not vendored, not installable offline — SURVEY.md §8f rank 1).

Only what ``scripts/train.py:41-146``, ``scripts/predict.py`` / ``src/predictors.py:24`` and ``src/ema.py:60-72`` touch is
restated: ``Model`` (params -> nn_module / loss / optimizer / device, ``fit``, ``validate``, ``save``), ``State``,
``Callback`` events, ``Metric`` + the default ``Loss`` metric, and ``load_model``.  Behaviour follows argus' published
semantics: epochs are numbered from 1, ``state.iteration`` restarts every epoch, a validation pass runs before the first
training epoch and after every training epoch, validation metrics are merged into the training state under a ``val_``
prefix *before* the user's epoch-complete callbacks (checkpoint names such as ``model-{epoch:03d}-{val_corr:.6f}.pth``
rely on that order, train.py:129-131).  argus itself is absent from /root/reference, so this file is pinned only by the
reference's call sites (parity unpinned by golden vectors; the numerics live in ``MouseModel`` which *is* pinned).

Host logic only: nothing here computes on tensors besides moving metrics to python floats.
"""
from __future__ import annotations

import logging
from pathlib import Path
from typing import Callable, Dict, Iterable, List, Optional

import torch

__all__ = ["State", "Callback", "Metric", "Loss", "Model", "load_model", "FunctionCallback", "on_epoch_complete"]

_EVENTS = ("start", "complete", "epoch_start", "epoch_complete", "iteration_start", "iteration_complete",
           "catch_exception")


class State:
    """Mutable record handed to steps, metrics and callbacks (argus.engine.State)."""

    def __init__(self, model=None, phase: str = "", logger: Optional[logging.Logger] = None):
        self.model = model
        self.phase = phase
        self.logger = logger or logging.getLogger("sensorium_amd")
        self.iteration = 0
        self.epoch = 0
        self.stopped = False
        self.batch = None
        self.step_output = None
        self.data_loader = None
        self.exception: Optional[BaseException] = None
        self.metrics: Dict[str, float] = {}

    def update(self, **kwargs):
        for k, v in kwargs.items():
            setattr(self, k, v)


class Callback:
    """Override any of: start, complete, epoch_start, epoch_complete, iteration_start, iteration_complete,
    catch_exception — each receives the ``State``."""

    def attach(self, engine: "_Engine"):
        for ev in _EVENTS:
            fn = getattr(self, ev, None)
            if callable(fn):
                engine.handlers[ev].append(fn)


class FunctionCallback(Callback):
    def __init__(self, event: str, fn: Callable):
        assert event in _EVENTS
        setattr(self, event, fn)


def on_epoch_complete(fn: Callable) -> FunctionCallback:
    return FunctionCallback("epoch_complete", fn)


class Metric(Callback):
    """argus.metrics.Metric: reset at epoch start, ``update(step_output)`` per iteration, ``compute()`` at epoch end
    into ``state.metrics[phase_ + name]``."""
    name: str = ""
    better: str = "min"

    def reset(self):
        pass

    def update(self, step_output: dict):
        pass

    def compute(self):
        raise NotImplementedError

    def epoch_start(self, state: State):
        self.reset()

    def iteration_complete(self, state: State):
        self.update(state.step_output)

    def epoch_complete(self, state: State):
        with torch.no_grad():
            value = self.compute()
        prefix = f"{state.phase}_" if state.phase else ""
        state.metrics[prefix + self.name] = value


class Loss(Metric):
    """Mean of ``step_output['loss']`` over the epoch's iterations -> ``train_loss`` / ``val_loss``."""
    name = "loss"
    better = "min"

    def __init__(self):
        self.total, self.count = 0.0, 0

    def reset(self):
        self.total, self.count = 0.0, 0

    def update(self, step_output: dict):
        self.total += float(step_output["loss"])
        self.count += 1

    def compute(self):
        if self.count == 0:
            raise RuntimeError("Loss metric: no iterations in this epoch")
        return self.total / self.count


class _Engine:
    def __init__(self, step_function: Callable, state: State):
        self.step_function = step_function
        self.state = state
        self.handlers: Dict[str, List[Callable]] = {ev: [] for ev in _EVENTS}

    def raise_event(self, event: str):
        for fn in self.handlers[event]:
            fn(self.state)

    def run(self, data_loader: Iterable, start_epoch: int = 0, end_epoch: int = 1) -> State:
        st = self.state
        st.update(data_loader=data_loader, epoch=start_epoch, iteration=0, stopped=False, exception=None)
        try:
            self.raise_event("start")
            while st.epoch < end_epoch and not st.stopped:
                st.iteration = 0
                st.metrics = {}
                st.epoch += 1
                self.raise_event("epoch_start")
                for batch in data_loader:
                    st.batch = batch
                    st.iteration += 1
                    self.raise_event("iteration_start")
                    st.step_output = self.step_function(batch, st)
                    self.raise_event("iteration_complete")
                    st.step_output = None
                    if st.stopped:
                        break
                st.batch = None
                self.raise_event("epoch_complete")
            self.raise_event("complete")
        except BaseException as exc:       # argus: give callbacks a chance (checkpoint after exception), then re-raise
            st.exception = exc
            self.raise_event("catch_exception")
            raise
        finally:
            st.update(batch=None, step_output=None, data_loader=None)
        return st


class _DefaultLogging(Callback):
    """argus' default epoch line: ``train - epoch: 3, lr: 0.0024, train_loss: ...`` / ``val - epoch: 3, val_loss: ...``."""

    def epoch_complete(self, state: State):
        parts = [f"{state.phase} - epoch: {state.epoch}"]
        if state.phase == "train" and state.model is not None:
            lr = state.model.get_lr()
            parts.append("lr: " + (", ".join(f"{v:.8g}" for v in lr) if isinstance(lr, (list, tuple)) else f"{lr:.8g}"))
        prefix = f"{state.phase}_"
        for k, v in state.metrics.items():
            if k.startswith(prefix):
                parts.append(f"{k}: {_fmt(v)}")
        state.logger.info(", ".join(parts))


def _fmt(v) -> str:
    try:
        return f"{float(v):.7g}"
    except (TypeError, ValueError):
        return str(v)


_MODEL_REGISTRY: Dict[str, type] = {}


def _pick(registry, spec, what: str):
    """argus component resolution: the class attribute is either a class (params = kwargs dict) or a dict
    name -> class (params = ``(name, kwargs)``)."""
    if isinstance(registry, dict):
        if not (isinstance(spec, (tuple, list)) and len(spec) == 2):
            raise ValueError(f"params['{what}'] must be (name, kwargs) — got {spec!r}")
        name, kwargs = spec
        if name not in registry:
            raise ValueError(f"unknown {what} '{name}'; known: {sorted(registry)}")
        return registry[name], dict(kwargs)
    return registry, dict(spec or {})


class Model:
    """argus.Model: ``params`` = {'nn_module': (name, kw), 'loss': (name, kw), 'optimizer': (name, kw), 'device': str}.

    Subclasses set the class attributes ``nn_module`` / ``loss`` / ``optimizer`` (class or {name: class}) and
    implement ``train_step(batch, state)`` / ``val_step(batch, state)`` returning ``{'prediction','target','loss'}``.
    """
    nn_module = None
    loss = None
    optimizer = None
    prediction_transform = None

    def __init_subclass__(cls, **kw):
        super().__init_subclass__(**kw)
        _MODEL_REGISTRY[cls.__name__] = cls

    def __init__(self, params: dict):
        self.params = params
        self.logger = logging.getLogger("sensorium_amd")
        self.device = torch.device(params.get("device", "cpu"))
        cls = type(self)
        mod_cls, mod_kw = _pick(cls.nn_module, params["nn_module"], "nn_module")
        self.nn_module = mod_cls(**mod_kw).to(self.device)
        self.loss = None
        if cls.loss is not None and params.get("loss") is not None:
            loss_cls, loss_kw = _pick(cls.loss, params["loss"], "loss")
            self.loss = loss_cls(**loss_kw)
        self.optimizer = None
        if cls.optimizer is not None and params.get("optimizer") is not None:
            self.optimizer = self._build_optimizer()
        self.prediction_transform = (lambda x: x)

    # -- components --------------------------------------------------------------------------------------------
    def _build_optimizer(self):
        opt_cls, opt_kw = _pick(type(self).optimizer, self.params["optimizer"], "optimizer")
        return opt_cls([p for p in self.nn_module.parameters() if p.requires_grad], **opt_kw)

    def get_optimizer(self):
        if self.optimizer is None:
            raise RuntimeError("model has no optimizer (built with optimizer=None)")
        return self.optimizer

    def get_lr(self):
        groups = self.get_optimizer().param_groups
        lrs = [g["lr"] for g in groups]
        return lrs[0] if len(lrs) == 1 else lrs

    def set_lr(self, lr):
        groups = self.get_optimizer().param_groups
        lrs = lr if isinstance(lr, (list, tuple)) else [lr] * len(groups)
        if len(lrs) != len(groups):
            raise ValueError("set_lr: one value per param group expected")
        for g, v in zip(groups, lrs):
            g["lr"] = v

    def train(self):
        self.nn_module.train()

    def eval(self):
        self.nn_module.eval()

    def get_nn_module(self):
        return self.nn_module

    # -- steps (overridden) ----------------------------------------------------------------------------------
    def train_step(self, batch, state: State) -> dict:
        raise NotImplementedError

    def val_step(self, batch, state: State) -> dict:
        raise NotImplementedError

    # -- loops ---------------------------------------------------------------------------------------------------
    def _engine(self, step, phase: str, metrics: List[Metric], callbacks) -> _Engine:
        eng = _Engine(step, State(model=self, phase=phase, logger=self.logger))
        for m in metrics:
            m.attach(eng)
        return eng

    def fit(self, train_loader, val_loader=None, num_epochs: int = 1, metrics: Optional[List[Metric]] = None,
            metrics_on_train: bool = False, callbacks: Optional[List[Callback]] = None,
            val_callbacks: Optional[List[Callback]] = None) -> State:
        if self.loss is None:
            raise RuntimeError("fit: model has no loss")
        self.get_optimizer()
        metrics = list(metrics or [])
        train_metrics: List[Metric] = [Loss()] + (metrics if metrics_on_train else [])
        train_engine = self._engine(self.train_step, "train", train_metrics, callbacks)
        if val_loader is not None:
            val_engine = self._engine(self.val_step, "val", [Loss()] + metrics, val_callbacks)
            _DefaultLogging().attach(val_engine)
            for cb in (val_callbacks or []):
                cb.attach(val_engine)

            def validation_epoch(train_state: State):
                epoch = train_state.epoch
                val_engine.run(val_loader, epoch - 1, epoch)
                train_state.metrics.update(val_engine.state.metrics)

            on_epoch_complete(validation_epoch).attach(train_engine)      # before the user's callbacks
            val_engine.run(val_loader, -1, 0)                             # argus validates once before training
        _DefaultLogging().attach(train_engine)
        for cb in (callbacks or []):
            cb.attach(train_engine)
        return train_engine.run(train_loader, 0, num_epochs)

    def validate(self, val_loader, metrics: Optional[List[Metric]] = None,
                 callbacks: Optional[List[Callback]] = None) -> Dict[str, float]:
        if self.loss is None:
            raise RuntimeError("validate: model has no loss")
        eng = self._engine(self.val_step, "val", [Loss()] + list(metrics or []), callbacks)
        _DefaultLogging().attach(eng)
        for cb in (callbacks or []):
            cb.attach(eng)
        return dict(eng.run(val_loader, 0, 1).metrics)

    # -- persistence (argus file format: ema.py:63-72) -------------------------------------------------------
    def _require_synced(self, what: str):
        """Writing is rank-local (usually rank 0 alone), so it must not start a collective.  With the sharded optimizer the
        slices other ranks own arrive through ``sync_for_read()``, which EVERY rank has to call first (``Checkpoint`` does,
        on all ranks, before its writer-only part)."""
        needs = getattr(self, "needs_sync", None)
        if needs is not None and needs():
            raise RuntimeError(f"{what}: with the sharded optimizer (ddp_shard_optimizer) call model.sync_for_read() on EVERY "
                               "rank after the last training step, then save on the writer rank")

    def state_dict_for_save(self):
        self._require_synced("state_dict_for_save")
        return {k: v.detach().to("cpu") for k, v in self.nn_module.state_dict().items()}

    def save(self, file_path, optimizer_state: bool = False):
        state = {"model_name": type(self).__name__, "params": self.params,
                 "nn_state_dict": self.state_dict_for_save()}
        if optimizer_state and self.optimizer is not None:
            buckets = getattr(self, "buckets", None)
            if buckets is not None and buckets.shard:
                raise RuntimeError("save(optimizer_state=True): the sharded optimizer keeps only this rank's 1/N slice of the readout "
                                   "moments; a checkpoint of it would silently drop the rest — save without optimizer state "
                                   "(the reference never stores it, src/ema.py:67-72) or train with ddp_shard_optimizer off")
            state["optimizer_state_dict"] = self.optimizer.state_dict()
        Path(file_path).parent.mkdir(parents=True, exist_ok=True)
        torch.save(state, str(file_path))
        self.logger.info(f"Model saved to '{file_path}'")


_KEEP = object()


def load_model(file_path, nn_module=_KEEP, optimizer=_KEEP, loss=_KEEP, device=_KEEP, model_name=_KEEP,
               change_params_func: Callable = lambda p: p, change_state_dict_func: Callable = lambda s, p: s,
               **kwargs) -> Model:
    """argus.load_model: rebuild the ``Model`` subclass named in the file from its ``params`` and load
    ``nn_state_dict`` strictly.  ``optimizer=None`` / ``loss=None`` skip those components (predictors.py:24);
    any other keyword replaces the entry of ``params``."""
    path = Path(file_path)
    if not path.exists():
        raise FileNotFoundError(f"No such file: '{file_path}'")
    state = torch.load(str(path), map_location="cpu", weights_only=False)
    name = state["model_name"] if model_name is _KEEP else model_name
    if name not in _MODEL_REGISTRY:
        raise ImportError(f"Model '{name}' not found in registered models: {sorted(_MODEL_REGISTRY)}")
    params = dict(state["params"])
    for key, value in (("nn_module", nn_module), ("optimizer", optimizer), ("loss", loss), ("device", device)):
        if value is not _KEEP:
            params[key] = value
    params.update(kwargs)
    params = change_params_func(params)
    model = _MODEL_REGISTRY[name](params)
    sd = change_state_dict_func(state["nn_state_dict"], params)
    model.get_nn_module().load_state_dict(sd, strict=True)
    if "optimizer_state_dict" in state and optimizer is _KEEP and params.get("optimizer") is not None:
        if getattr(model, "optimizer", None) is not None:
            model.optimizer.load_state_dict(state["optimizer_state_dict"])
        else:
            # models that build their optimizer lazily (MouseModel) restore the Adam moments / step counts when they do
            model._pending_optimizer_state = state["optimizer_state_dict"]
    model.eval()
    return model

