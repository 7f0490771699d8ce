"""Callbacks the reference attaches in scripts/train.py:115-139 — ``LoggingToFile``, ``LoggingToCSV``, ``Checkpoint``,
``LambdaLR``, ``CosineAnnealingLR`` — restated from pytorch-argus 1.0.0's published behaviour (the package is not
installable offline; see sensorium_amd/engine.py).  The learning-rate callbacks wrap the genuine
``torch.optim.lr_scheduler`` classes, so the schedule arithmetic (chained cosine recurrence, ``initial_lr`` carried
from the warm-up stage into the cosine stage) is torch's own, exactly as in the reference.
"""
from __future__ import annotations

import csv
import logging
import math
import os
import time
from pathlib import Path
from typing import Callable, List, Optional

import torch

from .engine import Callback, State

__all__ = ["LoggingToFile", "LoggingToCSV", "Checkpoint", "LRScheduler", "LambdaLR", "CosineAnnealingLR"]


class LoggingToFile(Callback):
    def __init__(self, file_path, create_dir: bool = True, formatter: str = "[%(asctime)s][%(levelname)s]: %(message)s",
                 append: bool = False):
        self.file_path = Path(file_path)
        self.create_dir = create_dir
        self.formatter = logging.Formatter(formatter)
        self.append = append
        self._handler: Optional[logging.Handler] = None

    def start(self, state: State):
        if self.create_dir:
            self.file_path.parent.mkdir(parents=True, exist_ok=True)
        if not self.append and self.file_path.exists():
            self.file_path.unlink()
        self._handler = logging.FileHandler(str(self.file_path))
        self._handler.setFormatter(self.formatter)
        state.logger.addHandler(self._handler)
        if state.logger.level == logging.NOTSET or state.logger.level > logging.INFO:
            state.logger.setLevel(logging.INFO)

    def _close(self, state: State):
        if self._handler is not None:
            state.logger.removeHandler(self._handler)
            self._handler.close()
            self._handler = None

    def complete(self, state: State):
        self._close(state)

    def catch_exception(self, state: State):
        state.logger.exception(state.exception)
        self._close(state)


class LoggingToCSV(Callback):
    """One row per epoch: time, epoch, lr, then every metric of the train state (val_* merged in)."""

    def __init__(self, file_path, create_dir: bool = True, separator: str = ",", write_header: bool = True,
                 append: bool = False):
        self.file_path = Path(file_path)
        self.create_dir = create_dir
        self.separator = separator
        self.write_header = write_header
        self.append = append
        self._file = None
        self._writer = None

    def start(self, state: State):
        if self.create_dir:
            self.file_path.parent.mkdir(parents=True, exist_ok=True)
        existed = self.file_path.exists() and self.file_path.stat().st_size > 0
        self._file = open(self.file_path, "a" if self.append else "w", newline="")
        self._writer = None
        self._skip_header = self.append and existed

    def epoch_complete(self, state: State):
        lr = state.model.get_lr()
        row = {"time": time.strftime("%Y-%m-%d %H:%M:%S"), "epoch": state.epoch,
               "lr": lr if not isinstance(lr, (list, tuple)) else lr[0]}
        row.update(state.metrics)
        if self._writer is None:
            self._writer = csv.DictWriter(self._file, fieldnames=list(row), delimiter=self.separator,
                                          extrasaction="ignore")
            if self.write_header and not self._skip_header:
                self._writer.writeheader()
        self._writer.writerow(row)
        self._file.flush()

    def _close(self, state: State):
        if self._file is not None:
            self._file.close()
            self._file = None

    complete = _close
    catch_exception = _close


class Checkpoint(Callback):
    """Save every ``period`` epochs under ``file_format.format(epoch=..., **state.metrics)``; keep the newest
    ``max_saves`` files (train.py:129-131 uses ``max_saves=1``).  ``save_model`` is the override point
    (src/ema.py:60-72 saves the EMA weights instead)."""

    def __init__(self, dir_path="", file_format: str = "model-{epoch:03d}-{train_loss:.6f}.pth",
                 max_saves: Optional[int] = None, period: int = 1, save_after_exception: bool = False,
                 optimizer_state: bool = False):
        if max_saves is not None and max_saves <= 0:
            raise ValueError("max_saves must be positive or None")
        self.dir_path = Path(dir_path)
        self.file_format = file_format
        self.max_saves = max_saves
        self.period = period
        self.save_after_exception = save_after_exception
        self.optimizer_state = optimizer_state
        self.saved_files_paths: List[Path] = []
        self.epochs_since_last_save = 0

    def _format_file_path(self, state: State) -> Path:
        return self.dir_path / self.file_format.format(epoch=state.epoch, **state.metrics)

    def save_model(self, state: State, file_path):
        state.model.save(file_path, optimizer_state=self.optimizer_state)

    def _is_writer(self) -> bool:
        import torch.distributed as dist
        return not (dist.is_available() and dist.is_initialized()) or dist.get_rank() == 0

    @staticmethod
    def _sync_all_ranks(state: State):
        """Sharded optimizer: the parameter / EMA slices other ranks own arrive by all-gather — a collective, so it runs here,
        on EVERY rank, before the writer-only part (a rank-0-only gather would hang or pair up with the other ranks' next
        reduce-scatter)."""
        sync = getattr(state.model, "sync_for_read", None)
        if sync is not None:
            sync()

    def save_checkpoint(self, state: State):
        file_path = self._format_file_path(state)
        self._sync_all_ranks(state)
        if not self._is_writer():                 # data parallel: parameters are identical, rank 0 writes
            return
        self.dir_path.mkdir(parents=True, exist_ok=True)
        self.save_model(state, file_path)
        self.saved_files_paths.append(file_path)
        if self.max_saves is not None:
            while len(self.saved_files_paths) > self.max_saves:
                old = self.saved_files_paths.pop(0)
                try:
                    os.remove(old)
                    state.logger.info(f"Model removed '{old}'")
                except OSError as err:
                    state.logger.warning(f"Fail to remove '{old}': {err}")

    def start(self, state: State):
        self.epochs_since_last_save = 0
        # refused HERE, on every rank and before training: Model.save would raise on the writer rank only, from inside
        # epoch_complete, and the other ranks would block in their next collective until the process group times out
        b = getattr(state.model, "buckets", None)
        shard = bool(getattr(state.model, "params", {}).get("ddp_shard_optimizer", False)) or (b is not None and b.shard)
        if self.optimizer_state and shard:
            raise RuntimeError("Checkpoint(optimizer_state=True) cannot be combined with the sharded optimizer: a rank holds only "
                               "1/N of the readout moments")

    def epoch_complete(self, state: State):
        self.epochs_since_last_save += 1
        if self.epochs_since_last_save >= self.period:
            self.save_checkpoint(state)
            self.epochs_since_last_save = 0

    def catch_exception(self, state: State):
        if self.save_after_exception and self._is_writer():
            needs = getattr(state.model, "needs_sync", None)
            if needs is not None and needs():
                # an exception is not known to have reached every rank: no collective from here
                state.logger.warning("Checkpoint: model not saved after the exception — the sharded optimizer's slices are "
                                     "spread over the ranks and gathering them needs every rank")
                return
            exc = type(state.exception).__name__
            self.save_model(state, self.dir_path / f"model-{state.epoch:03d}-{exc}.pth")


class LRScheduler(Callback):
    def __init__(self, scheduler_factory: Callable, step_on_iteration: bool = False):
        self.scheduler_factory = scheduler_factory
        self.step_on_iteration = step_on_iteration
        self._scheduler = None

    def start(self, state: State):
        if self._scheduler is None:
            self._scheduler = self.scheduler_factory(state.model.get_optimizer())

    def epoch_complete(self, state: State):
        if not self.step_on_iteration:
            self._scheduler.step()

    def iteration_complete(self, state: State):
        if self.step_on_iteration:
            self._scheduler.step()


class LambdaLR(LRScheduler):
    def __init__(self, lr_lambda, step_on_iteration: bool = False):
        super().__init__(lambda opt: torch.optim.lr_scheduler.LambdaLR(opt, lr_lambda), step_on_iteration)


class CosineAnnealingLR(LRScheduler):
    def __init__(self, T_max: int, eta_min: float = 0.0, step_on_iteration: bool = False):
        super().__init__(lambda opt: torch.optim.lr_scheduler.CosineAnnealingLR(opt, T_max=T_max, eta_min=eta_min),
                         step_on_iteration)


def cosine_lr_closed_form(base_lr: float, eta_min: float, t: int, t_max: int) -> float:
    """Closed form the chained torch recurrence follows for t <= T_max (used by the tests as the expected value)."""
    return eta_min + (base_lr - eta_min) * (1 + math.cos(math.pi * t / t_max)) / 2
