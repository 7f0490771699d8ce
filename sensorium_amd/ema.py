"""ModelEma drop-in (reference: src/ema.py:12-58) on the fused multi-tensor lerp kernel."""
from __future__ import annotations

from copy import deepcopy

import torch
from torch import nn

from .callbacks import Checkpoint
from .optim import ema_lerp_state


class ModelEma(nn.Module):
    """Keeps ``ema = decay*ema + (1-decay)*model`` over *everything* in the state_dict (parameters, BN running
    statistics, ``num_batches_tracked`` with the reference's float-then-truncate behaviour, ``inv_freq``), in
    state_dict order like the reference (ema.py:49), but as one kernel launch instead of 365 x 3 small ops."""

    def __init__(self, model: nn.Module, decay: float = 0.9999, device=None):
        super().__init__()
        self.ema = deepcopy(model)
        self.ema.eval()
        self.decay = decay
        self.device = device
        if device is not None:
            self.ema.to(device=device)

    @torch.no_grad()
    def update(self, model: nn.Module, skip_parameters: bool = False):
        """``skip_parameters=True`` when FusedAdamWEma already folded the parameter EMA into the optimizer pass."""
        pset = {p.data_ptr() for p in model.parameters()} if skip_parameters else set()
        e_list, m_list = [], []
        for e, m in zip(self.ema.state_dict().values(), model.state_dict().values()):
            if m.data_ptr() in pset:
                continue
            e_list.append(e)
            m_list.append(m)
        ema_lerp_state(e_list, m_list, self.decay)

    @torch.no_grad()
    def set(self, model: nn.Module):
        for e, m in zip(self.ema.state_dict().values(), model.state_dict().values()):
            e.copy_(m)


class EmaCheckpoint(Checkpoint):
    """Checkpoint that stores the EMA weights (reference: src/ema.py:60-72), in argus' file format
    ``{'model_name', 'params', 'nn_state_dict'}`` so ``load_model`` / the predictor read it back unchanged."""

    def save_model(self, state, file_path):
        state.model._require_synced("EmaCheckpoint.save_model")    # the gather ran on all ranks in Checkpoint.save_checkpoint
        nn_module = state.model.model_ema.ema
        torch.save({"model_name": type(state.model).__name__, "params": state.model.params,
                    "nn_state_dict": {k: v.detach().to("cpu") for k, v in nn_module.state_dict().items()}},
                   str(file_path))
        state.logger.info(f"Model saved to '{file_path}'")
