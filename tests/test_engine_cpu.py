"""Host-side training runtime (sensorium_amd/engine.py, callbacks.py, utils.py, CorrelationMetric): the argus surface
scripts/train.py:41-146 drives.  CPU only — a two-layer toy ``Model`` stands in for ``MouseModel`` so the loop,
callback order, schedules, checkpoints and ``load_model`` are exercised without a GPU.

PARITY UNPINNED for this file's subject: pytorch-argus 1.0.0 is neither vendored in the reference nor installable offline, so the
event order, epoch numbering and metric merging asserted below are argus' PUBLISHED behaviour restated (validation pass before
epoch 1, epochs from 1, ``val_*`` merged before the user's epoch_complete callbacks) — they pin this runtime against itself and
against how scripts/train.py uses the API, not against argus.  What IS pinned to the reference: ``init_weights`` / ``get_lr``
(bit-exact against src/utils.py, below), the checkpoint file format (src/ema.py:67-72) and ``corr`` (src/metrics.py:11-31)."""
import csv
import importlib.util
import math
import os

import numpy as np
import pytest
import torch
from torch import nn

from sensorium_amd import engine
from sensorium_amd.callbacks import (Checkpoint, CosineAnnealingLR, LambdaLR, LoggingToCSV, LoggingToFile,
                                     cosine_lr_closed_form)
from sensorium_amd.metrics import CorrelationMetric, corr
from sensorium_amd.utils import get_best_model_path, get_lr, init_weights


class ToyNet(nn.Module):
    def __init__(self, width: int = 4):
        super().__init__()
        self.fc = nn.Linear(3, width)
        self.out = nn.Linear(width, 2)

    def forward(self, x):
        return self.out(torch.tanh(self.fc(x)))


class ToyModel(engine.Model):
    nn_module = {"toy": ToyNet}
    loss = {"mse": nn.MSELoss}
    optimizer = {"SGD": torch.optim.SGD}

    def train_step(self, batch, state):
        self.train()
        self.optimizer.zero_grad()
        x, y = batch
        pred = self.nn_module(x)
        loss = self.loss(pred, y)
        loss.backward()
        self.optimizer.step()
        return {"prediction": pred.detach(), "target": y, "loss": loss.item()}

    def val_step(self, batch, state):
        self.eval()
        with torch.no_grad():
            x, y = batch
            pred = self.nn_module(x)
            return {"prediction": pred, "target": y, "loss": self.loss(pred, y).item()}


PARAMS = {"nn_module": ("toy", {"width": 5}), "loss": ("mse", {}), "optimizer": ("SGD", {"lr": 0.1}), "device": "cpu"}


def loader(n_batches, seed=0):
    g = torch.Generator().manual_seed(seed)
    return [(torch.randn(4, 3, generator=g), torch.randn(4, 2, generator=g)) for _ in range(n_batches)]


class Recorder(engine.Callback):
    def __init__(self, log, tag):
        self.log, self.tag = log, tag

    def start(self, state): self.log.append((self.tag, "start", state.epoch, state.iteration))
    def epoch_start(self, state): self.log.append((self.tag, "epoch_start", state.epoch, state.iteration))
    def iteration_complete(self, state): self.log.append((self.tag, "iter", state.epoch, state.iteration))
    def epoch_complete(self, state): self.log.append((self.tag, "epoch_complete", state.epoch, dict(state.metrics)))
    def complete(self, state): self.log.append((self.tag, "complete", state.epoch, state.iteration))


def test_fit_event_order_and_metric_merge():
    torch.manual_seed(0)
    model = ToyModel(dict(PARAMS))
    log = []
    st = model.fit(loader(3), val_loader=loader(2, 1), num_epochs=2, callbacks=[Recorder(log, "train")],
                   val_callbacks=[Recorder(log, "val")])
    kinds = [(t, k, e) for t, k, e, _ in log]
    # argus: one validation pass (epoch 0) before training starts
    assert kinds[:5] == [("val", "start", -1), ("val", "epoch_start", 0), ("val", "iter", 0), ("val", "iter", 0),
                         ("val", "epoch_complete", 0)]
    # epochs are numbered from 1 and iterations restart every epoch
    train_iters = [(e, i) for t, k, e, i in log if t == "train" and k == "iter"]
    assert train_iters == [(1, 1), (1, 2), (1, 3), (2, 1), (2, 2), (2, 3)]
    # the validation epoch of train epoch e runs *before* the user's epoch_complete callbacks and is merged in
    for e in (1, 2):
        i_val = kinds.index(("val", "epoch_complete", e))
        i_tr = kinds.index(("train", "epoch_complete", e))
        assert i_val < i_tr
        merged = log[i_tr][3]
        assert set(merged) == {"train_loss", "val_loss"}
        assert merged["val_loss"] == log[i_val][3]["val_loss"]
    assert st.epoch == 2 and set(st.metrics) == {"train_loss", "val_loss"}
    assert kinds[-1] == ("train", "complete", 2)


def test_loss_metric_is_mean_of_step_losses():
    torch.manual_seed(0)
    model = ToyModel(dict(PARAMS))
    losses = []

    class Tap(engine.Callback):
        def iteration_complete(self, state): losses.append(state.step_output["loss"])

    st = model.fit(loader(5), num_epochs=1, callbacks=[Tap()])
    assert st.metrics["train_loss"] == pytest.approx(sum(losses) / 5, rel=0, abs=1e-12)
    vals = model.validate(loader(2, 3))
    assert set(vals) == {"val_loss"}


def test_warmup_then_cosine_schedule_per_iteration():
    """train.py:121-136: stage 'warmup' = LambdaLR(x / n) then stage 'train' = CosineAnnealingLR(T_max, eta_min), both
    stepped per iteration on the same optimizer; the cosine stage must start from the *base* lr (``initial_lr``)."""
    torch.manual_seed(0)
    base = get_lr(3e-4, 32)
    assert base == pytest.approx(2.4e-3)
    params = dict(PARAMS)
    params["optimizer"] = ("SGD", {"lr": base})
    model = ToyModel(params)
    seen = []

    class LrTap(engine.Callback):
        def iteration_start(self, state): seen.append(state.model.get_lr())

    n_warm = 2 * 3
    model.fit(loader(3), num_epochs=2, callbacks=[LambdaLR(lambda x: x / n_warm, step_on_iteration=True), LrTap()])
    assert seen == pytest.approx([base * i / n_warm for i in range(n_warm)])
    assert model.get_lr() == pytest.approx(base)
    seen.clear()
    n_train, eta_min = 3 * 3, get_lr(3e-6, 32)
    model.fit(loader(3), num_epochs=3,
              callbacks=[CosineAnnealingLR(T_max=n_train, eta_min=eta_min, step_on_iteration=True), LrTap()])
    want = [cosine_lr_closed_form(base, eta_min, t, n_train) for t in range(n_train)]
    assert seen == pytest.approx(want, rel=1e-9)
    assert model.get_lr() == pytest.approx(eta_min)


def test_checkpoint_rotation_load_model_and_best_path(tmp_path):
    torch.manual_seed(0)
    model = ToyModel(dict(PARAMS))
    fmt = "model-{epoch:03d}-{val_loss:.6f}.pth"
    model.fit(loader(2), val_loader=loader(2, 1), num_epochs=3,
              callbacks=[Checkpoint(tmp_path, file_format=fmt, max_saves=2)])
    files = sorted(os.listdir(tmp_path))
    assert len(files) == 2 and files[0].startswith("model-002-") and files[1].startswith("model-003-")
    best = get_best_model_path(tmp_path, more_better=False)
    path, score = get_best_model_path(tmp_path, return_score=True, more_better=False)
    assert best == path and f"{score:.6f}" in path.name
    assert get_best_model_path(tmp_path / "missing") is None
    last = tmp_path / files[1]
    blob = torch.load(last, weights_only=False)
    assert set(blob) == {"model_name", "params", "nn_state_dict"} and blob["model_name"] == "ToyModel"
    loaded = engine.load_model(last, device="cpu")
    for k, v in model.nn_module.state_dict().items():
        assert torch.equal(loaded.nn_module.state_dict()[k], v), k
    assert not loaded.nn_module.training
    bare = engine.load_model(last, optimizer=None, loss=None)
    assert bare.optimizer is None and bare.loss is None
    with pytest.raises(RuntimeError):
        bare.fit(loader(1))
    with pytest.raises(FileNotFoundError):
        engine.load_model(tmp_path / "nope.pth")
    with pytest.raises(ImportError):
        engine.load_model(last, model_name="Unknown")


def test_logging_callbacks_write_epoch_rows(tmp_path):
    torch.manual_seed(0)
    model = ToyModel(dict(PARAMS))
    for _ in range(2):          # two stages appending to the same files, as train.py:115-118 does
        model.fit(loader(2), val_loader=loader(1, 1), num_epochs=2,
                  callbacks=[LoggingToFile(tmp_path / "log.txt", append=True),
                             LoggingToCSV(tmp_path / "log.csv", append=True)])
    rows = list(csv.DictReader(open(tmp_path / "log.csv")))
    assert [r["epoch"] for r in rows] == ["1", "2", "1", "2"]
    assert {"time", "epoch", "lr", "train_loss", "val_loss"} <= set(rows[0])
    text = (tmp_path / "log.txt").read_text()
    assert text.count("train - epoch: 1") == 2 and "val_loss" in text and "lr: 0.1" in text


def test_exception_reaches_callbacks_and_propagates(tmp_path):
    model = ToyModel(dict(PARAMS))
    seen = []

    class Boom(engine.Callback):
        def iteration_complete(self, state):
            if state.iteration == 2:
                raise KeyboardInterrupt()

        def catch_exception(self, state): seen.append(type(state.exception).__name__)

    with pytest.raises(KeyboardInterrupt):
        model.fit(loader(3), num_epochs=1,
                  callbacks=[Boom(), Checkpoint(tmp_path, save_after_exception=True)])
    assert seen == ["KeyboardInterrupt"]
    assert os.listdir(tmp_path) == ["model-001-KeyboardInterrupt.pth"]


def test_correlation_metric_matches_direct_corr():
    """src/metrics.py:34-82: only rows with non-zero mouse weight count; (B,N,T) -> (B*T, N); mean over neurons, then mice.
    The metric keeps running sums instead of the epoch's tensors; the expected value below is the reference's procedure."""
    rng = np.random.default_rng(3)
    metric = CorrelationMetric()
    metric.reset()
    keep = {0: ([], []), 1: ([], [])}
    for _ in range(3):
        w = torch.tensor([[1.0, 0.0], [0.0, 1.0], [1.0, 0.0], [0.0, 1.0]])
        preds = [torch.from_numpy(rng.random((4, n, 6)).astype(np.float32)) for n in (5, 7)]
        targs = [torch.from_numpy(rng.random((4, n, 6)).astype(np.float32)) for n in (5, 7)]
        metric.update({"prediction": preds, "target": (targs, w)})
        for m in (0, 1):
            rows = (w[:, m] != 0).numpy()
            keep[m][0].append(preds[m].numpy()[rows].transpose(0, 2, 1).reshape(-1, preds[m].shape[1]))
            keep[m][1].append(targs[m].numpy()[rows].transpose(0, 2, 1).reshape(-1, preds[m].shape[1]))
    st = engine.State(phase="val")
    metric.epoch_complete(st)
    want = [corr(np.concatenate(keep[m][0]), np.concatenate(keep[m][1]), axis=0).mean() for m in (0, 1)]
    # (the streaming float64 sums against the reference's float32 two-pass corr: rounding of the latter, ~1e-7)
    assert st.metrics["val_corr_mouse_0"] == pytest.approx(want[0], abs=1e-6)
    assert st.metrics["val_corr_mouse_1"] == pytest.approx(want[1], abs=1e-6)
    assert st.metrics["val_corr"] == pytest.approx(np.mean(want), abs=1e-6)
    # a mouse with no weighted rows in the epoch is left out of the mean
    metric.reset()
    w = torch.tensor([[1.0, 0.0]] * 4)
    metric.update({"prediction": preds, "target": (targs, w)})
    st = engine.State(phase="val")
    metric.epoch_complete(st)
    assert "val_corr_mouse_1" not in st.metrics and st.metrics["val_corr"] == st.metrics["val_corr_mouse_0"]


def _probe_net():
    return nn.Sequential(nn.Conv3d(4, 8, (1, 3, 3), groups=4, bias=True), nn.BatchNorm3d(8),
                         nn.Conv1d(8, 6, (1,), groups=2, bias=False), nn.BatchNorm1d(6), nn.Linear(6, 3))


def test_init_weights_rule():
    net = _probe_net()
    with torch.no_grad():
        net[1].weight.fill_(3.0); net[1].bias.fill_(3.0)
    torch.manual_seed(7)
    init_weights(net)
    assert torch.all(net[1].weight == 1) and torch.all(net[1].bias == 0) and torch.all(net[0].bias == 0)
    big = nn.Conv3d(64, 448, (1, 1, 1), bias=False)
    init_weights(big)
    assert float(big.weight.detach().std()) == pytest.approx(math.sqrt(2.0 / 448), rel=0.05)      # fan_out = 1*448/1
    dw = nn.Conv3d(448, 448, (1, 3, 3), groups=448, bias=False)
    init_weights(dw)
    assert float(dw.weight.detach().std()) == pytest.approx(math.sqrt(2.0 / 9), rel=0.05)         # fan_out = 9*448/448
    assert float(net[4].weight.detach().abs().max()) <= 1.0 / math.sqrt(3)


@pytest.mark.skipif(not os.path.exists("/root/reference/src/utils.py"), reason="reference checkout not present")
def test_init_weights_bit_exact_against_reference():
    spec = importlib.util.spec_from_file_location("ref_utils", "/root/reference/src/utils.py")
    ref = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ref)
    a, b = _probe_net(), _probe_net()
    torch.manual_seed(11)
    ref.init_weights(a)
    torch.manual_seed(11)
    init_weights(b)
    for (k, va), vb in zip(a.state_dict().items(), b.state_dict().values()):
        assert torch.equal(va, vb), k
    assert ref.get_lr(3e-4, 32) == get_lr(3e-4, 32)
