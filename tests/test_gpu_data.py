"""On-device batch assembly (dwn_assemble_inputs / dwn_assemble_targets through sensorium_amd/data_gpu.py): bit-exact
against the reference's StackInputsProcessor + CutMix outputs (tests/golden/data_pipeline.npz) and against the numpy
oracle on multi-mouse batches; then the whole training runtime on top of it (fit loop, schedules, EMA checkpoint,
load_model, predictor)."""
import ctypes as C
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import data_oracle as dorc  # noqa: E402
from tests.gpu_helpers import dev  # noqa: E402


def _poison_allocator():
    """Leave NaN bit patterns in the caching allocator so an element the kernels fail to write cannot pass as zero."""
    junk = torch.full((64 << 20,), float("nan"), device=dev())
    del junk


@pytest.mark.parametrize("compact", [True, False])
def test_assembly_matches_reference_golden(golden_dir, compact):
    from sensorium_amd.data_gpu import BatchAssembler, ClipPick, DeviceTrialStore
    gold = np.load(golden_dir / "data_pipeline.npz")
    checked = 0
    for c in range(int(gold["num_cases"])):
        h0, w0, sw, sh, e0, e1, size, step = (int(v) for v in gold[f"c{c}_meta"])
        store = DeviceTrialStore(dev())
        for i in range(2):
            store.add_trial(0, gold[f"c{c}_video{i}"], gold[f"c{c}_beh{i}"], gold[f"c{c}_pup{i}"],
                            gold[f"c{c}_resp{i}"], compact=compact)
        want_u8 = compact or gold[f"c{c}_video0"].dtype == np.uint8
        assert (store.trials[0][0].video.dtype == torch.uint8) == want_u8
        asm = BatchAssembler(store, (gold[f"c{c}_resp0"].shape[0],), dict(size=size, step=step, position="last"),
                             (sw, sh), float(gold[f"c{c}_fill"]))
        picks, want = [ClipPick(0, 0, e0)], [(gold[f"c{c}_x0"], gold[f"c{c}_t0"])]
        for seed in range(6):
            if bool(gold[f"c{c}_s{seed}_used"]):
                box = tuple(int(v) for v in gold[f"c{c}_s{seed}_box"])
                picks.append(ClipPick(0, 0, e0, (1, e1), box))
                want.append((gold[f"c{c}_s{seed}_x"], gold[f"c{c}_s{seed}_t"]))
        _poison_allocator()
        x, (targets, weights) = asm.assemble(picks)
        torch.cuda.synchronize()
        x, t = x.cpu().numpy(), targets[0].cpu().numpy()
        for b, (wx, wt) in enumerate(want):
            assert np.array_equal(x[b], wx), (c, b)
            assert np.array_equal(t[b], wt), (c, b)
            checked += 1
        assert torch.equal(weights.cpu(), torch.ones(len(picks), 1))
    assert checked >= 12


@pytest.mark.parametrize("compact", [True, False])
def test_mixup_and_random_choice_match_reference_golden(golden_dir, compact):
    """Mixup (whole-tensor blend of inputs and targets) and RandomChoiceMixer([CutMix, Mixup]) — src/mixers.py:22-33,70-79 —
    bit-exact against tests/golden/data_mixup.npz (reference outputs)."""
    from sensorium_amd.data_gpu import BatchAssembler, ClipPick, DeviceTrialStore
    gold = np.load(golden_dir / "data_mixup.npz")
    checked = 0
    for c in range(int(gold["num_cases"])):
        h0, w0, sw, sh, e0, e1, size, step = (int(v) for v in gold[f"c{c}_meta"])
        store = DeviceTrialStore(dev())
        for i in range(2):
            store.add_trial(0, gold[f"c{c}_video{i}"], gold[f"c{c}_beh{i}"], gold[f"c{c}_pup{i}"],
                            gold[f"c{c}_resp{i}"], compact=compact)
        asm = BatchAssembler(store, (gold[f"c{c}_resp0"].shape[0],), dict(size=size, step=step, position="last"),
                             (sw, sh), float(gold[f"c{c}_fill"]))
        picks, want = [], []
        for seed in range(5):
            key = f"c{c}_s{seed}"
            if bool(gold[key + "_mixup_used"]):
                picks.append(ClipPick(0, 0, e0, (1, e1), None, float(gold[key + "_mixup_lam"])))
                want.append((gold[key + "_mixup_x"], gold[key + "_mixup_t"]))
            if int(gold[key + "_choice"]) == 0:
                picks.append(ClipPick(0, 0, e0, (1, e1), tuple(int(v) for v in gold[key + "_choice_box"])))
            else:
                picks.append(ClipPick(0, 0, e0, (1, e1), None, float(gold[key + "_choice_lam"])))
            want.append((gold[key + "_choice_x"], gold[key + "_choice_t"]))
        _poison_allocator()
        x, (targets, weights) = asm.assemble(picks)
        torch.cuda.synchronize()
        x, t = x.cpu().numpy(), targets[0].cpu().numpy()
        for b, (wx, wt) in enumerate(want):
            assert np.array_equal(x[b], wx), (c, b)
            assert np.array_equal(t[b], wt), (c, b)
            checked += 1
    assert checked >= 14


def test_mixup_batch_drawn_on_the_host_matches_oracle():
    from sensorium_amd.data_gpu import BatchAssembler
    rng = np.random.default_rng(9)
    n_neurons = (11, 29)
    store, host = _synthetic_store(rng, n_neurons, 9, 13, u8=False)
    mixer = ("random_choice", {"mixers": [("cutmix", {"alpha": 1.0}), ("mixup", {"alpha": 0.4})],
                               "choice_probs": [0.4, 0.6], "prob": 0.8})
    asm = BatchAssembler(store, n_neurons, dict(size=8, step=2, position="last"), (16, 12), 2.5, mixer=mixer)
    picks = asm.draw_train_picks(np.random.RandomState(11), [0, 1, 1, 0, 1, 0, 0, 1, 1, 0, 1, 0])
    assert sum(p.box is not None for p in picks) >= 1 and sum(p.lam is not None for p in picks) >= 2
    assert any(p.mix is None for p in picks)
    _poison_allocator()
    x, (targets, weights) = asm.assemble(picks)
    torch.cuda.synchronize()
    wx, (wt, ww) = dorc.assemble_batch(host, [(p.mouse, p.trial, p.end_frame, p.mix) for p in picks], n_neurons, (16, 12),
                                       2.5, (8, 2), [p.box for p in picks], [p.lam for p in picks])
    assert np.array_equal(x.cpu().numpy(), wx)
    for m in range(2):
        assert np.array_equal(targets[m].cpu().numpy(), wt[m]), m
    assert np.array_equal(weights.cpu().numpy(), ww)


def _synthetic_store(rng, n_neurons, h0, w0, trials_per_mouse=3, length=70, u8=True):
    from sensorium_amd.data_gpu import DeviceTrialStore
    store = DeviceTrialStore(dev())
    host = {}
    for m, n in enumerate(n_neurons):
        host[m] = []
        for _ in range(trials_per_mouse):
            total = length + 5                                     # NaN tail past `length`, like the real files
            video = rng.integers(0, 256, size=(h0, w0, total)).astype(np.uint8 if u8 else np.float64)
            if not u8:
                video[..., length:] = np.nan
            beh = (rng.normal(size=(2, total)) * 10 + 20).astype(np.float32)
            pup = (rng.normal(size=(2, total)) * 20 + 90).astype(np.float32)
            resp = (rng.normal(size=(n, total)) * 5).astype(np.float32)
            beh[:, length:] = np.nan
            host[m].append(dict(video=video, behavior=beh, pupil_center=pup, responses=resp))
            store.add_trial(m, video, beh, pup, resp, length=length)
    return store, host


@pytest.mark.parametrize("h0,w0,size,u8", [(36, 64, (64, 36), True), (36, 64, (64, 64), False), (9, 13, (16, 12), True)])
def test_multi_mouse_batch_matches_oracle(h0, w0, size, u8):
    from sensorium_amd.data_gpu import BatchAssembler
    rng = np.random.default_rng(5)
    n_neurons = (11, 29, 16)
    store, host = _synthetic_store(rng, n_neurons, h0, w0, u8=u8)
    fs = dict(size=8, step=2, position="last")
    asm = BatchAssembler(store, n_neurons, fs, size, 0.0, cutmix=dict(alpha=1.0, prob=0.5))
    rs = np.random.RandomState(3)
    mice = [0, 1, 2, 2, 1, 0, 1, 2, 0, 0, 1]
    picks = asm.draw_train_picks(rs, mice)
    assert 2 <= sum(p.box is not None for p in picks) <= 9
    for p in picks:
        assert asm.gen.behind <= p.end_frame < 70
    _poison_allocator()
    x, (targets, weights) = asm.assemble(picks)
    torch.cuda.synchronize()
    wx, (wt, ww) = dorc.assemble_batch(host, [(p.mouse, p.trial, p.end_frame, p.mix) for p in picks], n_neurons, size,
                                       0.0, (8, 2), [p.box for p in picks])
    assert np.array_equal(x.cpu().numpy(), wx)
    for m in range(3):
        assert np.array_equal(targets[m].cpu().numpy(), wt[m]), m
    assert np.array_equal(weights.cpu().numpy(), ww)
    # inputs only (prediction path): same x, no targets touched
    x2 = asm.assemble(picks, with_targets=False)
    assert torch.equal(x2, x)


def test_assembly_argument_errors():
    import sensorium_amd._lib as L
    from sensorium_amd.data_gpu import BatchAssembler, ClipPick
    rng = np.random.default_rng(1)
    store, _ = _synthetic_store(rng, (5,), 6, 8, trials_per_mouse=1, length=30)
    asm = BatchAssembler(store, (5,), dict(size=8, step=2, position="last"), (8, 6))
    with pytest.raises(IndexError):
        asm.assemble([ClipPick(0, 0, 13)])             # window would start at frame -1
    with pytest.raises(IndexError):
        asm.assemble([ClipPick(0, 0, 35)])             # past the arrays
    with pytest.raises(ValueError):
        BatchAssembler(store, (6,), dict(size=8, step=2, position="last"), (8, 6)).assemble([ClipPick(0, 0, 20)])
    with pytest.raises(L.DwnError):
        BatchAssembler(store, (5,), dict(size=8, step=2, position="last"), (4, 6)).assemble([ClipPick(0, 0, 20)])
    assert L.lib.dwn_assemble_inputs(None, 1, 8, 6, 8, 6, 8, 0.0, None, 0, None) < 0
    assert b"null" in L.lib.dwn_last_error()
    assert L.lib.dwn_assemble_targets(None, 1, 8, None, None, 1, 5, None, 0, None) < 0


TINY = dict(readout_outputs=(7, 10), in_channels=5, core_features=(8, 8, 16), spatial_strides=(2, 1, 2),
            spatial_kernel=3, temporal_kernel=5, expansion_ratio=3, se_reduce_ratio=4, cortex_features=(32, 64),
            groups=2, softplus_beta=0.07, drop_rate=0.1, drop_path_rate=0.05)


def test_fit_on_device_loader_checkpoint_and_reload(tmp_path):
    """scripts/train.py:41-146 end to end on synthetic trials: warm-up stage + train stage, EMA checkpoint named by
    val_corr, best-model lookup, load_model, sliding-window predictor on the reloaded weights."""
    from sensorium_amd.argus_models import MouseModel
    from sensorium_amd.callbacks import CosineAnnealingLR, LambdaLR, LoggingToCSV, LoggingToFile
    from sensorium_amd.data_gpu import BatchAssembler, DeviceBatchLoader, DeviceValLoader
    from sensorium_amd.ema import EmaCheckpoint, ModelEma
    from sensorium_amd.engine import load_model
    from sensorium_amd.metrics import CorrelationMetric
    from sensorium_amd.predictors import Predictor
    from sensorium_amd.utils import get_best_model_path, get_lr, init_weights
    rng = np.random.default_rng(9)
    store, host = _synthetic_store(rng, TINY["readout_outputs"], 12, 16, trials_per_mouse=2, length=60)
    frame_stack = dict(size=8, step=2, position="last")
    params = {"nn_module": ("dwiseneuro", dict(TINY)), "loss": ("mice_poisson", {"log_input": False, "full": False,
                                                                                   "eps": 1e-8}),
              "optimizer": ("AdamW", {"lr": get_lr(3e-4, 8), "weight_decay": 0.05}), "device": "cuda:0",
              "frame_stack": frame_stack, "inputs_processor": ("stack_inputs", {"size": (16, 16), "pad_fill_value": 0.0}),
              "responses_processor": ("identity", {}), "amp": False, "iter_size": 1}
    torch.manual_seed(0)
    model = MouseModel(params)
    init_weights(model.nn_module)
    model.model_ema = ModelEma(model.nn_module, decay=0.9)
    asm = BatchAssembler(store, TINY["readout_outputs"], frame_stack, (16, 16), 0.0, cutmix=dict(alpha=1.0, prob=0.5))
    train_loader = DeviceBatchLoader(asm, batch_size=8, epoch_size=32, seed=1)
    val_loader = DeviceValLoader(BatchAssembler(store, TINY["readout_outputs"], frame_stack, (16, 16), 0.0), batch_size=8)
    assert len(train_loader) == 4 and len(val_loader) == 2          # 2 mice x 2 trials x (60 // 15) windows = 16 samples
    n_warm = len(train_loader) * 1
    model.fit(train_loader, val_loader=val_loader, num_epochs=1, metrics=[CorrelationMetric()],
              callbacks=[LoggingToFile(tmp_path / "log.txt", append=True), LoggingToCSV(tmp_path / "log.csv", append=True),
                         LambdaLR(lambda x: x / n_warm, step_on_iteration=True)])
    assert model.get_lr() == pytest.approx(get_lr(3e-4, 8))
    n_train = len(train_loader) * 2
    st = model.fit(train_loader, val_loader=val_loader, num_epochs=2, metrics=[CorrelationMetric()],
                   callbacks=[LoggingToFile(tmp_path / "log.txt", append=True),
                              LoggingToCSV(tmp_path / "log.csv", append=True),
                              EmaCheckpoint(tmp_path, file_format="model-{epoch:03d}-{val_corr:.6f}.pth", max_saves=1),
                              CosineAnnealingLR(T_max=n_train, eta_min=get_lr(3e-6, 8), step_on_iteration=True)])
    assert model.get_lr() == pytest.approx(get_lr(3e-6, 8))
    assert {"train_loss", "val_loss", "val_corr", "val_corr_mouse_0", "val_corr_mouse_1"} <= set(st.metrics)
    assert np.isfinite(st.metrics["train_loss"]) and -1.0 <= st.metrics["val_corr"] <= 1.0
    saved = [f for f in os.listdir(tmp_path) if f.endswith(".pth")]
    assert len(saved) == 1 and saved[0].startswith("model-002-")
    path, score = get_best_model_path(tmp_path, return_score=True)
    assert score == pytest.approx(abs(st.metrics["val_corr"]), abs=1e-6) or st.metrics["val_corr"] < 0
    # reload: the file holds the EMA weights; predictions of the reloaded model equal the live EMA module's
    loaded = load_model(path, device="cuda:0", optimizer=None, loss=None)
    for (k, a), b in zip(model.model_ema.ema.state_dict().items(), loaded.nn_module.state_dict().values()):
        assert torch.equal(a.cpu(), b.cpu()), k
    clip = asm.assemble([asm.val_picks(1)[0]], with_targets=False)
    live = model.predict(clip, 1)
    again = loaded.predict(clip, 1)
    assert torch.allclose(live, again, rtol=1e-5, atol=1e-6)
    # sliding-window trial prediction from the reloaded checkpoint against the oracle's accumulate/divide on the same model
    d = host[1][0]
    inputs = torch.from_numpy(dorc.stack_inputs(d["video"][..., :40], d["behavior"][..., :40],
                                                d["pupil_center"][..., :40], (16, 16), 0.0))
    pred = Predictor(loaded, frame_stack_size=8, frame_stack_step=2, windows_per_batch=4).predict_trial(inputs, 1)
    assert pred.shape == (10, 40) and np.isfinite(pred).all()
    one = Predictor(loaded, frame_stack_size=8, frame_stack_step=2, windows_per_batch=1).predict_trial(inputs, 1)
    assert np.allclose(pred, one, rtol=1e-4, atol=1e-5)
