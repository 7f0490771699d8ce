"""GPU tests of the step-level mirror of src/argus_models.py (train_step / val_step / predict), the sliding-window
predictor of src/predictors.py, and distillation — against the CPU oracle and the golden fixtures."""
import math

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import dwiseneuro_oracle as orc  # noqa: E402
from tests.gpu_helpers import dev, rel  # noqa: E402

TINY_KW = dict(readout_outputs=(7, 10), in_channels=5, core_features=(8, 8, 16), spatial_strides=(2, 1, 2),
               spatial_kernel=3, temporal_kernel=5, expansion_ratio=3, se_reduce_ratio=4, cortex_features=(32, 64),
               groups=2, softplus_beta=0.07, drop_rate=0.0, drop_path_rate=0.0)


def tiny_params(lr=2.4e-3, amp=False):
    return {"nn_module": ("dwiseneuro", dict(TINY_KW)), "loss": ("mice_poisson", {}),
            "optimizer": ("AdamW", {"lr": lr, "weight_decay": 0.05}), "device": "cuda:0", "amp": amp, "iter_size": 1}


def golden_sd(golden_dir, name):
    z = np.load(golden_dir / name)
    return z, {k[3:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("sd:")}


def test_train_step_matches_oracle_adamw_ema(golden_dir):
    """One full MouseModel.train_step (fp32): forward, Poisson loss, backward, fused AdamW, EMA — against the oracle
    forward/backward + the oracle's AdamW/EMA restatements (argus_models.py:43-71, ema.py:47-55)."""
    from sensorium_amd.argus_models import MouseModel
    z, sd = golden_sd(golden_dir, "tiny_model_train.npz")
    model = MouseModel(tiny_params())
    model.nn_module.load_state_dict(sd, strict=True)
    model.set_ema(0.999)
    x = torch.from_numpy(z["x"])
    targets = [torch.from_numpy(z[f"target_{m}"]) for m in range(2)]
    w = torch.from_numpy(z["mice_weights"])
    out = model.train_step([x, [targets, w]])
    torch.cuda.synchronize()
    assert set(out) == {"prediction", "target", "loss"} and isinstance(out["loss"], float)
    assert abs(out["loss"] - float(z["loss"])) <= 1e-3 * max(1.0, abs(float(z["loss"])))
    # expected parameters: reference gradients (golden) through the oracle's AdamW, then EMA
    new_sd = model.nn_module.state_dict()
    ema_sd = model.model_ema.ema.state_dict()
    worst = 0.0
    for k in z.files:
        if not k.startswith("grad:"):
            continue
        name = k[5:]
        p0 = sd[name]
        g = torch.from_numpy(z[k])
        p1, _, _ = orc.adamw_step(p0, g, torch.zeros_like(p0), torch.zeros_like(p0), 1, 2.4e-3, weight_decay=0.05)
        # Adam's first step moves every weight by ~lr*sign(g): compare the *update*, not the weight
        upd_ref, upd = (p1 - p0), (new_sd[name].cpu() - p0)
        gn = float(g.norm())
        if gn > 1e-3 * math.sqrt(g.numel()):          # analytically-zero grads give sign noise under Adam
            # elements whose gradient is not tiny must move the same way
            big = g.abs() > 1e-3 * g.abs().max()
            err = float((upd - upd_ref)[big].norm() / (upd_ref[big].norm() + 1e-12))
            worst = max(worst, err)
        e1 = orc.ema_update(p0, new_sd[name].cpu(), 0.999)
        assert rel(ema_sd[name], e1) < 1e-6, name
    assert worst < 2e-2, worst
    # BN buffers: model updated (momentum 0.1), EMA buffers lerped; num_batches_tracked int64 truncation -> 0
    assert int(new_sd["core.stem.1.bn.num_batches_tracked"]) == 1
    assert int(ema_sd["core.stem.1.bn.num_batches_tracked"]) == 0
    rm0, rm1 = sd["core.stem.1.bn.running_mean"], new_sd["core.stem.1.bn.running_mean"].cpu()
    assert rel(ema_sd["core.stem.1.bn.running_mean"], 0.999 * rm0 + 0.001 * rm1) < 1e-6


def test_val_step_and_predict(golden_dir):
    from sensorium_amd.argus_models import MouseModel
    z, sd = golden_sd(golden_dir, "tiny_model_eval.npz")
    model = MouseModel(tiny_params())
    model.nn_module.load_state_dict(sd, strict=True)
    x = torch.from_numpy(z["x"])
    targets = [torch.from_numpy(z[f"target_{m}"]) for m in range(2)]
    w = torch.from_numpy(z["mice_weights"])
    out = model.val_step([x, [targets, w]])
    for m in range(2):
        assert rel(out["prediction"][m], torch.from_numpy(z[f"pred_{m}"])) < 1e-3
    assert abs(out["loss"] - float(z["loss"])) <= 1e-3 * abs(float(z["loss"]))
    p1 = model.predict(x, 1)
    assert rel(p1, torch.from_numpy(z["pred_1"])) < 1e-3
    assert not model.nn_module.training


@pytest.mark.parametrize("windows_per_batch,use_graph", [(1, False), (4, False), (4, True)])
def test_predict_trial_matches_reference(golden_dir, windows_per_batch, use_graph):
    """Sliding-window blend (predictors.py:37-55) against the fixture produced by the reference loop."""
    from sensorium_amd.argus_models import MouseModel
    from sensorium_amd.predictors import Predictor
    z = np.load(golden_dir / "predict_trial.npz")
    _, sd = golden_sd(golden_dir, "tiny_model_eval.npz")
    model = MouseModel(tiny_params())
    model.nn_module.load_state_dict(sd, strict=True)
    pred = Predictor(model, frame_stack_size=int(z["size"]), frame_stack_step=int(z["step"]),
                     windows_per_batch=windows_per_batch, use_graph=use_graph)
    out = pred.predict_trial(torch.from_numpy(z["inputs"]), 1)
    assert out.shape == z["responses"].shape and out.dtype == np.float32
    assert rel(torch.from_numpy(out), torch.from_numpy(z["responses"])) < 1e-3
    # a second call is bit-identical (integer SE pooling sums; the blend is accumulated in a fixed order)
    again = pred.predict_trial(torch.from_numpy(z["inputs"]), mouse_index=1)      # the keyword spelling of the same call
    assert np.array_equal(again, out)
    with pytest.raises(TypeError):
        pred.predict_trial(torch.from_numpy(z["inputs"]))


def _fold_models(z, amp=False):
    from sensorium_amd.argus_models import MouseModel
    kw = {k: v for k, v in TINY_KW.items() if k not in ("drop_rate", "drop_path_rate", "softplus_beta", "spatial_strides")}
    models = []
    for seed in z["seeds"]:
        params = tiny_params(amp=amp)
        params["inputs_processor"] = ("stack_inputs", {"size": tuple(int(v) for v in z["frame"]), "pad_fill_value": 0})
        params["frame_stack"] = {"size": int(z["size"]), "step": int(z["step"]), "position": "last"}
        m = MouseModel(params)
        m.nn_module.load_state_dict(orc.make_state_dict(seed=int(seed), randomize_bn=True, **kw), strict=True)
        models.append(m)
    return models


@pytest.mark.parametrize("blend", ["ones", "linear"])
@pytest.mark.parametrize("windows_per_batch,use_graph", [(1, False), (5, False), (5, True)])
def test_fold_ensemble_matches_reference(golden_dir, blend, windows_per_batch, use_graph):
    """BASELINE.json configs[4] in miniature: three DIFFERENT fold models (fixture generated from the reference by
    oracle/make_golden_configs.py), a trial in the on-disk layout through the reference's call
    ``predict_trial(video, behavior, pupil_center, mouse_index)`` (src/predictors.py:36-41), sliding-window blend and the fold
    mean of scripts/predict.py:44-50 — all folds inside one forward / one captured graph per window batch."""
    from sensorium_amd.inputs import get_inputs_processor
    from sensorium_amd.predictors import EnsemblePredictor, Predictor, ensemble_predict_trial
    z = np.load(golden_dir / "ensemble_predict.npz")
    models = _fold_models(z)
    proc = get_inputs_processor("stack_inputs", {"size": tuple(int(v) for v in z["frame"]), "pad_fill_value": 0})
    assert np.array_equal(proc(z["video"], z["behavior"], z["pupil_center"]).numpy(), z["inputs"])      # bit-exact inputs
    kw = dict(frame_stack_size=int(z["size"]), frame_stack_step=int(z["step"]), windows_per_batch=windows_per_batch,
              use_graph=use_graph)
    ens = EnsemblePredictor(models, blend_weights=blend, **kw)
    ens.inputs_processor = proc
    out = ens.predict_trial(z["video"], z["behavior"], z["pupil_center"], 1)
    assert out.shape == z[f"ensemble_{blend}"].shape
    assert rel(torch.from_numpy(out), torch.from_numpy(z[f"ensemble_{blend}"])) < 1e-3
    # each fold on its own, and the reference's one-predictor-after-the-other mean
    singles = [Predictor(m, blend_weights=blend, **kw) for m in models]
    for k, p in enumerate(singles):
        one = p.predict_trial(torch.from_numpy(z["inputs"]), 1)
        assert rel(torch.from_numpy(one), torch.from_numpy(z[f"per_model_{blend}"][k])) < 1e-3
    seq = ensemble_predict_trial(singles, torch.from_numpy(z["inputs"]), 1)
    assert rel(torch.from_numpy(seq), torch.from_numpy(out)) < 1e-5
    # the folds really differ (a self-ensemble would pass every check above)
    assert rel(torch.from_numpy(z[f"per_model_{blend}"][0]), torch.from_numpy(z[f"per_model_{blend}"][1])) > 1e-2


def test_predictor_from_checkpoint_path(golden_dir, tmp_path):
    """The reference constructor ``Predictor(model_path, device, blend_weights)`` (src/predictors.py:22-34): frame stack and
    inputs processor come from the params stored in the checkpoint."""
    from sensorium_amd.predictors import Predictor
    z = np.load(golden_dir / "ensemble_predict.npz")
    model = _fold_models(z)[0]
    path = tmp_path / "fold_0.pth"
    model.save(path)
    pred = Predictor(str(path), device="cuda:0", blend_weights="ones")
    assert pred.model.loss is None and pred.frame_stack_size == int(z["size"]) and pred.frame_stack_step == int(z["step"])
    out = pred.predict_trial(z["video"], z["behavior"], z["pupil_center"], mouse_index=1)
    assert rel(torch.from_numpy(out), torch.from_numpy(z["per_model_ones"][0])) < 1e-3


def test_distillation_step_runs_and_uses_teacher(golden_dir):
    """configs/distillation_001.py semantics: a frozen teacher fills the zero-weight (sample, mouse) pairs."""
    from sensorium_amd.argus_models import MouseModel
    z, sd = golden_sd(golden_dir, "tiny_model_train.npz")
    student, teacher = MouseModel(tiny_params()), MouseModel(tiny_params())
    student.nn_module.load_state_dict(sd, strict=True)
    teacher.nn_module.load_state_dict(sd, strict=True)
    teacher.eval()
    student.distill_model = teacher.nn_module
    student.distill_ratio = 0.36
    x = torch.from_numpy(z["x"])
    targets = [torch.from_numpy(z[f"target_{m}"]).clone() for m in range(2)]
    w = torch.from_numpy(z["mice_weights"]).clone()
    out = student.train_step([x, [targets, w]])
    tw = out["target"][1].cpu()
    b = w.shape[0]
    expect_w = 0.36 / 0.64 * float(w.sum()) / float((w == 0).sum())
    assert torch.allclose(tw[w == 0], torch.full_like(tw[w == 0], expect_w))
    assert math.isfinite(out["loss"]) and abs(out["loss"] - float(z["loss"])) > 1e-6     # soft labels changed the loss


def test_eval_forward_is_hipgraph_capturable(golden_dir):
    """The C-ABI never allocates or synchronises, so a whole eval forward can be captured into a hipGraph
    (torch.cuda.CUDAGraph) and replayed on new inputs — the launch pattern of the sliding-window predictor."""
    from sensorium_amd import DwiseNeuro
    z, sd = golden_sd(golden_dir, "tiny_model_eval.npz")
    model = DwiseNeuro(**TINY_KW)
    model.load_state_dict(sd, strict=True)
    model = model.to(dev()).eval()
    x = torch.from_numpy(z["x"]).to(dev())
    static_in = x.clone()
    with torch.no_grad():
        for _ in range(2):                       # warm-up on a side stream (allocator, lazy kernel loading)
            s = torch.cuda.Stream()
            s.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(s):
                model(static_in, 1)
            torch.cuda.current_stream().wait_stream(s)
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            static_out = model(static_in, 1)
        ref = torch.from_numpy(z["pred_1"])
        graph.replay()
        torch.cuda.synchronize()
        assert rel(static_out, ref) < 1e-3
        # new input through the same captured graph
        x2 = torch.flip(x, dims=[0]).contiguous()
        static_in.copy_(x2)
        graph.replay()
        torch.cuda.synchronize()
        ref2 = torch.flip(ref, dims=[0])         # eval mode: samples are independent
        eager = model(x2, 1)
        assert rel(eager, ref2) < 1e-3, "eager forward on the flipped batch"
        assert rel(static_out, ref2) < 1e-3, "graph replay on the flipped batch"
        graph.replay()                            # replays must be idempotent (all workspaces re-zeroed inside the graph)
        torch.cuda.synchronize()
        assert rel(static_out, eager) < 1e-5


def test_pointer_table_cache_uploads_only_on_change():
    """optim._TableCache: the device copy of a pointer table is reused while its contents are unchanged."""
    import numpy as np
    from sensorium_amd.optim import _ENTRY_DTYPE, _TableCache
    cache = _TableCache()
    e = np.zeros(3, dtype=_ENTRY_DTYPE)
    e["numel"] = [1, 2, 3]
    a = cache.get(e, dev())
    b = cache.get(e.copy(), dev())
    assert b is a
    e["numel"][0] = 9
    c = cache.get(e, dev())
    assert c is not a
    torch.cuda.synchronize()
    assert bytes(c.cpu().numpy().tobytes()) == e.view(np.uint8).tobytes()


def _tiny_batch(golden_dir):
    z, sd = golden_sd(golden_dir, "tiny_model_train.npz")
    x = torch.from_numpy(z["x"])
    targets = [torch.from_numpy(z[f"target_{m}"]) for m in range(2)]
    return sd, [x, [targets, torch.from_numpy(z["mice_weights"])]]


def test_model_ema_assigned_after_optimizer_exists(golden_dir):
    """scripts/train.py:53 assigns ``model.model_ema = ModelEma(...)`` as a plain attribute, possibly after
    get_lr()/set_lr() have built the optimizer: the parameter EMA must still move (it rides in the AdamW kernel only when
    the optimizer is bound to that ModelEma), and re-assigning it later must keep the Adam moments and step counts."""
    from sensorium_amd.argus_models import MouseModel
    from sensorium_amd.ema import ModelEma
    sd, batch = _tiny_batch(golden_dir)
    model = MouseModel(tiny_params())
    model.nn_module.load_state_dict(sd, strict=True)
    assert model.get_lr() == pytest.approx(2.4e-3)                 # builds the optimizer: no EMA exists yet
    model.model_ema = ModelEma(model.nn_module, decay=0.9)
    before = {k: v.clone() for k, v in model.model_ema.ema.state_dict().items()}
    model.train_step(batch)
    torch.cuda.synchronize()
    name = "core.blocks.1.conv_pw.0.weight"
    new_p = model.nn_module.state_dict()[name]
    exp = orc.ema_update(before[name].cpu(), new_p.cpu(), 0.9)
    assert rel(model.model_ema.ema.state_dict()[name], exp) < 1e-6
    assert not torch.equal(model.model_ema.ema.state_dict()[name], before[name])
    # replace the EMA mid-training: optimizer state survives, the new copy is the one that gets updated
    p0 = next(iter(model.optimizer.state.values()))
    step0, m0 = int(p0["step"]), p0["exp_avg"].clone()
    model.model_ema = ModelEma(model.nn_module, decay=0.5)
    st = next(iter(model.optimizer.state.values()))
    assert int(st["step"]) == step0 and torch.equal(st["exp_avg"], m0)
    before2 = model.model_ema.ema.state_dict()[name].clone()
    model.train_step(batch)
    torch.cuda.synchronize()
    exp2 = orc.ema_update(before2.cpu(), model.nn_module.state_dict()[name].cpu(), 0.5)
    assert rel(model.model_ema.ema.state_dict()[name], exp2) < 1e-6
    assert int(next(iter(model.optimizer.state.values()))["step"]) == step0 + 1
    # detaching: plain AdamW keeps training
    model.model_ema = None
    model.train_step(batch)


def test_save_load_restores_optimizer_state(golden_dir, tmp_path):
    """Model.save(optimizer_state=True) -> load_model: MouseModel builds its optimizer lazily, so the Adam moments and step
    counts are restored when it does; the next step must equal the step an uninterrupted run takes."""
    from sensorium_amd.argus_models import MouseModel
    from sensorium_amd.engine import load_model
    sd, batch = _tiny_batch(golden_dir)
    model = MouseModel(tiny_params())
    model.nn_module.load_state_dict(sd, strict=True)
    model.train_step(batch)
    model.train_step(batch)
    path = tmp_path / "ckpt.pth"
    model.save(path, optimizer_state=True)
    loaded = load_model(str(path), device="cuda:0")
    assert loaded.optimizer is None
    # the moments and step counts come back bit for bit as soon as the optimizer is built
    opt = loaded.get_optimizer()
    saved = {n: model.optimizer.state[p] for n, p in model.nn_module.named_parameters()}
    for n, p in loaded.nn_module.named_parameters():
        st = opt.state[p]
        assert int(st["step"]) == 2 and int(saved[n]["step"]) == 2, n
        assert torch.equal(st["exp_avg"], saved[n]["exp_avg"]) and torch.equal(st["exp_avg_sq"], saved[n]["exp_avg_sq"]), n
    model.train_step(batch)
    loaded.train_step(batch)
    torch.cuda.synchronize()
    assert int(next(iter(loaded.optimizer.state.values()))["step"]) == 3
    # same third step up to the run-to-run noise of the atomically accumulated gradients (Adam turns noise on tiny gradients
    # into visible parameter differences, hence the loose bound; a fresh optimizer would take a first-step-sized jump instead)
    for (k, a), b in zip(model.nn_module.state_dict().items(), loaded.nn_module.state_dict().values()):
        if a.is_floating_point() and "running" not in k:
            assert rel(b, a) < 3e-2, k
