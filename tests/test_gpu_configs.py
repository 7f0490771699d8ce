"""BASELINE.json configs[2] (ten readouts) and configs[3] (configs/distillation_001.py) on the HIP path against digests
generated from the REAL reference by oracle/make_golden_configs.py (tests/golden/ten_mice_digest.npz, distill_digest.npz),
plus the bf16 (benchmark dtype) legs at full width and depth with their measured error bounds."""
import math

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import dwiseneuro_oracle as orc  # noqa: E402
from tests.gpu_helpers import NUM_NEURONS_ALL, dev, synth_inputs  # noqa: E402

TRACKED = ("core.stem.0.weight", "core.blocks.1.conv_pw.0.weight", "core.blocks.9.spat_covn_dw.0.weight",
           "core.blocks.17.conv_pwl.0.weight", "cortex.layers.0.conv.weight", "cortex.layers.2.conv.weight")


def _close(got, ref, tol, what):
    got, ref = np.asarray(got, dtype=np.float64), np.asarray(ref, dtype=np.float64)
    err = np.abs(got - ref) / (np.abs(ref) + 1e-12)
    assert float(err.max()) <= tol, f"{what}: rel err {float(err.max()):.3e} > {tol:g} (got {got}, ref {ref})"


@pytest.mark.parametrize("dtype,tol_fwd,tol_grad", [(torch.float32, 1e-3, 5e-3), (torch.bfloat16, 2e-2, 6e-2)])
def test_ten_readout_model_digest(golden_dir, dtype, tol_fwd, tol_grad):
    """configs[2]: all ten readouts (src/constants.py:38, dwiseneuro.py:402-403), expansion 7, full width and depth, B=2, T=8,
    36x64, train mode, dense mice weights — loss, per-mouse prediction digests and gradient norms from the reference.
    fp32: the north-star 1e-3 (activations) bound; bf16 storage: measured 5e-3 (fwd) / 2e-2 (grad norms), bounds 2e-2 / 6e-2."""
    from sensorium_amd import DwiseNeuro, MicePoissonLoss
    z = np.load(golden_dir / "ten_mice_digest.npz")
    sd = orc.make_state_dict(readout_outputs=NUM_NEURONS_ALL, expansion_ratio=7, seed=12)
    model = DwiseNeuro(readout_outputs=NUM_NEURONS_ALL, expansion_ratio=7, drop_rate=0.0, drop_path_rate=0.0)
    model.load_state_dict(sd, strict=True)
    model = model.to(dev()).train()
    rng = np.random.default_rng(20231123)
    x, targets, _ = synth_inputs(rng, 2, 8, 36, 64, NUM_NEURONS_ALL)
    weights = rng.random(size=(2, 10)).astype(np.float32) + 0.25
    assert np.array_equal(weights, z["weights"])
    with torch.autocast("cuda", dtype=torch.bfloat16, enabled=dtype == torch.bfloat16):
        preds = model(torch.from_numpy(x).to(dev()))
        loss = MicePoissonLoss()(preds, ([torch.from_numpy(t).to(dev()) for t in targets], torch.from_numpy(weights).to(dev())))
    loss.backward()
    torch.cuda.synchronize()
    assert len(preds) == 10 and all(p.shape == (2, n, 8) for p, n in zip(preds, NUM_NEURONS_ALL))
    _close(float(loss.detach()), z["loss"], tol_fwd, "loss")
    _close([float(p.detach().double().norm()) for p in preds], z["pred_l2"], tol_fwd, "per-mouse prediction L2")
    _close([float(p.detach().mean()) for p in preds], z["pred_mean"], tol_fwd, "per-mouse prediction mean")
    named = dict(model.named_parameters())
    tot = math.sqrt(sum(float(v.grad.double().norm()) ** 2 for v in named.values()))
    _close(tot, z["grad_total_norm"], tol_grad, "total gradient norm")
    _close([float(named[f"readouts.{m}.layer.1.weight"].grad.double().norm()) for m in range(10)], z["grad_readout_w"], tol_grad,
           "readout weight gradient norms")
    _close([float(named[f"readouts.{m}.layer.1.bias"].grad.double().norm()) for m in range(10)], z["grad_readout_b"], tol_grad,
           "readout bias gradient norms")
    _close([float(named[k].grad.double().norm()) for k in TRACKED], z["grad_tracked"], tol_grad, "trunk gradient norms")


def test_distillation_step_matches_reference(golden_dir):
    """configs[3] at the real shapes: frozen expansion-7 teacher -> MouseModel.add_distill_predictions (argus_models.py:31-41,
    ratio 0.36) -> expansion-6 student forward + MicePoissonLoss + backward, fp32.  The filled weights must equal the
    reference's exactly, the filled targets must be the teacher's predictions element-wise (sampled elements + per-(sample,
    mouse) sums / norms), and the student's loss / gradients must match the reference's digests."""
    from sensorium_amd import DwiseNeuro
    from sensorium_amd.argus_models import MouseModel
    z = np.load(golden_dir / "distill_digest.npz")
    b, t = 3, 4
    params = {"nn_module": ("dwiseneuro", dict(readout_outputs=NUM_NEURONS_ALL, expansion_ratio=6, drop_rate=0.0,
                                                drop_path_rate=0.0)),
              "loss": ("mice_poisson", {}), "optimizer": ("AdamW", {"lr": 1e-4}), "device": "cuda:0", "amp": False}
    model = MouseModel(params)
    model.nn_module.load_state_dict(orc.make_state_dict(readout_outputs=NUM_NEURONS_ALL, expansion_ratio=6, seed=22), strict=True)
    teacher = DwiseNeuro(readout_outputs=NUM_NEURONS_ALL, expansion_ratio=7, drop_rate=0.0, drop_path_rate=0.0)
    teacher.load_state_dict(orc.make_state_dict(readout_outputs=NUM_NEURONS_ALL, expansion_ratio=7, seed=21), strict=True)
    model.distill_model = teacher.to(dev()).eval()
    model.distill_ratio = float(z["ratio"])
    rng = np.random.default_rng(20231124)
    x, targets, weights = synth_inputs(rng, b, t, 36, 64, NUM_NEURONS_ALL)
    for m in range(10):
        targets[m] *= weights[:, m][:, None, None]
    xd = torch.from_numpy(x).to(dev())
    tt = [torch.from_numpy(tg).to(dev()) for tg in targets]
    wt = torch.from_numpy(weights).to(dev())
    model.add_distill_predictions(xd, (tt, wt))
    torch.cuda.synchronize()
    assert np.array_equal(wt.cpu().numpy(), z["filled_weights"]), "soft-label weights differ from the reference's"
    for m in range(10):
        tm = tt[m].cpu().numpy()
        idx = z["sample_idx"][m]
        got = tm[idx[:, 0], idx[:, 1], idx[:, 2]]
        # (the fp32 teacher runs in eval mode: its GEMMs use the bf16 hi/lo split products, ~5e-7 relative L2 on the
        # predictions — the exponentially small softplus outputs among the sampled values move by up to ~3e-4 relative)
        assert np.allclose(got, z["sample_val"][m], rtol=1e-3, atol=1e-5), f"filled targets of mouse {m}"
        _close([float(np.linalg.norm(tm[i].astype(np.float64))) for i in range(b)], z["target_l2"][:, m], 2e-4, f"target L2, mouse {m}")
        _close([float(tm[i].astype(np.float64).sum()) for i in range(b)], z["target_sum"][:, m], 2e-4, f"target sum, mouse {m}")
    model.train()
    preds = model.nn_module(xd)
    loss = model.loss(preds, (tt, wt))
    loss.backward()
    torch.cuda.synchronize()
    _close(float(loss.detach()), z["loss"], 1e-3, "student loss")
    _close([float(p.detach().double().norm()) for p in preds], z["pred_l2"], 1e-3, "student prediction L2")
    named = dict(model.nn_module.named_parameters())
    tot = math.sqrt(sum(float(v.grad.double().norm()) ** 2 for v in named.values()))
    _close(tot, z["grad_total_norm"], 5e-3, "total gradient norm")
    _close([float(named[f"readouts.{m}.layer.1.weight"].grad.double().norm()) for m in range(10)], z["grad_readout_w"], 5e-3,
           "readout gradient norms")
    _close([float(named[k].grad.double().norm()) for k in TRACKED], z["grad_tracked"], 5e-3, "trunk gradient norms")


def test_sparse_readout_backward_equals_dense_with_one_hot_mice():
    """Config 3's batch structure (src/datasets.py:172-187: one-hot mouse weights): MouseModel.train_step runs every readout's
    backward on its own samples only (the other rows' loss gradient is exactly zero, src/losses.py:15-17).  Same loss, same
    gradients as the dense backward (summation order only); an absent mouse gets zero gradients."""
    from sensorium_amd.argus_models import MouseModel
    from sensorium_amd.synthetic import make_batch
    kw = dict(readout_outputs=(40, 56, 24, 72), core_features=(16, 16, 32), spatial_strides=(2, 1, 2), expansion_ratio=3,
              se_reduce_ratio=4, cortex_features=(64, 128), drop_rate=0.3, drop_path_rate=0.0)
    params = {"nn_module": ("dwiseneuro", kw), "loss": ("mice_poisson", {}), "optimizer": ("AdamW", {"lr": 1e-3, "weight_decay": 0.05}),
              "device": "cuda:0", "amp": False, "iter_size": 1}
    x, (targets, weights) = make_batch(6, 4, 12, 16, kw["readout_outputs"][:3] + (72,), seed=5, device=dev())
    host = weights._dwn_host.clone()
    host[:, 3] = 0.0                                     # mouse 3 owns no sample of this batch; mice 0-2 own two each
    host[5] = torch.tensor([0.0, 1.0, 0.0, 0.0])
    weights.copy_(host)
    for m in range(4):
        targets[m] *= host[:, m].to(dev())[:, None, None]
    results = {}
    for mode in ("dense", "sparse"):
        torch.manual_seed(3)
        torch.cuda.manual_seed_all(3)
        model = MouseModel(params)
        w = weights.clone()
        if mode == "sparse":
            w._dwn_host = host
        out = model.train_step([x, [targets, w]])
        grads = {k: (None if p.grad is None else p.grad.detach().clone()) for k, p in model.nn_module.named_parameters()}
        results[mode] = (out["loss"], grads)
        if mode == "sparse":
            assert model._active_cache[1] is not None and [int(i.numel()) for i in model._active_cache[1]] == [2, 2, 1, 0]
        else:
            assert getattr(model, "_active_cache", None) is None
    (l0, g0), (l1, g1) = results["dense"], results["sparse"]
    assert abs(l0 - l1) <= 1e-6 * abs(l0)
    tot = math.sqrt(sum(float(g.double().norm()) ** 2 for g in g0.values() if g is not None))
    for k in g0:
        assert (g0[k] is None) == (g1[k] is None), k
        if g0[k] is not None:
            # (floor: the analytically-zero gradients are summation noise of ~1e-6 in either order)
            assert float((g0[k].double() - g1[k].double()).norm()) <= 1e-5 * float(g0[k].double().norm()) + 1e-6 * tot, k
    assert not g1["readouts.3.layer.1.weight"].any() and not g0["readouts.3.layer.1.weight"].any()
