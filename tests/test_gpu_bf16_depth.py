"""bf16 storage (the benchmarked dtype) at FULL width and depth — expansion 7, nine blocks, 7863 neurons — against the
reference's digest and against the fp32 HIP path (which is pinned to the reference at 1e-3).  Bounds are ~3x the errors
measured by tests/bf16_parity_report.py (profiles/r2_bf16_parity.json): loss 5e-5, gradient norm 2e-3, worst gradient cosine 0.983,
30-step loss gap 6e-4 of the loss drop, |corr difference| 4e-4."""
import math

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import dwiseneuro_oracle as orc  # noqa: E402
from tests.gpu_helpers import analytically_zero_grad, dev, synth_inputs  # noqa: E402


def _model():
    from sensorium_amd import DwiseNeuro
    sd = orc.make_state_dict(readout_outputs=(7863,), expansion_ratio=7, seed=11)
    m = DwiseNeuro(readout_outputs=(7863,), expansion_ratio=7, drop_rate=0.0, drop_path_rate=0.0)
    m.load_state_dict(sd, strict=True)
    return m.to(dev()).train()


def _batch():
    rng = np.random.default_rng(20231122)
    x, targets, _ = synth_inputs(rng, 2, 8, 36, 64, (7863,))
    return torch.from_numpy(x).to(dev()), torch.from_numpy(targets[0]).to(dev()), torch.ones(2, 1, device=dev()), targets[0]


def _fwd_bwd(model, x, t, w, bf16):
    from sensorium_amd import MicePoissonLoss
    model.zero_grad(set_to_none=True)
    with torch.autocast("cuda", dtype=torch.bfloat16, enabled=bf16):
        preds = model(x)
        loss = MicePoissonLoss()(preds, ([t], w))
    loss.backward()
    torch.cuda.synchronize()
    return float(loss.detach()), preds[0].detach().float(), {k: p.grad.detach().double().clone() for k, p in model.named_parameters()}


def test_bf16_full_width_digest_and_gradient_directions(golden_dir):
    z = np.load(golden_dir / "full_width_digest.npz")
    model = _model()
    x, t, w, _ = _batch()
    l32, p32, g32 = _fwd_bwd(model, x, t, w, False)
    l16, p16, g16 = _fwd_bwd(model, x, t, w, True)
    # bf16 against the reference's digest
    assert abs(l16 - float(z["loss"])) <= 2e-4 * abs(float(z["loss"]))
    assert abs(float(p16.double().norm()) - float(z["pred_l2"])) <= 1e-3 * float(z["pred_l2"])
    tot16 = math.sqrt(sum(float(g.norm()) ** 2 for g in g16.values()))
    tot32 = math.sqrt(sum(float(g.norm()) ** 2 for g in g32.values()))
    # Round-4 root cause (profiles/r4_bf16_parity.json, tests/bf16_gain_report.py): bf16 storage leaves every core gradient with
    # 6-9 % of rounding NOISE (one stage in bf16 alone: 0.4-3.5 %, adding in quadrature over stem + 9 blocks + head), not a gain
    # error: the GAIN <g16, g32> / |g32|^2 of the whole gradient vector is 1 to a few 1e-4, and noise can only ADD length —
    # |g16| = |g32| sqrt(gain^2 + rho^2), rho ~ 0.08-0.11 => +0.3 ... +0.9 % — which is what the round-3 "+0.62 %" was.  (The
    # stem weight's x1.04-1.07 at this batch is that noise projected on a 320-element gradient: 0.98-1.00 at B=32, T=32, below.)
    # So: the 8e-3 bound sits on the gain, where a systematic error would show; the length is bounded through the noise.
    dot = sum(float((g16[k] * g32[k]).sum()) for k in g32)
    gain = dot / tot32 ** 2
    rho = math.sqrt(max(tot16 ** 2 / tot32 ** 2 - gain ** 2, 0.0))
    # Round 5 (tools/gain_probe.py, twelve passes on one box, both y1 modes): a single pass's gain scatters 1.0024 ... 1.0080 from
    # run to run (the product build re-draws the bf16 rounding noise with its summation order), so ONE draw against 8e-3 failed
    # every ~15th run; the bound stays where it was, on the MEAN of four draws (scatter / 2), which is what a systematic gain
    # error would move.
    gains = [gain]
    for _ in range(3):
        _, _, gx = _fwd_bwd(model, x, t, w, True)
        gains.append(sum(float((gx[k] * g32[k]).sum()) for k in g32) / tot32 ** 2)
    gain_mean = sum(gains) / len(gains)
    assert abs(gain_mean - 1.0) <= 8e-3, gains
    assert all(abs(gv - 1.0) <= 1.4e-2 for gv in gains), gains
    assert rho <= 0.15, rho
    assert abs(tot16 - float(z["grad_total_norm"])) <= 2e-2 * float(z["grad_total_norm"])
    # bf16 against fp32 HIP, element-wise
    assert float((p16 - p32).norm() / p32.norm()) <= 8e-3
    worst = (1.0, None)
    for k in g32:
        if analytically_zero_grad(k):
            continue
        c = float((g32[k] * g16[k]).sum() / (g32[k].norm() * g16[k].norm()))
        worst = min(worst, (c, k))
    assert worst[0] >= 0.95, worst


@pytest.mark.parametrize("bf16,bound", [(False, 1e-4), (True, 0.25)])
def test_run_to_run_gradient_noise_is_bounded(bf16, bound):
    """The weight gradients are accumulated with fp32 atomics (order varies run to run): the noise on every gradient that is
    not analytically zero stays below `bound` of its norm (measured by tests/bf16_parity_report.py: fp32 1.1e-5; bf16 8e-2, on the
    tiny SE-bias gradients — the large weight gradients are two orders of magnitude quieter)."""
    model = _model()
    x, t, w, _ = _batch()
    _, _, ga = _fwd_bwd(model, x, t, w, bf16)
    _, _, gb = _fwd_bwd(model, x, t, w, bf16)
    worst = max((float((ga[k] - gb[k]).norm() / ga[k].norm()), k) for k in ga if not analytically_zero_grad(k))
    assert worst[0] <= bound, worst


def test_bf16_training_trajectory_tracks_fp32():
    """30 steps of MouseModel.train_step (AdamW + EMA) on the seeded synthetic batch in both modes: the loss curves stay within
    3e-3 of the total loss drop of each other, the single-trial correlation of the final predictions differs by <= 2e-3 and
    the two models' predictions are >= 0.999 correlated."""
    from sensorium_amd.argus_models import MouseModel
    x, t, w, t_np = _batch()
    losses, finals = {}, {}
    for mode, bf in (("fp32", False), ("bf16", True)):
        params = {"nn_module": ("dwiseneuro", dict(readout_outputs=(7863,), expansion_ratio=7, drop_rate=0.0, drop_path_rate=0.0)),
                  "loss": ("mice_poisson", {}), "optimizer": ("AdamW", {"lr": 3e-4, "weight_decay": 0.05}), "device": "cuda:0",
                  "amp": bf, "iter_size": 1}
        mm = MouseModel(params)
        mm.nn_module.load_state_dict(orc.make_state_dict(readout_outputs=(7863,), expansion_ratio=7, seed=11), strict=True)
        mm.set_ema(0.99)
        losses[mode] = np.array([mm.train_step([x, [[t], w]])["loss"] for _ in range(30)])
        mm.eval()
        with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16, enabled=bf):
            finals[mode] = mm.nn_module(x)[0].float().cpu().numpy()
    drop = abs(losses["fp32"][0] - losses["fp32"][-1])
    assert losses["fp32"][-1] < losses["fp32"][0] and losses["bf16"][-1] < losses["bf16"][0]
    assert float(np.max(np.abs(losses["bf16"] - losses["fp32"]))) <= 3e-3 * drop
    tt = t_np.transpose(0, 2, 1).reshape(-1, 7863)
    c32 = orc.corr(finals["fp32"].transpose(0, 2, 1).reshape(tt.shape), tt, axis=0).mean()
    c16 = orc.corr(finals["bf16"].transpose(0, 2, 1).reshape(tt.shape), tt, axis=0).mean()
    assert abs(c32 - c16) <= 2e-3
    assert np.corrcoef(finals["fp32"].ravel(), finals["bf16"].ravel())[0, 1] >= 0.999


def test_bf16_gradient_gain_at_the_metric_batch():
    """B=32, T=32, 36x64 (the benchmarked step), bf16 against the fp32 HIP path (itself pinned to the reference at 1e-3; no
    oracle needed at this size): gain = <g16, g32> / |g32|^2.  Whole gradient: 1 +- 5e-3.  Every parameter carrying >= 1 % of the
    gradient's length: 1 +- 2e-2.  The rest (tiny gradients that are small differences of large sums: the SqueezeExcite reduce
    layers, some BatchNorm biases; measured 0.95-1.15 and moving from run to run with the product build's summation order):
    1 +- 0.25, cosine >= 0.9.  Measured: profiles/r4_bf16_parity.json."""
    model = _model()
    rng = np.random.default_rng(20231122)
    x, targets, _ = synth_inputs(rng, 32, 32, 36, 64, (7863,))
    x, t, w = torch.from_numpy(x).to(dev()), torch.from_numpy(targets[0]).to(dev()), torch.ones(32, 1, device=dev())
    _, _, g32 = _fwd_bwd(model, x, t, w, False)
    _, _, g16 = _fwd_bwd(model, x, t, w, True)
    tot32 = math.sqrt(sum(float(g.norm()) ** 2 for g in g32.values()))
    gain_all = sum(float((g16[k] * g32[k]).sum()) for k in g32) / tot32 ** 2
    assert abs(gain_all - 1.0) <= 5e-3, gain_all
    for k in g32:
        if analytically_zero_grad(k):
            continue
        n32 = float(g32[k].norm())
        gain = float((g16[k] * g32[k]).sum()) / n32 ** 2
        cos = float((g16[k] * g32[k]).sum()) / (n32 * float(g16[k].norm()))
        if k == "core.stem.0.weight":
            # 320 elements, a small correlation of two large tensors at the far end of the bf16 chain.  On THIS data seed it measures
            # 0.984 / 0.998 with y1 unmaterialised and 0.979 / 0.992 with y1 stored (round 6, profiles/r6_stem_gain_seeds.txt): the same
            # in both modes since the rebuilt y1 is no longer rounded and the Gram sums are fp64 (round 5: 0.96-0.98 vs 0.987-1.006).
            # Bound: 5e-2 in either mode (2.5 x the largest deviation seen on this seed).  Across DATA seeds the gain of this one
            # parameter scatters 0.93 ... 1.02, stored and unmaterialised alike — a property of the gradient, not of a mode.
            assert abs(gain - 1.0) <= 5e-2 and cos >= 0.95, (k, gain, cos)
        elif n32 >= 1e-2 * tot32:
            assert abs(gain - 1.0) <= 2e-2, (k, gain)
        else:
            assert abs(gain - 1.0) <= 0.25 and cos >= 0.9, (k, gain, cos)
