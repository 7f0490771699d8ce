"""GPU parity of the whole DwiseNeuro path against the golden fixtures generated from the reference
(tests/golden/*.npz, oracle/make_golden.py) and against the CPU oracle: forward (eval + train), Poisson loss,
every parameter gradient, BN buffers, fused AdamW/EMA, channel-shuffle / index ops.
"""
import math

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import dwiseneuro_oracle as orc  # noqa: E402
from tests.gpu_helpers import dev, rel  # noqa: E402

TINY = dict(readout_outputs=(7, 10), in_channels=5, core_features=(8, 8, 16), spatial_strides=(2, 1, 2),
            spatial_kernel=3, temporal_kernel=5, expansion_ratio=3, se_reduce_ratio=4, cortex_features=(32, 64),
            groups=2, softplus_beta=0.07, drop_rate=0.0, drop_path_rate=0.0)


def load_golden(golden_dir, name):
    z = np.load(golden_dir / name)
    sd = {k[3:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("sd:")}
    return z, sd


def build(sd, **over):
    from sensorium_amd import DwiseNeuro
    cfg = dict(TINY)
    cfg.update(over)
    model = DwiseNeuro(**cfg)
    res = model.load_state_dict(sd, strict=True)
    assert not res.missing_keys and not res.unexpected_keys
    return model.to(dev())


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_tiny_model_eval_matches_reference(golden_dir, dtype):
    z, sd = load_golden(golden_dir, "tiny_model_eval.npz")
    model = build(sd, compute_dtype=dtype).eval()
    x = torch.from_numpy(z["x"]).to(dev())
    with torch.no_grad():
        preds = model(x)
        p1 = model(x, 1)
    for m in range(2):
        assert preds[m].shape == z[f"pred_{m}"].shape and preds[m].dtype == torch.float32
        e = rel(preds[m], torch.from_numpy(z[f"pred_{m}"]))
        assert e < (1e-3 if dtype == torch.float32 else 3e-2), (m, e)
    # the eval forward is exact from call to call on the product build: the SqueezeExcite pooling sums are integer adds
    # (64-bit fixed point), nothing else in the eval path is accumulated across workgroups in floating point
    assert torch.equal(p1, preds[1])
    with torch.no_grad():
        again = model(x)
    assert all(torch.equal(a, b) for a, b in zip(again, preds))


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_tiny_model_train_step_matches_reference(golden_dir, dtype):
    from sensorium_amd import MicePoissonLoss
    z, sd = load_golden(golden_dir, "tiny_model_train.npz")
    model = build(sd, compute_dtype=dtype).train()
    x = torch.from_numpy(z["x"]).to(dev())
    targets = [torch.from_numpy(z[f"target_{m}"]).to(dev()) for m in range(2)]
    w = torch.from_numpy(z["mice_weights"]).to(dev())
    preds = model(x)
    loss = MicePoissonLoss()(preds, (targets, w))
    loss.backward()
    torch.cuda.synchronize()
    ft, gt = (1e-3, 1e-3) if dtype == torch.float32 else (3e-2, 1e-1)
    for m in range(2):
        assert rel(preds[m], torch.from_numpy(z[f"pred_{m}"])) < ft
    ref_loss = float(z["loss"])
    scale = sum(float(np.abs(z[f"pred_{m}"]).sum()) for m in range(2)) / preds[0].shape[0]
    # the Poisson loss is a cancelling sum (pred - target*log(pred)): its error scales with sum|pred| per sample, not
    # with the net value; fp32 is held to 1e-3 of that sum, bf16 (whose BN statistics see fp32-atomic ordering noise
    # amplified by 8-bit mantissas) to 1e-2 of it
    floor = (1e-3 if dtype == torch.float32 else 1e-2) * scale
    assert abs(float(loss.detach()) - ref_loss) <= ft * max(abs(ref_loss), floor), (float(loss.detach()), ref_loss)
    grads = {k[5:]: z[k] for k in z.files if k.startswith("grad:")}
    gnorm = math.sqrt(sum(float((g.astype(np.float64) ** 2).sum()) for g in grads.values()))
    named = dict(model.named_parameters())
    worst = ("", 0.0)
    for k, g in grads.items():
        mine = named[k].grad
        assert mine is not None, k
        # analytically-zero gradients are rounding noise: floor the denominator by a fraction of
        # the global gradient norm (1e-4 fp32; 1e-2 for bf16 storage)
        floor = (1e-4 if dtype == torch.float32 else 1e-2) * gnorm
        err = float(np.linalg.norm(mine.double().cpu().numpy() - g)) / (float(np.linalg.norm(g)) + floor)
        if err > worst[1]:
            worst = (k, err)
    assert worst[1] < gt, worst
    msd = model.state_dict()
    for k in z.files:
        if k.startswith("newsd:"):
            name = k[6:]
            if msd[name].is_floating_point():
                assert rel(msd[name], torch.from_numpy(z[k])) < (1e-4 if dtype == torch.float32 else 3e-2), name
            else:
                assert int(msd[name]) == int(z[k]), name


def test_single_trial_correlation_parity(golden_dir):
    """corr (src/metrics.py:11-31) of HIP predictions vs the reference's predictions against the same targets."""
    z, sd = load_golden(golden_dir, "tiny_model_eval.npz")
    model = build(sd).eval()
    with torch.no_grad():
        preds = model(torch.from_numpy(z["x"]).to(dev()))
    for m in range(2):
        t = z[f"target_{m}"].transpose(0, 2, 1).reshape(-1, z[f"target_{m}"].shape[1])
        mine = preds[m].cpu().numpy().transpose(0, 2, 1).reshape(t.shape)
        refp = z[f"pred_{m}"].transpose(0, 2, 1).reshape(t.shape)
        c_mine, c_ref = orc.corr(mine, t, axis=0).mean(), orc.corr(refp, t, axis=0).mean()
        assert abs(c_mine - c_ref) <= 1e-4


def test_full_width_model_digest(golden_dir):
    """exp-7, 1 mouse (7863 neurons), B=2, T=8, 36x64, train mode: scalar digests from the reference."""
    from sensorium_amd import DwiseNeuro, MicePoissonLoss
    z = np.load(golden_dir / "full_width_digest.npz")
    sd = orc.make_state_dict(readout_outputs=(7863,), expansion_ratio=7, seed=11)
    model = DwiseNeuro(readout_outputs=(7863,), expansion_ratio=7, drop_rate=0.0, drop_path_rate=0.0)
    model.load_state_dict(sd, strict=True)
    model = model.to(dev()).train()
    rng = np.random.default_rng(20231122)
    b, t, h, w = 2, 8, 36, 64
    x = np.zeros((b, 5, t, h, w), dtype=np.float32)
    x[:, 0] = rng.integers(0, 256, size=(b, t, h, w)).astype(np.float32)
    scale = np.array([10, 5, 20, 20], dtype=np.float32)
    shift = np.array([30, 5, 100, 70], dtype=np.float32)
    beh = np.clip(rng.normal(size=(b, 4, t)).astype(np.float32) * scale[None, :, None] + shift[None, :, None], 0, None)
    x[:, 1:] = beh[:, :, :, None, None]
    target = np.maximum(rng.normal(size=(b, 7863, t)), 0).astype(np.float32) * 10
    weights = np.ones((b, 1), dtype=np.float32)
    preds = model(torch.from_numpy(x).to(dev()))
    loss = MicePoissonLoss()(preds, ([torch.from_numpy(target).to(dev())], torch.from_numpy(weights).to(dev())))
    loss.backward()
    torch.cuda.synchronize()
    p = preds[0].detach()
    assert abs(float(loss) - float(z["loss"])) <= 1e-3 * abs(float(z["loss"]))
    assert abs(float(p.mean()) - float(z["pred_mean"])) <= 1e-3 * abs(float(z["pred_mean"]))
    assert abs(float(p.double().norm()) - float(z["pred_l2"])) <= 1e-3 * float(z["pred_l2"])
    named = dict(model.named_parameters())
    tot = math.sqrt(sum(float(v.grad.double().norm()) ** 2 for v in named.values()))
    assert abs(tot - float(z["grad_total_norm"])) <= 5e-3 * float(z["grad_total_norm"])
    for key, name in (("grad_stem", "core.stem.0.weight"), ("grad_readout_w", "readouts.0.layer.1.weight"),
                      ("grad_pw0", "core.blocks.1.conv_pw.0.weight"), ("grad_cortex2", "cortex.layers.2.conv.weight")):
        got = float(named[name].grad.double().norm())
        assert abs(got - float(z[key])) <= 5e-3 * float(z[key]), (name, got, float(z[key]))


def test_cortex_layer_matches_oracle_on_integer_input(golden_dir):
    """A cortex layer with routing-only weights against the oracle (the permutation / tile maps themselves are recovered from
    the GPU output and compared with the golden maps in tests/test_gpu_index_ops.py)."""
    from sensorium_amd.dwiseneuro import ShuffleLayer
    cin, c, groups = 16, 32, 2
    layer = ShuffleLayer(cin, c, groups=groups).to(dev()).eval()
    with torch.no_grad():
        w = torch.zeros(c, cin // groups, 1)
        for o in range(c):                       # output channel o copies input channel (o % (cin/groups)) of its group
            w[o, o % (cin // groups), 0] = 1.0
        layer.conv.weight.copy_(w)
        for bn in (layer.bn.bn, layer.bn_sc.bn):  # identity BN in eval mode
            bn.weight.fill_(1.0); bn.bias.zero_(); bn.running_mean.zero_(); bn.running_var.fill_(1.0 - bn.eps)
    x = (torch.arange(2 * 3 * cin, dtype=torch.float32).reshape(2, 3, cin) % 11) + 1.0     # positive small ints
    with torch.no_grad():
        out = layer(x.to(dev()), torch.float32).cpu()
    sdict = {"l." + k: v.cpu() for k, v in layer.state_dict().items()}
    ref = orc.cortex_layer(x, "l", sdict, groups, False, None, None)
    assert rel(out, ref) < 1e-6


def test_adamw_ema_multi_matches_reference(golden_dir):
    from sensorium_amd.optim import FusedAdamWEma
    z = np.load(golden_dir / "adamw_ema.npz")
    p = torch.nn.Parameter(torch.from_numpy(z["p0"]).to(dev()))
    ema = torch.from_numpy(z["p0"]).to(dev()).clone()
    opt = FusedAdamWEma([p], lr=float(z["lr"]), weight_decay=float(z["wd"]), ema_params=[ema],
                        ema_decay=float(z["decay"]))
    for i in range(3):
        p.grad = torch.from_numpy(z[f"grad_{i}"]).to(dev())
        opt.step()
        torch.cuda.synchronize()
        assert rel(p, torch.from_numpy(z[f"p_{i + 1}"])) < 1e-6
        assert rel(ema, torch.from_numpy(z[f"ema_{i + 1}"])) < 1e-6
    st = opt.state_for(p)
    assert rel(st["exp_avg"], torch.from_numpy(z["exp_avg"])) < 1e-6
    assert rel(st["exp_avg_sq"], torch.from_numpy(z["exp_avg_sq"])) < 1e-6
