#!/usr/bin/env python3
"""Generate the golden fixtures under tests/golden/ from the REAL reference (build container only).

Run:  python oracle/make_golden.py          (needs /root/reference; never runs on the GPU box)

The reference modules are loaded *by file path* (``src/__init__.py`` pulls in ``argus``, which is
not installed — SURVEY.md §8c).  For every case the script (1) runs the reference, (2) runs
``oracle/dwiseneuro_oracle.py`` on the same weights/inputs and asserts agreement, (3) stores
inputs + reference outputs as small ``.npz`` fixtures.  Only data is stored (inputs, weights drawn
from a seeded numpy rng, expected outputs) — no reference source travels.
"""
from __future__ import annotations

import importlib.util
import math
import sys
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parents[1]
REF = Path("/root/reference")
GOLD = ROOT / "tests" / "golden"
sys.path.insert(0, str(ROOT))

from oracle import dwiseneuro_oracle as orc  # noqa: E402


def load_by_path(name: str, rel: str):
    spec = importlib.util.spec_from_file_location(name, REF / rel)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)  # type: ignore[union-attr]
    return mod


ref_model = load_by_path("ref_dwiseneuro", "src/models/dwiseneuro.py")
ref_losses = load_by_path("ref_losses", "src/losses.py")
ref_utils = load_by_path("ref_utils", "src/utils.py")

TINY = dict(
    readout_outputs=(7, 10),
    in_channels=5,
    core_features=(8, 8, 16),
    spatial_strides=(2, 1, 2),
    spatial_kernel=3,
    temporal_kernel=5,
    expansion_ratio=3,
    se_reduce_ratio=4,
    cortex_features=(32, 64),
    groups=2,
    softplus_beta=0.07,
    drop_rate=0.0,
    drop_path_rate=0.0,
)


def synth_inputs(rng, b, t, h, w, readout_outputs):
    """Synthetic clip per SURVEY.md §8d: ch0 video 0..255, ch1-4 per-(b,t) scalars broadcast over HxW."""
    x = np.zeros((b, 5, t, h, w), dtype=np.float32)
    x[:, 0] = rng.integers(0, 256, size=(b, t, h, w)).astype(np.float32)
    scale = np.array([10, 5, 20, 20], dtype=np.float32)
    shift = np.array([30, 5, 100, 70], dtype=np.float32)
    beh = np.clip(rng.normal(size=(b, 4, t)).astype(np.float32) * scale[None, :, None] + shift[None, :, None], 0, None)
    x[:, 1:] = beh[:, :, :, None, None]
    targets = [np.maximum(rng.normal(size=(b, n, t)), 0).astype(np.float32) * 10 for n in readout_outputs]
    weights = np.zeros((b, len(readout_outputs)), dtype=np.float32)
    for i in range(b):
        weights[i, i % len(readout_outputs)] = 1.0
    return x, targets, weights


def rel_err(a, b):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    return float(np.linalg.norm(a - b) / (np.linalg.norm(b) + 1e-30))


def tiny_model_case(training: bool, dtype=torch.float32):
    rng = np.random.default_rng(1234 + int(training))
    sd = orc.make_state_dict(
        readout_outputs=TINY["readout_outputs"], core_features=TINY["core_features"],
        expansion_ratio=TINY["expansion_ratio"], se_reduce_ratio=TINY["se_reduce_ratio"],
        cortex_features=TINY["cortex_features"], groups=TINY["groups"], seed=77, randomize_bn=True)
    model = ref_model.DwiseNeuro(**TINY).to(dtype)
    missing = model.load_state_dict({k: v.to(dtype) if v.is_floating_point() else v for k, v in sd.items()},
                                    strict=True)
    assert not missing.missing_keys and not missing.unexpected_keys
    assert list(model.state_dict().keys()) == list(sd.keys()), "oracle key order != reference key order"
    model.train(training)
    b, t, h, w = 3, 6, 9, 11
    x, targets, weights = synth_inputs(rng, b, t, h, w, TINY["readout_outputs"])
    xt = torch.from_numpy(x).to(dtype)
    tt = [torch.from_numpy(a).to(dtype) for a in targets]
    wt = torch.from_numpy(weights).to(dtype)

    preds = model(xt)
    loss = ref_losses.MicePoissonLoss()(preds, (tt, wt))
    loss.backward()
    ref_grads = {k: p.grad.detach().clone() for k, p in model.named_parameters()}
    ref_new_sd = {k: v.detach().clone() for k, v in model.state_dict().items()}

    # ---- oracle on the same data
    sd_o = {k: (v.to(dtype).clone().requires_grad_(True) if (v.is_floating_point() and k in ref_grads) else
                (v.to(dtype) if v.is_floating_point() else v)) for k, v in sd.items()}
    new_stats: dict = {}
    preds_o = orc.forward(sd_o, xt, strides=TINY["spatial_strides"], readout_outputs=TINY["readout_outputs"],
                          groups=TINY["groups"], softplus_beta=TINY["softplus_beta"], training=training,
                          new_stats=new_stats)
    loss_o = orc.mice_poisson_loss(preds_o, tt, wt)
    loss_o.backward()
    for m in range(len(preds)):
        e = rel_err(preds_o[m].detach(), preds[m].detach())
        assert e < 2e-5, ("pred", m, e)
    e = rel_err(loss_o.detach(), loss.detach())
    assert e < 1e-5, ("loss", e)
    worst = 0.0
    gnorm = math.sqrt(sum(float((g.double() ** 2).sum()) for g in ref_grads.values()))
    for k, g in ref_grads.items():
        go = sd_o[k].grad
        # 21 grads are analytically zero in train mode: compare absolutely vs global norm
        err = float((go.double() - g.double()).norm()) / (float(g.double().norm()) + 1e-4 * gnorm)
        worst = max(worst, err)
        assert err < 5e-3, ("grad", k, err, float(g.double().norm()), gnorm)
    if training:
        for k, v in new_stats.items():
            e = rel_err(v, ref_new_sd[k]) if v.is_floating_point() else float(v != ref_new_sd[k])
            assert e < 1e-5, ("bn buffer", k, e)
    # single index path (dwiseneuro.py:404-405)
    if not training:
        with torch.no_grad():
            p1 = model(xt, 1)
            p1o = orc.forward(sd_o, xt, strides=TINY["spatial_strides"], readout_outputs=TINY["readout_outputs"],
                              groups=TINY["groups"], softplus_beta=TINY["softplus_beta"], index=1)
        assert rel_err(p1o, p1) < 2e-5
    print(f"tiny model training={training}: oracle==reference (worst grad err {worst:.2e}, loss {float(loss.detach()):.4f})")

    out = {"x": x, "mice_weights": weights, "loss": np.float64(loss.detach().double().item())}
    for m, a in enumerate(targets):
        out[f"target_{m}"] = a
        out[f"pred_{m}"] = preds[m].detach().float().numpy()
    for k, v in sd.items():
        out["sd:" + k] = v.numpy()
    for k, g in ref_grads.items():
        out["grad:" + k] = g.float().numpy()
    if training:
        for k, v in ref_new_sd.items():
            if "running_" in k or "num_batches" in k:
                out["newsd:" + k] = v.numpy()
    GOLD.mkdir(parents=True, exist_ok=True)
    name = "tiny_model_train.npz" if training else "tiny_model_eval.npz"
    np.savez_compressed(GOLD / name, **out)
    return sd, model


def index_and_pe_cases():
    out = {}
    # nearest interpolate source indices vs F.interpolate (dwiseneuro.py:127-129)
    for size_in in (36, 18, 9, 5, 64, 32, 16, 8, 11, 7):
        for stride in (2, 3):
            size_out = math.ceil(size_in / stride)
            probe = torch.arange(size_in, dtype=torch.float32).view(1, 1, 1, 1, size_in)
            got = torch.nn.functional.interpolate(probe, size=(1, 1, size_out), mode="nearest").view(-1).long().numpy()
            mine = orc.nearest_src_index(size_out, size_in)
            assert np.array_equal(got, mine), (size_in, stride, got, mine)
            out[f"nearest_{size_in}_{stride}"] = got
    # channel shuffle / tile via the reference's own methods
    for c, g in ((8, 2), (64, 2), (4096, 2), (12, 3)):
        layer = ref_model.ShuffleLayer(c, c, groups=g)
        probe = torch.arange(c, dtype=torch.float32).view(1, c, 1)
        got = layer.shuffle_channels(probe).view(-1).long().numpy()
        assert np.array_equal(got, orc.shuffle_source_index(c, g))
        out[f"shuffle_{c}_{g}"] = got
    for c_in, c_out in ((64, 128), (256, 1024), (8, 20), (16, 16)):
        layer = ref_model.ShuffleLayer(c_in, c_out, groups=1)
        layer.bn_sc = torch.nn.Identity()
        probe = torch.arange(c_in, dtype=torch.float32).view(1, c_in, 1)
        got = layer.tile_shortcut(probe).view(-1).long().numpy()
        assert np.array_equal(got, orc.tile_channel_index(c_out, c_in))
        out[f"tile_{c_in}_{c_out}"] = got
    # positional encoding tables vs the reference module (bit-exact closed form)
    for c, (t, h, w) in ((64, (4, 5, 6)), (128, (3, 9, 16)), (256, (2, 5, 8)), (8, (6, 9, 11)), (20, (3, 4, 5))):
        pe = ref_model.PositionalEncoding3d(c)
        enc = pe.create_cached_encoding(torch.zeros(1, c, t, h, w))[0]          # [C,t,h,w]
        mine = orc.pe_table(c, t, h, w).permute(3, 0, 1, 2)
        assert torch.equal(enc, mine), ("pe", c)
        out[f"pe_{c}_{t}_{h}_{w}"] = enc.numpy()
    # readout pad/slice (dwiseneuro.py:278,285): 7863 -> 7864 -> 7863
    r = ref_model.Readout(64, 7863, groups=2, softplus_beta=0.07)
    out["readout_pad_7863"] = np.array(r.layer[1].weight.shape[0])
    assert int(out["readout_pad_7863"]) == 7864
    # softplus beta/threshold behaviour
    z = torch.linspace(-400, 400, 201)
    sp = torch.nn.Softplus(beta=0.07)(z)
    assert rel_err(orc.softplus(z, 0.07), sp) < 1e-6
    out["softplus_in"] = z.numpy()
    out["softplus_out"] = sp.numpy()
    np.savez_compressed(GOLD / "index_and_pe.npz", **out)
    print("index / PE / softplus cases: oracle==reference (bit-exact for index ops and PE)")


def optimizer_ema_case():
    """AdamW (true_batch_001.py:45-48) + ModelEma.update (ema.py:47-55) over 3 steps, hand-driven."""
    rng = np.random.default_rng(5)
    p0 = rng.normal(size=(4, 33)).astype(np.float32)
    grads = [rng.normal(size=(4, 33)).astype(np.float32) * s for s in (1.0, 0.1, 3.0)]
    lr, wd, decay = 2.4e-3, 0.05, 0.999
    p = torch.nn.Parameter(torch.from_numpy(p0.copy()))
    opt = torch.optim.AdamW([p], lr=lr, weight_decay=wd)
    ema = torch.from_numpy(p0.copy())
    nbt_model = torch.tensor(0, dtype=torch.int64)
    nbt_ema = torch.tensor(0, dtype=torch.int64)
    po, mo, vo = torch.from_numpy(p0.copy()), torch.zeros(4, 33), torch.zeros(4, 33)
    emao = torch.from_numpy(p0.copy())
    out = {"p0": p0, "lr": np.float64(lr), "wd": np.float64(wd), "decay": np.float64(decay)}
    for i, g in enumerate(grads):
        p.grad = torch.from_numpy(g.copy())
        opt.step()
        nbt_model = nbt_model + 1
        with torch.no_grad():
            ema.copy_(decay * ema + (1.0 - decay) * p.detach())
            nbt_ema.copy_(decay * nbt_ema + (1.0 - decay) * nbt_model)       # int64 truncation
        po, mo, vo = orc.adamw_step(po, torch.from_numpy(g), mo, vo, i + 1, lr, weight_decay=wd)
        emao = orc.ema_update(emao, po, decay)
        assert rel_err(po, p.detach()) < 1e-6 and rel_err(emao, ema) < 1e-6
        out[f"grad_{i}"] = g
        out[f"p_{i + 1}"] = p.detach().numpy().copy()
        out[f"ema_{i + 1}"] = ema.numpy().copy()
    st = opt.state[p]
    out["exp_avg"] = st["exp_avg"].numpy()
    out["exp_avg_sq"] = st["exp_avg_sq"].numpy()
    out["nbt_ema"] = nbt_ema.numpy()
    assert int(orc.ema_update(torch.tensor(0, dtype=torch.int64), torch.tensor(3, dtype=torch.int64), decay)) == 0
    np.savez_compressed(GOLD / "adamw_ema.npz", **out)
    print("AdamW + EMA 3-step case: oracle==torch.optim.AdamW / ema.py semantics")


def predictor_case(sd, model):
    """Sliding-window blend of predictors.py:46-54 with the tiny eval model (frame stack 4, step 2)."""
    model.eval()
    rng = np.random.default_rng(9)
    length, size, step = 14, 4, 2
    inputs = torch.from_numpy(rng.normal(size=(5, length, 9, 11)).astype(np.float32) * 20 + 50)
    n = TINY["readout_outputs"][1]
    responses = np.zeros((n, length), dtype=np.float32)
    blend = np.zeros(length, np.float32)
    behind = (size - 1) * step
    with torch.no_grad():
        for index in range(behind, length):
            idx = list(range(index - behind, index + 1, step))
            pred = model(inputs[:, idx].unsqueeze(0), 1)[0]
            responses[..., idx] += pred.numpy()
            blend[idx] += np.ones(size, dtype=np.float32)
    responses /= np.clip(blend, 1.0, None)
    sdo = {k: v for k, v in sd.items()}
    with torch.no_grad():
        mine = orc.predict_trial(
            lambda win: orc.forward(sdo, win, strides=TINY["spatial_strides"],
                                    readout_outputs=TINY["readout_outputs"], index=1)[0],
            inputs, n, size=size, step=step)
    assert rel_err(mine, responses) < 2e-5
    np.savez_compressed(GOLD / "predict_trial.npz", inputs=inputs.numpy(), responses=responses,
                        size=np.array(size), step=np.array(step))
    print("predict_trial sliding-window case: oracle==reference loop")


def corr_case():
    src = (REF / "src/metrics.py").read_text().split("class CorrelationMetric")[0]
    src = src.replace("from argus.metrics import Metric", "")
    ns: dict = {}
    exec(compile(src, "ref_metrics_corr", "exec"), ns)
    rng = np.random.default_rng(3)
    a = rng.normal(size=(50, 13)).astype(np.float32)
    b = (a * 0.3 + rng.normal(size=(50, 13))).astype(np.float32)
    got = ns["corr"](a, b, axis=0)
    assert np.allclose(got, orc.corr(a, b, axis=0), rtol=0, atol=0)
    np.savez_compressed(GOLD / "corr.npz", a=a, b=b, corr=got)
    print("corr case: oracle==reference")


def full_size_digest():
    """Scalar digests of the full-width model (exp 7, 1 mouse) at a small clip, eval + train loss."""
    torch.manual_seed(0)
    cfg = dict(readout_outputs=(7863,), expansion_ratio=7, drop_rate=0.0, drop_path_rate=0.0)
    sd = orc.make_state_dict(readout_outputs=(7863,), expansion_ratio=7, seed=11)
    model = ref_model.DwiseNeuro(**cfg)
    model.load_state_dict(sd, strict=True)
    rng = np.random.default_rng(20231122)
    x, targets, weights = synth_inputs(rng, 2, 8, 36, 64, (7863,))
    xt = torch.from_numpy(x)
    model.train()
    preds = model(xt)
    loss = ref_losses.MicePoissonLoss()(preds, ([torch.from_numpy(targets[0])], torch.from_numpy(weights)))
    loss.backward()
    gn = {k: float(p.grad.double().norm()) for k, p in model.named_parameters()}
    sd_o = {k: (v.clone().requires_grad_(True) if k in gn else v) for k, v in sd.items()}
    preds_o = orc.forward(sd_o, xt, strides=(2, 1, 1, 1, 2, 1, 1, 2, 1), readout_outputs=(7863,), training=True)
    loss_o = orc.mice_poisson_loss(preds_o, [torch.from_numpy(targets[0])], torch.from_numpy(weights))
    loss_o.backward()
    assert rel_err(preds_o[0].detach(), preds[0].detach()) < 1e-4
    assert rel_err(loss_o.detach(), loss.detach()) < 1e-5
    tot = math.sqrt(sum(v * v for v in gn.values()))
    for k in gn:
        err = float((sd_o[k].grad.double() - dict(model.named_parameters())[k].grad.double()).norm()) / (gn[k] + 1e-5 * tot)
        assert err < 2e-2, (k, err)
    p = preds[0].detach()
    digest = dict(loss=float(loss), pred_mean=float(p.mean()), pred_std=float(p.std()),
                  pred_l2=float(p.double().norm()), grad_total_norm=tot,
                  grad_stem=gn["core.stem.0.weight"], grad_readout_w=gn["readouts.0.layer.1.weight"],
                  grad_pw0=gn["core.blocks.1.conv_pw.0.weight"], grad_cortex2=gn["cortex.layers.2.conv.weight"])
    np.savez_compressed(GOLD / "full_width_digest.npz", **{k: np.float64(v) for k, v in digest.items()})
    print("full-width (exp7, 1 mouse, B=2,T=8,36x64) digest: oracle==reference;", digest)


def main():
    torch.set_num_threads(8)
    index_and_pe_cases()
    sd, model = tiny_model_case(training=False)
    tiny_model_case(training=True)
    # float64 run of the same oracle: "ground truth" error of the fp32 reference itself
    optimizer_ema_case()
    predictor_case(sd, model)
    corr_case()
    full_size_digest()
    print("fixtures written to", GOLD)


if __name__ == "__main__":
    main()
