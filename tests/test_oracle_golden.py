"""CPU tests: the oracle (oracle/dwiseneuro_oracle.py) against the committed golden fixtures that
oracle/make_golden.py generated from the real reference (lRomul/sensorium, loaded by file path)."""
import math

import numpy as np
import pytest
import torch

from oracle import dwiseneuro_oracle as orc

TINY = dict(readout_outputs=(7, 10), strides=(2, 1, 2), groups=2, softplus_beta=0.07)


def rel(a, b):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    return float(np.linalg.norm(a - b) / (np.linalg.norm(b) + 1e-30))


def load_case(golden_dir, name):
    z = np.load(golden_dir / name)
    sd = {k[3:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("sd:")}
    return z, sd


def test_index_maps_bit_exact(golden_dir):
    z = np.load(golden_dir / "index_and_pe.npz")
    for k in z.files:
        if k.startswith("nearest_"):
            _, size_in, stride = k.split("_")
            size_in, stride = int(size_in), int(stride)
            assert np.array_equal(orc.nearest_src_index(math.ceil(size_in / stride), size_in), z[k]), k
        elif k.startswith("shuffle_"):
            _, c, g = k.split("_")
            assert np.array_equal(orc.shuffle_source_index(int(c), int(g)), z[k]), k
        elif k.startswith("tile_"):
            _, ci, co = k.split("_")
            assert np.array_equal(orc.tile_channel_index(int(co), int(ci)), z[k]), k
    assert list(orc.nearest_src_index(5, 9)) == [0, 1, 3, 5, 7]
    assert int(z["readout_pad_7863"]) == 7864


def test_positional_encoding_bit_exact(golden_dir):
    z = np.load(golden_dir / "index_and_pe.npz")
    n = 0
    for k in z.files:
        if k.startswith("pe_"):
            _, c, t, h, w = k.split("_")
            mine = orc.pe_table(int(c), int(t), int(h), int(w)).permute(3, 0, 1, 2).numpy()
            assert np.array_equal(mine, z[k]), k
            n += 1
    assert n >= 5
    assert orc.pe_num_channels(64) == 22 and orc.pe_num_channels(128) == 44 and orc.pe_num_channels(256) == 86


def test_softplus(golden_dir):
    z = np.load(golden_dir / "index_and_pe.npz")
    out = orc.softplus(torch.from_numpy(z["softplus_in"]), 0.07).numpy()
    assert rel(out, z["softplus_out"]) < 1e-6


@pytest.mark.parametrize("training", [False, True])
def test_tiny_model_forward_backward(golden_dir, training):
    z, sd = load_case(golden_dir, "tiny_model_train.npz" if training else "tiny_model_eval.npz")
    grads = {k[5:]: z[k] for k in z.files if k.startswith("grad:")}
    sd = {k: (v.clone().requires_grad_(True) if k in grads else v) for k, v in sd.items()}
    x = torch.from_numpy(z["x"])
    targets = [torch.from_numpy(z[f"target_{m}"]) for m in range(2)]
    w = torch.from_numpy(z["mice_weights"])
    new_stats = {}
    preds = orc.forward(sd, x, training=training, new_stats=new_stats, **TINY)
    loss = orc.mice_poisson_loss(preds, targets, w)
    loss.backward()
    for m in range(2):
        assert preds[m].shape == z[f"pred_{m}"].shape
        assert rel(preds[m].detach().numpy(), z[f"pred_{m}"]) < 2e-5
    assert abs(float(loss.detach()) - float(z["loss"])) <= 1e-4 * max(1.0, abs(float(z["loss"])))
    gnorm = math.sqrt(sum(float((g.astype(np.float64) ** 2).sum()) for g in grads.values()))
    for k, g in grads.items():
        err = np.linalg.norm(sd[k].grad.numpy().astype(np.float64) - g) / (np.linalg.norm(g) + 1e-4 * gnorm)
        assert err < 5e-3, (k, err)
    if training:
        for k in z.files:
            if k.startswith("newsd:"):
                name = k[6:]
                got = new_stats[name].detach().numpy()
                if got.dtype.kind == "f":
                    assert rel(got, z[k]) < 1e-5, name
                else:
                    assert int(got) == int(z[k])


def test_tiny_model_float64_agrees(golden_dir):
    """fp64 run of the oracle vs the reference's fp32 outputs: bounds the reference's own rounding."""
    z, sd = load_case(golden_dir, "tiny_model_eval.npz")
    sd64 = {k: (v.double() if v.is_floating_point() else v) for k, v in sd.items()}
    preds = orc.forward(sd64, torch.from_numpy(z["x"]).double(), **TINY)
    for m in range(2):
        assert rel(preds[m].numpy(), z[f"pred_{m}"]) < 2e-5


def test_adamw_and_ema(golden_dir):
    z = np.load(golden_dir / "adamw_ema.npz")
    p = torch.from_numpy(z["p0"])
    m = torch.zeros_like(p)
    v = torch.zeros_like(p)
    ema = p.clone()
    for i in range(3):
        p, m, v = orc.adamw_step(p, torch.from_numpy(z[f"grad_{i}"]), m, v, i + 1, float(z["lr"]),
                                 weight_decay=float(z["wd"]))
        ema = orc.ema_update(ema, p, float(z["decay"]))
        assert rel(p.numpy(), z[f"p_{i + 1}"]) < 1e-6
        assert rel(ema.numpy(), z[f"ema_{i + 1}"]) < 1e-6
    assert rel(m.numpy(), z["exp_avg"]) < 1e-6 and rel(v.numpy(), z["exp_avg_sq"]) < 1e-6
    # int64 num_batches_tracked is truncated by copy_ (ema.py:52)
    nbt = torch.tensor(0, dtype=torch.int64)
    for step in range(1, 4):
        nbt = orc.ema_update(nbt, torch.tensor(step, dtype=torch.int64), float(z["decay"]))
    assert int(nbt) == int(z["nbt_ema"])


def test_predict_trial_blend(golden_dir):
    z = np.load(golden_dir / "predict_trial.npz")
    _, sd = load_case(golden_dir, "tiny_model_eval.npz")
    with torch.no_grad():
        out = orc.predict_trial(lambda win: orc.forward(sd, win, index=1, **TINY)[0],
                                torch.from_numpy(z["inputs"]), 10, size=int(z["size"]), step=int(z["step"]))
    assert rel(out, z["responses"]) < 2e-5


def test_corr(golden_dir):
    z = np.load(golden_dir / "corr.npz")
    assert np.array_equal(orc.corr(z["a"], z["b"], axis=0), z["corr"])


def test_window_indexes():
    assert orc.window_indexes(30, 16, 2) == list(range(0, 31, 2))


def test_make_state_dict_layout():
    sd = orc.make_state_dict(readout_outputs=(7863,), expansion_ratio=7)
    assert sd["core.blocks.1.conv_pw.0.weight"].shape == (448, 64, 1, 1, 1)
    assert sd["core.blocks.1.spat_covn_dw.0.weight"].shape == (448, 1, 1, 3, 3)
    assert sd["core.blocks.1.temp_covn_dw.0.weight"].shape == (448, 1, 5, 1, 1)
    assert sd["core.blocks.17.conv_pwl.0.weight"].shape == (256, 1792, 1, 1, 1)
    assert sd["cortex.layers.2.conv.weight"].shape == (4096, 1024, 1)
    assert sd["readouts.0.layer.1.weight"].shape == (7864, 2048, 1)
    n_params = sum(v.numel() for k, v in sd.items() if "running" not in k and "num_batches" not in k
                   and "inv_freq" not in k)
    assert abs(n_params - 25.19e6) < 0.02e6        # 25.19 M params (1 mouse, exp 7)


def test_library_conv_path_equals_stencil_path(golden_dir):
    """bench.py's cpu_baseline uses the oracle with depth-wise convs routed through torch's conv3d (what the
    reference calls on CPU); it must compute the same function as the explicit-stencil oracle."""
    z, sd = load_case(golden_dir, "tiny_model_eval.npz")
    x = torch.from_numpy(z["x"])
    a = orc.forward(sd, x, training=True, **TINY)
    try:
        orc.DW_IMPL = "library"
        b = orc.forward(sd, x, training=True, **TINY)
    finally:
        orc.DW_IMPL = "stencil"
    for m in range(2):
        assert rel(b[m].numpy(), a[m].numpy()) < 1e-5
