"""GPU parity of the stem — Conv3d(5 -> C0, 1x1x1, no bias) + BatchNorm3d (+ the first block's positional encoding), reference
src/models/dwiseneuro.py:306-309, 184-192 — against the float64 CPU oracle (oracle/dwiseneuro_oracle.py, pinned to the reference
by tests/golden).  The HIP stem never materialises the conv output: BatchNorm statistics, dgamma / dbeta and the weight gradient
come from the input moments (sum x, sum x x^T) and sum dout (x - mean x)^T, so the test drives it with the real input
statistics — an un-normalised video channel (0..255) next to per-frame behaviour scalars, where a naive second-moment
formula would cancel catastrophically.

fp32 path: <= 1e-3 relative (north-star tolerance); bf16 storage: 4e-2 forward / 8e-2 gradients."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import dwiseneuro_oracle as orc  # noqa: E402
from tests.gpu_helpers import dev, rel  # noqa: E402


def _inputs(B, T, H, W, seed):
    from sensorium_amd.synthetic import make_batch
    x, _ = make_batch(B, T, H, W, (8,), seed=seed)
    return x                                                     # (B, 5, T, H, W) fp32: video 0..255, behaviour / pupil scalars


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("shape", [(2, 4, 9, 11, 16), (3, 5, 6, 8, 64), (1, 2, 36, 64, 64), (2, 3, 5, 7, 24)])
@pytest.mark.parametrize("with_pe", [True, False])
def test_stem_train_forward_backward(shape, dtype, with_pe):
    from sensorium_amd import ops
    from sensorium_amd.dwiseneuro import DepthwiseCore
    B, T, H, W, C0 = shape
    torch.manual_seed(C0 + T)
    core = DepthwiseCore(in_channels=5, features=(C0,), spatial_strides=(1,), expansion_ratio=3, se_reduce_ratio=4)
    g = torch.Generator().manual_seed(3)
    core.stem[0].weight.data = torch.randn(C0, 5, 1, 1, 1, generator=g) * 0.02
    bn = core.stem[1].bn
    bn.weight.data = torch.rand(C0, generator=g) + 0.5
    bn.bias.data = torch.randn(C0, generator=g) * 0.2
    bn.running_mean.data = torch.randn(C0, generator=g) * 0.1
    bn.running_var.data = torch.rand(C0, generator=g) + 0.5
    x = _inputs(B, T, H, W, seed=11)
    sd = {"core.stem.0.weight": core.stem[0].weight.detach().double().requires_grad_(True),
          "core.stem.1.bn.weight": bn.weight.detach().double().requires_grad_(True),
          "core.stem.1.bn.bias": bn.bias.detach().double().requires_grad_(True),
          "core.stem.1.bn.running_mean": bn.running_mean.clone().double(), "core.stem.1.bn.running_var": bn.running_var.clone().double(),
          "core.stem.1.bn.num_batches_tracked": bn.num_batches_tracked.clone()}
    # ---- oracle
    xl = x.double().permute(0, 2, 3, 4, 1)                        # channels-last
    new_stats = {}
    ref = orc.batch_norm(orc.pointwise(xl, sd["core.stem.0.weight"]), "core.stem.1.bn", sd, True, new_stats)
    pe = None
    if with_pe:
        inv_freq = orc.pe_inv_freq(C0)
        ref = ref + orc.pe_table(C0, T, H, W, inv_freq, torch.float64)
    gout = torch.randn(ref.shape, generator=torch.Generator().manual_seed(5)).double()
    (ref * gout).sum().backward()
    # ---- HIP
    core = core.to(dev()).train()
    if with_pe:
        pe = tuple(t.to(dev()) for t in ops.pe_axis_tables(C0, orc.pe_inv_freq(C0), T, H, W))
    bnd = core.stem[1].bn
    out = ops.StemFn.apply(x.to(dev()), core.stem[0].weight, bnd.weight, bnd.bias, core, dtype, pe)
    out.backward(gout.to(dev()).to(dtype))
    torch.cuda.synchronize()
    ft, gt = (1e-3, 1e-3) if dtype == torch.float32 else (4e-2, 8e-2)
    assert rel(out.float(), ref) < ft
    assert rel(bnd.running_mean, new_stats["core.stem.1.bn.running_mean"]) < 1e-4
    assert rel(bnd.running_var, new_stats["core.stem.1.bn.running_var"]) < 1e-4
    assert int(bnd.num_batches_tracked) == 1
    for mine, key in ((core.stem[0].weight.grad, "core.stem.0.weight"), (bnd.weight.grad, "core.stem.1.bn.weight"),
                      (bnd.bias.grad, "core.stem.1.bn.bias")):
        assert mine is not None and rel(mine.reshape(-1), sd[key].grad.reshape(-1)) < gt, key


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_stem_eval_forward(dtype):
    from sensorium_amd import ops
    from sensorium_amd.dwiseneuro import DepthwiseCore
    B, T, H, W, C0 = 2, 3, 6, 10, 32
    torch.manual_seed(1)
    core = DepthwiseCore(in_channels=5, features=(C0,), spatial_strides=(1,), expansion_ratio=3, se_reduce_ratio=4)
    bn = core.stem[1].bn
    g = torch.Generator().manual_seed(4)
    core.stem[0].weight.data = torch.randn(C0, 5, 1, 1, 1, generator=g) * 0.02
    bn.running_mean.data = torch.randn(C0, generator=g)
    bn.running_var.data = torch.rand(C0, generator=g) + 0.5
    x = _inputs(B, T, H, W, seed=2)
    sd = {"core.stem.0.weight": core.stem[0].weight.detach().double(), "core.stem.1.bn.weight": bn.weight.detach().double(),
          "core.stem.1.bn.bias": bn.bias.detach().double(), "core.stem.1.bn.running_mean": bn.running_mean.double(),
          "core.stem.1.bn.running_var": bn.running_var.double()}
    ref = orc.batch_norm(orc.pointwise(x.double().permute(0, 2, 3, 4, 1), sd["core.stem.0.weight"]), "core.stem.1.bn", sd, False)
    core = core.to(dev()).eval()
    before = {k: v.clone() for k, v in core.state_dict().items()}
    with torch.no_grad():
        out = ops.StemFn.apply(x.to(dev()), core.stem[0].weight, core.stem[1].bn.weight, core.stem[1].bn.bias, core, dtype, None)
    assert rel(out.float(), ref) < (1e-3 if dtype == torch.float32 else 4e-2)
    for k, v in core.state_dict().items():
        assert torch.equal(v, before[k]), k
