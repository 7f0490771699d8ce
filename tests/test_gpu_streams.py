"""The backward's second stream (sensorium_amd/ops.py: the stand-alone conv_pw weight-gradient GEMMs are launched on a side
stream behind an event and joined at the end of the backward pass / before the optimizer / before a gradient bucket's
all-reduce).  Same gradients as the single-stream order; the join must cover readers that only synchronise the main stream."""
import pytest
import torch

pytestmark = pytest.mark.gpu

from tests.gpu_helpers import dev  # noqa: E402


def _model(dtype):
    from sensorium_amd.dwiseneuro import DwiseNeuro
    torch.manual_seed(3)
    net = DwiseNeuro(readout_outputs=(40,), core_features=(64, 128, 128, 256), spatial_strides=(2, 1, 2, 1), expansion_ratio=7,
                     se_reduce_ratio=32, cortex_features=(64, 128), drop_rate=0.0, drop_path_rate=0.0, compute_dtype=dtype)
    return net.to(dev()).train()


def _grads(net, x, side):
    from sensorium_amd import ops
    ops.set_side_stream(side)
    try:
        net.zero_grad(set_to_none=True)
        out = net(x)[0]
        out.float().square().mean().backward()
        # .cpu() synchronises the MAIN stream only: the end-of-backward join must already be queued on it
        return {n: p.grad.detach().float().cpu().clone() for n, p in net.named_parameters()}
    finally:
        ops.set_side_stream(False)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_side_stream_gradients_match_single_stream(dtype):
    import sensorium_amd._lib as L
    from sensorium_amd import ops
    net = _model(dtype)
    x = torch.randn(2, 5, 4, 12, 16, device=dev()) * 30 + 60
    g1 = _grads(net, x, side=False)
    g2 = _grads(net, x, side=True)
    assert ops._SIDE and not any(st.pending for st in ops._SIDE.values()), "the side stream was never used / never joined"
    from tests.gpu_helpers import analytically_zero_grad
    gnorm = sum(float(v.norm()) ** 2 for v in g1.values()) ** 0.5
    for n in g1:
        a, b = g1[n], g2[n]
        if dtype == torch.float32:
            tol = 2e-4                                          # fp32 atomics reorder the sums
        elif "conv_pw.0.weight" in n:
            # the deferred gradients themselves; bf16 activations make every gradient noisy run to run (DESIGN.md section 2:
            # fp32 atomics order the BatchNorm sums differently), so the gate is the direction, as in test_gpu_bf16_depth.py
            cos = float((a * b).sum() / (a.norm() * b.norm() + 1e-30))
            assert cos > 0.98, (n, cos)
            continue
        else:
            continue
        if analytically_zero_grad(n):
            continue
        assert float((a - b).norm()) <= tol * float(a.norm()) + 1e-6 * gnorm, n
    # blocks whose conv_pw backward is not the fused kernel are the deferred ones: at least one exists in this model
    import ctypes as C
    a = L.BlockArgs()
    a.dtype = L.DWN_BF16 if dtype == torch.bfloat16 else L.DWN_F32
    a.B, a.T, a.Hin, a.Win, a.Cin, a.Cmid, a.defer_pw_wgrad = 2, 4, 6, 8, 128, 896, 1
    assert L.lib.dwn_block_pw_wgrad_deferred(C.byref(a)) == 1
    a.defer_pw_wgrad = 0
    assert L.lib.dwn_block_pw_wgrad_deferred(C.byref(a)) == 0


def test_optimizer_step_joins_the_side_stream():
    """A full train_step (fused AdamW reads the gradients through raw pointers) with the side stream on and off: same update."""
    from sensorium_amd import ops
    from sensorium_amd.argus_models import MouseModel
    from sensorium_amd.synthetic import make_batch
    kw = dict(readout_outputs=(24,), core_features=(64, 128, 128), spatial_strides=(2, 1, 1), expansion_ratio=7, se_reduce_ratio=32,
              cortex_features=(32, 64), drop_rate=0.0, drop_path_rate=0.0)
    params = {"nn_module": ("dwiseneuro", kw), "loss": ("mice_poisson", {}), "optimizer": ("AdamW", {"lr": 1e-3, "weight_decay": 0.05}),
              "device": "cuda:0", "amp": False, "iter_size": 1}
    batch = make_batch(2, 4, 12, 16, (24,), seed=5, device=dev())
    res = []
    for side in (False, True):
        ops.set_side_stream(side)
        try:
            torch.manual_seed(0)
            m = MouseModel(params)
            for _ in range(2):
                m.train_step(batch)
            # (parameters whose gradient is analytically zero receive +-lr of summation noise from Adam, run to run)
            from tests.gpu_helpers import analytically_zero_grad
            res.append(torch.cat([p.detach().reshape(-1) for n, p in m.nn_module.named_parameters()
                                  if not analytically_zero_grad(n)]).cpu())
        finally:
            ops.set_side_stream(False)
    assert float((res[0] - res[1]).norm()) <= 1e-3 * float(res[0].norm())
