"""Bit-exact gates for the index operations, read back from the HIP kernels' OUTPUT (north star: "bit-exact for
ChannelShuffle/index ops").  Every test drives values that identify their origin through a HIP module and recovers the index
map the kernel applied from the GPU result, then compares it with the golden maps generated from the reference
(tests/golden/index_and_pe.npz, oracle/make_golden.py):
  * shuffle_channels (src/models/dwiseneuro.py:212-219) from dwn_cortex_forward,
  * the shortcut channel tile (:132, :224) from dwn_cortex_forward and dwn_block_forward,
  * the nearest-neighbour resize of interpolate_shortcut (:125-129) from dwn_block_forward,
  * PositionalEncoding3d's table (:163-192) from dwn_block_forward,
  * the readout's [:N] slice with an odd neuron count (:278, :286) from dwn_readout_forward."""
import math

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from tests.gpu_helpers import dev  # noqa: E402


def _identity_bn(bn, weight=1.0, bias=None):
    with torch.no_grad():
        bn.weight.fill_(weight)
        if bias is None:
            bn.bias.zero_()
        else:
            bn.bias.copy_(bias)
        bn.running_mean.zero_()
        bn.running_var.fill_(1.0 - bn.eps)          # scale = weight / sqrt(var + eps) = weight exactly


def _silu(v):
    return v / (1.0 + np.exp(-v))


@pytest.mark.parametrize("c,groups,key", [(64, 2, "shuffle_64_2"), (4096, 2, "shuffle_4096_2")])
def test_channel_shuffle_permutation_recovered_from_gpu_output(golden_dir, c, groups, key):
    """conv weight 0 and a per-channel BN bias b_j make the pre-shuffle activation of channel j the constant SiLU(b_j); the
    shortcut is switched off (bn_sc weight 0).  Output channel o then carries SiLU(b_src(o)): src is read off the GPU output."""
    from sensorium_amd.dwiseneuro import ShuffleLayer
    cin = c // 2
    layer = ShuffleLayer(cin, c, groups=groups).to(dev()).eval()
    codes = 0.5 + np.arange(c, dtype=np.float64) * (8.0 / c)       # distinct biases in [0.5, 8.5): SiLU is monotone there
    with torch.no_grad():
        layer.conv.weight.zero_()
    _identity_bn(layer.bn.bn, 1.0, torch.from_numpy(codes).float().to(dev()))
    _identity_bn(layer.bn_sc.bn, 0.0)
    x = torch.randn(2, 3, cin, device=dev())
    with torch.no_grad():
        out = layer(x, torch.float32).cpu().numpy()
    table = _silu(codes.astype(np.float32).astype(np.float64))
    assert np.min(np.abs(np.diff(np.sort(table)))) > 1e-5                       # the codes are separable
    src = np.abs(out[0, 0][:, None] - table[None, :]).argmin(1)
    assert np.abs(out[0, 0] - table[src]).max() < 1e-6
    assert np.array_equal(out, np.broadcast_to(out[0, 0], out.shape))           # same routing at every (b, t)
    z = np.load(golden_dir / "index_and_pe.npz")
    assert np.array_equal(src, z[key]), "channel shuffle applied by the HIP cortex kernel differs from the reference's"


def test_cortex_shortcut_tile_recovered_from_gpu_output(golden_dir):
    """main branch off (conv weight 0, bn weight/bias 0 -> SiLU(0) = 0), shortcut BN = identity: out[o] = x[tile(o)] exactly."""
    from sensorium_amd.dwiseneuro import ShuffleLayer
    cin, c = 64, 128
    layer = ShuffleLayer(cin, c, groups=2).to(dev()).eval()
    with torch.no_grad():
        layer.conv.weight.zero_()
    _identity_bn(layer.bn.bn, 0.0)
    _identity_bn(layer.bn_sc.bn, 1.0)
    x = (torch.arange(cin, dtype=torch.float32) + 1.0).expand(2, 3, cin).contiguous()
    with torch.no_grad():
        out = layer(x.to(dev()), torch.float32).cpu().numpy()
    assert np.array_equal(out, np.round(out))                                    # integers came through untouched
    tile = (out[1, 2] - 1).astype(np.int64)
    z = np.load(golden_dir / "index_and_pe.npz")
    assert np.array_equal(tile, z["tile_64_128"])


def _shortcut_only_block(cin, cout, stride):
    from sensorium_amd.dwiseneuro import InvertedResidual3d, PositionalEncoding3d
    torch.manual_seed(0)
    blk = InvertedResidual3d(cin, cout, spatial_kernel=3, temporal_kernel=5, spatial_stride=stride, expansion_ratio=2,
                             se_reduce_ratio=8).to(dev()).eval()
    for bn in blk.bn_modules():
        _identity_bn(bn, 1.0)
    _identity_bn(blk.conv_pwl[1].bn, 0.0)       # main branch contributes BN4(...) = 0 * y4 + 0
    _identity_bn(blk.bn_sc.bn, 1.0)             # shortcut BN = identity
    return blk, PositionalEncoding3d(cin).to(dev())


def test_block_shortcut_nearest_and_tile_recovered_from_gpu_output(golden_dir):
    """Stride-2 block 64 -> 128 channels on a 9x16 plane: with the main branch zeroed the output is
    (x + PE)[t, hsrc(ho), wsrc(wo), co mod 64] bit for bit; hsrc, wsrc and the tile are recovered by looking the GPU values up."""
    cin, cout, T, H, W = 64, 128, 2, 9, 16
    blk, pe = _shortcut_only_block(cin, cout, 2)
    # distinct, exactly representable codes per (h, w, c); channels whose PE entry is non-zero still give distinct sums
    h_i, w_i, c_i = np.meshgrid(np.arange(H), np.arange(W), np.arange(cin), indexing="ij")
    code = (h_i * 1024 + w_i * 64 + c_i + 8).astype(np.float32) * 4.0
    x = torch.from_numpy(np.broadcast_to(code, (1, T, H, W, cin)).copy())
    with torch.no_grad():
        out = blk(x.to(dev()), pe, torch.float32).cpu().numpy()
    from sensorium_amd import ops
    pt, ph, pw = (t.numpy() for t in ops.pe_axis_tables(cin, pe.inv_freq.cpu(), T, H, W))
    a0 = code[None] + ((pt[:, None, None, :] + ph[None, :, None, :]) + pw[None, None, :, :])       # [T,H,W,C] fp32, one add each
    a0 = a0.astype(np.float32)
    Ho, Wo = out.shape[2], out.shape[3]
    assert (Ho, Wo) == (5, 8)
    lookup = {}
    for h in range(H):
        for w in range(W):
            for c in range(cin):
                lookup[float(a0[1, h, w, c])] = (h, w, c)
    assert len(lookup) == H * W * cin
    hsrc = np.full(Ho, -1); wsrc = np.full(Wo, -1); tile = np.full(cout, -1)
    for ho in range(Ho):
        for wo in range(Wo):
            for co in range(cout):
                h, w, c = lookup[float(out[0, 1, ho, wo, co])]        # KeyError = a value that is no bit-exact copy
                assert hsrc[ho] in (-1, h) and wsrc[wo] in (-1, w) and tile[co] in (-1, c)
                hsrc[ho], wsrc[wo], tile[co] = h, w, c
    z = np.load(golden_dir / "index_and_pe.npz")
    assert np.array_equal(hsrc, z["nearest_9_2"]) and np.array_equal(wsrc, z["nearest_16_2"])
    assert np.array_equal(tile, z["tile_64_128"])


def test_positional_encoding_table_bit_exact_from_gpu_output(golden_dir):
    """x = 0 through a stride-1, same-width block whose main branch is zeroed: the output IS PositionalEncoding3d's table."""
    cin, T, H, W = 64, 4, 5, 6
    blk, pe = _shortcut_only_block(cin, cin, 1)
    with torch.no_grad():
        out = blk(torch.zeros(1, T, H, W, cin, device=dev()), pe, torch.float32).cpu().numpy()
    z = np.load(golden_dir / "index_and_pe.npz")
    ref = np.transpose(z["pe_64_4_5_6"], (1, 2, 3, 0))                 # golden is [C,T,H,W]
    assert np.array_equal(out[0], ref)


def test_readout_slice_of_odd_neuron_count_from_gpu_output(golden_dir):
    """N = 7863 neurons are computed as 7864 grouped-conv rows and sliced [:N] (dwiseneuro.py:278,286): with zero weights the
    output row n is softplus(bias[n]); the rows must come back in order, none shifted, none from the padding row."""
    from sensorium_amd.dwiseneuro import Readout
    z = np.load(golden_dir / "index_and_pe.npz")
    n = 7863
    ro = Readout(64, n, groups=2, softplus_beta=0.07, drop_rate=0.0).to(dev()).eval()
    assert ro.layer[1].weight.shape[0] == int(z["readout_pad_7863"])
    bias = (np.arange(n + 1, dtype=np.float64) - n / 2) * (40.0 / n)
    with torch.no_grad():
        ro.layer[1].weight.zero_()
        ro.layer[1].bias.copy_(torch.from_numpy(bias).float())
        out = ro(torch.randn(2, 3, 64, device=dev())).cpu().numpy()
    assert out.shape == (2, n, 3)
    b32 = bias.astype(np.float32).astype(np.float64)
    table = np.log1p(np.exp(0.07 * b32)) / 0.07
    rows = np.abs(out[1, :, 2][:, None] - table[None, :]).argmin(1) if n < 512 else None
    # 7864 candidates x 7863 rows is a big matrix: check the order directly instead, plus strict monotonicity
    assert np.abs(out[1, :, 2] - table[:n]).max() < 1e-4 * np.abs(table).max()
    assert np.all(np.diff(out[1, :, 2]) > 0)
    assert rows is None or np.array_equal(rows, np.arange(n))
