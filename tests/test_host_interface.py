"""CPU tests of the host side: the C-ABI library loads without a GPU and exports every symbol include/dwn.h
declares; ctypes struct layouts match; the drop-in module reproduces the reference's state_dict layout; host
index/PE helpers agree bit-exactly with the oracle; the product path refuses CPU tensors (no fallback)."""
import ctypes as C
import re
from pathlib import Path

import numpy as np
import pytest
import torch

from oracle import dwiseneuro_oracle as orc

ROOT = Path(__file__).resolve().parents[1]


def declared_functions():
    text = (ROOT / "include" / "dwn.h").read_text()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(dwn_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    import sensorium_amd._lib as L
    names = declared_functions()
    assert len(names) >= 30
    for n in names:
        assert hasattr(L.lib, n), f"{n} declared in include/dwn.h but not exported"
        assert n in L.SYMBOLS, f"{n} has no ctypes prototype in sensorium_amd/_lib.py"
    assert L.lib.dwn_abi_version() == 7


def test_struct_layouts_match():
    import sensorium_amd._lib as L
    for cname, struct in L._STRUCTS.items():
        assert L.lib.dwn_sizeof(cname.encode()) == C.sizeof(struct), cname
    assert L.lib.dwn_sizeof(b"no_such_struct") == -1


def test_state_dict_layout_matches_reference(golden_dir):
    from sensorium_amd import DwiseNeuro
    z = np.load(golden_dir / "tiny_model_eval.npz")
    ref_keys = [k[3:] for k in z.files if k.startswith("sd:")]
    model = DwiseNeuro(readout_outputs=(7, 10), core_features=(8, 8, 16), spatial_strides=(2, 1, 2),
                       expansion_ratio=3, se_reduce_ratio=4, cortex_features=(32, 64))
    sd = model.state_dict()
    assert list(sd.keys()) == ref_keys                      # same keys, same ORDER (ModelEma zips values in order)
    for k in ref_keys:
        assert tuple(sd[k].shape) == tuple(z["sd:" + k].shape), k
    res = model.load_state_dict({k: torch.from_numpy(z["sd:" + k]) for k in ref_keys}, strict=True)
    assert not res.missing_keys and not res.unexpected_keys
    # full-size (10 mice): 365 entries, 170.66 M parameters
    num_neurons = [7863, 7908, 8202, 7939, 8122, 7440, 7928, 8285, 7671, 7495]
    big = DwiseNeuro(readout_outputs=num_neurons, expansion_ratio=7)
    assert len(big.state_dict()) == 365
    assert abs(sum(p.numel() for p in big.parameters()) - 170.66e6) < 0.02e6
    assert big.state_dict()["core.blocks.1.spat_covn_dw.0.weight"].shape == (448, 1, 1, 3, 3)
    assert big.state_dict()["readouts.0.layer.1.weight"].shape == (7864, 2048, 1)


def test_module_is_deepcopyable_and_init_weights_compatible():
    import copy
    import math
    from sensorium_amd import DwiseNeuro
    model = DwiseNeuro(readout_outputs=(7,), core_features=(8, 8), spatial_strides=(2, 1), expansion_ratio=3,
                       se_reduce_ratio=4, cortex_features=(16, 32))
    clone = copy.deepcopy(model)
    assert list(clone.state_dict().keys()) == list(model.state_dict().keys())
    # reference init rule (src/utils.py:46-56) finds the layers by isinstance
    n_conv = n_bn = 0
    for m in model.modules():
        if isinstance(m, (torch.nn.Conv1d, torch.nn.Conv3d)):
            fan_out = math.prod(m.kernel_size) * m.out_channels // m.groups
            torch.nn.init.normal_(m.weight, 0, math.sqrt(2.0 / fan_out))
            n_conv += 1
        elif isinstance(m, (torch.nn.BatchNorm1d, torch.nn.BatchNorm3d)):
            n_bn += 1
    assert n_conv == 1 + 2 * 6 + 2 + 1 and n_bn == 1 + 2 * 5 + 2 * 2


def test_host_index_and_pe_helpers_match_oracle(golden_dir):
    from sensorium_amd import ops
    for out_size, in_size in ((18, 36), (5, 9), (3, 5), (32, 64), (6, 11)):
        assert np.array_equal(ops.nearest_src_index(out_size, in_size), orc.nearest_src_index(out_size, in_size))
        inv = ops.inverse_index(ops.nearest_src_index(out_size, in_size), in_size)
        assert sorted(i for i in inv if i >= 0) == list(range(out_size))
    for c, (t, h, w) in ((64, (4, 5, 6)), (8, (6, 9, 11)), (256, (2, 5, 8))):
        mine = ops.pe_axis_tables(c, orc.pe_inv_freq(c), t, h, w)
        ref = orc.pe_axis_tables(c, t, h, w)
        for a, b in zip(mine, ref):
            assert torch.equal(a, b)
        full = mine[0][:, None, None, :] + mine[1][None, :, None, :] + mine[2][None, None, :, :]
        z = np.load(golden_dir / "index_and_pe.npz")
        key = f"pe_{c}_{t}_{h}_{w}"
        assert np.array_equal(full.permute(3, 0, 1, 2).numpy(), z[key])     # bit-exact vs the reference module


def test_no_cpu_fallback():
    from sensorium_amd import DwiseNeuro, MicePoissonLoss
    model = DwiseNeuro(readout_outputs=(7,), core_features=(8, 8), spatial_strides=(2, 1), expansion_ratio=3,
                       se_reduce_ratio=4, cortex_features=(16, 32))
    with pytest.raises(RuntimeError, match="GPU"):
        model(torch.zeros(1, 5, 4, 9, 11))
    with pytest.raises(RuntimeError, match="GPU"):
        MicePoissonLoss()([torch.ones(1, 7, 4)], ([torch.ones(1, 7, 4)], torch.ones(1, 1)))


def test_deep_chunk_and_synthetic_batch():
    from sensorium_amd.argus_models import deep_chunk
    from sensorium_amd.synthetic import make_batch
    x, (targets, w) = make_batch(4, 6, 9, 11, (7, 10), seed=1)
    assert x.shape == (4, 5, 6, 9, 11) and w.shape == (4, 2) and targets[1].shape == (4, 10, 6)
    assert torch.equal(w.sum(1), torch.ones(4))
    assert float(targets[0][1].abs().sum()) == 0.0          # sample 1 belongs to mouse 1: zero target for mouse 0
    chunks = deep_chunk([x, [targets, w]], 2)
    assert len(chunks) == 2 and chunks[0][0].shape[0] == 2 and chunks[1][1][0][1].shape == (2, 10, 6)


def test_fill_distill_targets_matches_reference_loop():
    """argus_models.py:31-41 restated literally (python loop over argwhere) vs the vectorised product code."""
    from sensorium_amd.argus_models import fill_distill_targets
    g = torch.Generator().manual_seed(0)
    b, mice, t, ratio = 6, 3, 4, 0.36
    sizes = (5, 7, 4)
    weights = torch.zeros(b, mice)
    weights[torch.arange(b), torch.arange(b) % mice] = 1.0
    targets = [torch.rand(b, n, t, generator=g) * weights[:, m][:, None, None] for m, n in enumerate(sizes)]
    teacher = [torch.rand(b, n, t, generator=g) + 1.0 for n in sizes]
    # literal restatement of the reference loop
    ref_t = [x.clone() for x in targets]
    ref_w = weights.clone()
    mask = ref_w == 0.0
    dw = ratio / (1.0 - ratio) * ref_w.sum() / mask.sum()
    for bi, mi in torch.argwhere(mask):
        ref_t[mi][bi] = teacher[mi][bi]
        ref_w[bi, mi] = dw
    mine_t = [x.clone() for x in targets]
    mine_w = weights.clone()
    fill_distill_targets(teacher, (mine_t, mine_w), ratio)
    assert torch.equal(mine_w, ref_w)
    for a, b_ in zip(mine_t, ref_t):
        assert torch.equal(a, b_)


def test_corr_and_window_indexes(golden_dir):
    from sensorium_amd.metrics import corr
    from sensorium_amd.predictors import IndexesGenerator
    z = np.load(golden_dir / "corr.npz")
    assert np.array_equal(corr(z["a"], z["b"], axis=0), z["corr"])
    gen = IndexesGenerator(16, 2, "last")
    assert (gen.behind, gen.ahead, gen.width) == (30, 0, 31)
    assert gen.make_indexes(30) == list(range(0, 31, 2)) == orc.window_indexes(30, 16, 2)


def test_grad_bucket_layout_for_the_benchmark_model():
    """Reverse registration order, ~12 MB core buckets, the big readout weight first (ready first in backward)."""
    import sys
    from pathlib import Path
    sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
    from bench import model_params
    from sensorium_amd import DwiseNeuro
    from sensorium_amd.ddp import GradBuckets
    model = DwiseNeuro(**model_params(7)["nn_module"][1])
    buckets = GradBuckets(model)
    sizes = [b["flat"].numel() * 4 / 2 ** 20 for b in buckets.buckets]
    params = [p for p in model.parameters() if p.requires_grad]
    assert buckets.buckets[0]["params"][0] is params[-1]                    # last registered parameter leads
    assert buckets.num_elements() == sum(p.numel() for p in params)
    # slices are padded to 16 bytes (the backward kernels write them with 16-byte stores) + one "used" flag per optional
    # parameter: < 0.01 % over the parameter count, every slice aligned
    extra = sum(b["flat"].numel() for b in buckets.buckets) - sum(p.numel() for p in params)
    assert 0 <= extra <= 4 * len(params)
    assert all(o % 4 == 0 for b in buckets.buckets for o in b["offsets"])
    assert sizes[0] > 60 and all(s < 20 for s in sizes[1:]) and len(sizes) >= 3, sizes
    assert sizes[-1] < 12, sizes                                            # the only bucket exposed after backward



def test_pooled_drop_path_draws_follow_the_reference_rates():
    """DwiseNeuro._draw_drop_paths: one uniform draw for all stochastic-depth layers of a forward pass.  Per layer the factor
    is Bernoulli(keep) / keep with keep = 1 - 0.1 * i / 9 in core block i and 0.9 in every cortex layer
    (reference: src/models/dwiseneuro.py:46-54, 317, 256/382); a layer consumes its draw exactly once."""
    import torch
    from sensorium_amd.dwiseneuro import DropPath, DwiseNeuro
    net = DwiseNeuro(readout_outputs=(8,), core_features=(8,) * 9, spatial_strides=(1,) * 9, cortex_features=(16, 16, 16),
                     expansion_ratio=3, se_reduce_ratio=4, drop_path_rate=0.1)
    net.train()
    layers = [m for m in net.modules() if isinstance(m, DropPath)]
    assert len(layers) == 12
    want = [0.1 * i / 9 for i in range(9)] + [0.1] * 3
    assert [round(m.drop_prob, 6) for m in layers] == [round(p, 6) for p in want]
    torch.manual_seed(0)
    B, rounds = 512, 40
    kept = torch.zeros(12)
    for _ in range(rounds):
        net._draw_drop_paths(B, torch.device("cpu"))
        for i, m in enumerate(layers):
            if m.drop_prob == 0.0:
                assert m._pooled is None and m.sample(B, torch.device("cpu")) is None        # layer 0 never drops
                continue
            f = m.sample(B, torch.device("cpu"))
            assert m._pooled is None, "a layer's draw must be consumed exactly once"
            keep = 1.0 - m.drop_prob
            vals = torch.unique(f)
            assert all(float(v) == 0.0 or abs(float(v) - 1.0 / keep) < 1e-6 for v in vals), vals
            kept[i] += float((f > 0).sum())
    n = B * rounds
    for i, m in enumerate(layers):
        if m.drop_prob == 0.0:
            continue
        keep = 1.0 - m.drop_prob
        sigma = (keep * (1 - keep) / n) ** 0.5
        assert abs(kept[i] / n - keep) < 5 * sigma + 1e-9, (i, kept[i] / n, keep)
    # a stale pooled draw of another batch size is not used
    net._draw_drop_paths(4, torch.device("cpu"))
    f = layers[5].sample(7, torch.device("cpu"))
    assert f.shape == (7,) and layers[5]._pooled is None
    # a changed rate (drop-path schedule) rebuilds the cached keep table; drop_prob 1.0 gives factor 0, not NaN
    layers[5].drop_prob = 1.0
    net._draw_drop_paths(16, torch.device("cpu"))
    f = layers[5].sample(16, torch.device("cpu"))
    assert torch.equal(f, torch.zeros(16))
    # eval: no draws
    net.eval()
    net._draw_drop_paths(4, torch.device("cpu"))
    assert all(m.sample(4, torch.device("cpu")) is None for m in layers)


def test_deterministic_build_of_the_library_loads_and_matches_the_abi():
    """csrc/Makefile builds the ordered-reduction variant next to the product library (DWN_DETERMINISTIC=1 selects it):
    same ABI version, same struct layouts, every symbol of include/dwn.h."""
    import ctypes as C
    from pathlib import Path
    import sensorium_amd._lib as L
    path = Path(L.__file__).resolve().parent / "csrc" / "libdwiseneuro_hip_det.so"
    assert path.exists(), "run __graft_entry__.build()"
    det = C.CDLL(str(path))
    det.dwn_abi_version.restype = C.c_int
    assert det.dwn_abi_version() == L.lib.dwn_abi_version()
    det.dwn_sizeof.restype = C.c_size_t
    det.dwn_sizeof.argtypes = [C.c_char_p]
    for cname, struct in L._STRUCTS.items():
        assert det.dwn_sizeof(cname.encode()) == C.sizeof(struct), cname
    for name in L.SYMBOLS:
        assert hasattr(det, name), name


def test_grad_out_hands_a_bucket_slot_out_once_per_backward():
    """ops.grad_out: the first producer of a parameter's gradient writes into the parameter's all-reduce bucket slot, a second
    one in the same backward pass (a module applied twice) gets a tensor of its own — autograd then adds two different buffers
    instead of two aliases of one (round-3 advisor finding)."""
    import torch
    from sensorium_amd import ops
    p = torch.nn.Parameter(torch.zeros(3, 4))
    flat = torch.zeros(32)
    p._dwn_grad_slot = (flat, 8)
    g1 = ops.grad_out(p)
    g2 = ops.grad_out(p, zero=True)
    assert g1.data_ptr() == flat[8:].data_ptr() and g1.shape == p.shape
    assert g2.data_ptr() != g1.data_ptr() and not g2.any()
    p._dwn_slot_out = False                     # what GradBuckets' hook / zero_grad do
    assert ops.grad_out(p).data_ptr() == g1.data_ptr()


def test_fp32_eval_products_setting_reaches_every_gemm_module():
    import torch
    from sensorium_amd import DwiseNeuro, ops
    import sensorium_amd._lib as L
    net = DwiseNeuro(readout_outputs=(7,), core_features=(8, 8), spatial_strides=(2, 1), expansion_ratio=3, se_reduce_ratio=4,
                     cortex_features=(16, 32))
    blk, layer, ro = net.core.blocks[1], net.cortex.layers[0], net.readouts[0]
    assert ops._f32_products(blk) == L.F32_AUTO and ops._f32_products(ro, inference_readout=True) == L.F32_SPLIT3
    net.set_fp32_eval_products("native")
    assert {ops._f32_products(m) for m in (blk, layer)} == {L.F32_NATIVE}
    assert ops._f32_products(ro, inference_readout=True) == L.F32_NATIVE
    with pytest.raises(ValueError):
        net.set_fp32_eval_products("fp16")


def test_library_is_built_from_the_sources_in_the_tree():
    """Binaries are not in git; what the GPU box runs must be what is committed: the library carries the hash of the sources it was
    built from (csrc/Makefile HASH_SRCS) and sensorium_amd/_lib.py refuses to load one built from other sources."""
    import sensorium_amd._lib as L
    mk = (Path(L.__file__).resolve().parent / "csrc" / "Makefile").read_text()
    srcs = re.search(r"^SRCS = (.*)$", mk, re.M).group(1).split()
    hdrs = re.search(r"^HDRS = (.*)$", mk, re.M).group(1).split()
    assert tuple(srcs + hdrs) == L.HASH_SRCS
    assert L.lib.dwn_source_hash().decode() == L.source_hash()
